// qocx_sweepd.hip - K2d, the dense-state sweep: 8 <= S <= 32 states of a seed as the COLUMNS of
// MFMA GEMMs (17 <= n <= 32, padded to 32).
//
// The column-chain sweep (qocx_kernels.hip) walks the two triangular solves of a step once per
// state - 62 dependent stages each - and deals the states out to four waves: 46 us per step at
// S = 32. With many states the step is a dense contraction: K1b's sibling (qocx_lu.h inv_body)
// leaves P^-1 where the factors would be, and a sub-step is two complex 32 x 32 x S products on
// v_mfma_f64_16x16x4_f64,
//   forward:  Z = Q Psi,          Psi' = P^-1 Z
//   adjoint:  X = P^-H Lambda',   Lambda = Q^H X        (x goes to HBM for K3, as before)
// with the four waves of the workgroup owning one 16 x 16 tile of the result each (wave w: row tile
// w & 1, column tile w >> 1) and two LDS barriers per sub-step. Same step as
// _evolve_step_schroedinger_discrete (qoc/core/schroedingerdiscrete.py:483-497) with
// expm = P^-1 Q (expm.py:246) - the reference multiplies the n x n propagator onto the states too.
//
// LDS (112 KiB): two sets of step operands (Q and P^-1 images, 16 KiB each; the next step's set
// streams in by LDS-DMA while the current one is used; the adjoint gathers the transposed images,
// so that both directions read their A fragments along columns), two state matrices [32][32]
// row-major (Psi / Z, resp. Lambda / X: B fragments and C tiles are rows of 16 states), and one
// [S][32] vector array for the cost routines (which want one vector per state; the second one they
// need aliases the Z matrix). States and x vectors go to HBM in the layout K3 reads:
// [slot][state][32].
//
// Everything else follows sweep_kernel: time segments (phase, j_begin, j_end), squaring sub-steps,
// step costs, host-supplied cotangents, the capacity check. The unit adjoint is not offered here
// (the sweep is no longer what an evaluation with many states waits for).
#include "qocx_sweep_common.h"

namespace qocx {

namespace sweepd {

constexpr int NB = 2, NP = 32, MAT = NP * NP, SP = 32;
constexpr int Q_OFF = 0;                       // 2 x Q image
constexpr int PI_OFF = Q_OFF + 2 * MAT * 16;   // 2 x P^-1 image
constexpr int M0_OFF = PI_OFF + 2 * MAT * 16;  // state matrix (Psi, Lambda)
constexpr int M1_OFF = M0_OFF + NP * SP * 16;  // second matrix (Z, X); also `vecs` of the cost routines
constexpr int V_OFF = M1_OFF + NP * SP * 16;   // [S][NP] cotangent vectors of the cost routines
constexpr int LDS_BYTES = V_OFF + 32 * NP * 16;

__device__ __forceinline__ void lds_barrier() {
    // LDS hand-off only (a __syncthreads() would also wait for the fetch and the stores in flight)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// C tile (ti, tj) of A B (CONJ: conj(A) B): A = a 32 x 32 image in LDS, element (r, k) at
// a_img[k * NP + r]; B = a state matrix in LDS, element (k, s) at b_mat[k * SP + s]. 3M scheme.
template <bool CONJ>
__device__ __forceinline__ void gemm_tile(const double2* a_img, const double2* b_mat, int ti, int tj,
                                          d4& cre, d4& cim) {
    const int q = lane_id() >> 4, c = lane_id() & 15;
    d4 t1 = {0, 0, 0, 0}, t2 = {0, 0, 0, 0}, t3 = {0, 0, 0, 0};
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
        const double2 a = a_img[(4 * kk + q) * NP + 16 * ti + c];
        const double2 bv = b_mat[(4 * kk + q) * SP + 16 * tj + c];
        const double ai = CONJ ? -a.y : a.y;
        t1 = mfma_f64(a.x, bv.x, t1);
        t2 = mfma_f64(ai, bv.y, t2);
        t3 = mfma_f64(a.x + ai, bv.x + bv.y, t3);
    }
    cre = t1 - t2;
    cim = t3 - t1 - t2;
}

__global__ __launch_bounds__(256) void sweepd_kernel(SweepArgs args) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __builtin_amdgcn_s_setprio(3);
    double2* qbuf = reinterpret_cast<double2*>(smem + Q_OFF);
    double2* pibuf = reinterpret_cast<double2*>(smem + PI_OFF);
    double2* m0 = reinterpret_cast<double2*>(smem + M0_OFF);
    double2* m1 = reinterpret_cast<double2*>(smem + M1_OFF);
    double2* vecs = m1;  // [S][NP], only between sub-steps
    double2* lam = reinterpret_cast<double2*>(smem + V_OFF);
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, c = lane & 15, ti = w & 1, tj = w >> 1;
    const int i = lane % NP, h = lane / NP;  // (cost routines, wave 0)
    const int b = blockIdx.x, S = args.S, nsteps = args.nsteps;
    const bool tile_on = 16 * tj < S;  // this wave's column tile holds states
    const size_t cap = args.slot_cap;
    double2* states_b = args.states + (size_t)b * cap * S * NP;
    double2* xs_b = args.xs + (size_t)b * cap * S * NP;
    int* offs_b = args.offs + (size_t)b * (nsteps + 1);
    const size_t mbase = (size_t)b * nsteps;
    const int jb = args.j_begin, je = args.j_end;
    const bool do_fwd = (args.phase & 1) != 0, do_bwd = (args.phase & 2) != 0;
    if (jb > 0 || !do_fwd)
        if ((*(volatile int*)args.status) & 4) return;  // an earlier segment overflowed

    // ---- layout changes between the state matrix [k][s] and the vectors [s][k] -------------
    auto mat_to_vecs = [&](const double2* mat, double2* vec) {
        for (int e = tid; e < S * NP; e += 256) vec[e] = mat[(e % NP) * SP + e / NP];
    };
    auto vecs_to_mat = [&](const double2* vec, double2* mat) {
        for (int e = tid; e < NP * SP; e += 256) {
            const int k = e / SP, s = e % SP;
            mat[e] = s < S ? vec[s * NP + k] : make_double2(0, 0);
        }
    };
    // the states of slot `sl` from HBM into a state matrix / into the vectors
    auto load_matrix = [&](const double2* src_b, size_t sl, double2* mat) {
        for (int e = tid; e < NP * SP; e += 256) {
            const int s = e / NP, k = e % NP;  // (consecutive lanes: consecutive k of one state)
            mat[k * SP + s] = s < S ? src_b[(sl * S + s) * NP + k] : make_double2(0, 0);
        }
    };
    // ---- the step operands: wave w fetches eight of the 32 KiB pieces -----------------------
    auto issue_fetch = [&](size_t m, int par, bool adjoint) {
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const int j = 8 * w + jj, jl = j & 15;
            const double2* img = (j < 16 ? args.q_img : args.lu_img) + m * MAT;
            double2* dst = (j < 16 ? qbuf : pibuf) + par * MAT + jl * 64;
            // plain image: piece jl is its KiB jl; transposed: LDS element (col 2 jl + l / 32,
            // row l % 32) is image element (row 2 jl + l / 32, col l % 32)
            const size_t el = adjoint ? (size_t)(lane % NP) * NP + 2 * jl + lane / NP
                                      : (size_t)jl * 64 + lane;
            dma16(img + el, dst);
        }
    };
    // this wave's C tile: rows 16 ti + 4 r + q, column (state) 16 tj + c
    auto store_tile = [&](const d4& cre, const d4& cim, double2* mat, double2* hbm_b, size_t sl) {
        const int s = 16 * tj + c;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = 16 * ti + 4 * r + q;
            const double2 v = make_double2(cre[r], cim[r]);
            mat[k * SP + s] = v;
            if (hbm_b != nullptr && s < S) hbm_b[(sl * S + s) * NP + k] = v;
        }
    };

    double cost = 0;
    int slot = 0;
    bool overflow = false;
    if (do_fwd) {
        if (jb == 0) {
            for (int e = tid; e < NP * SP; e += 256) {
                const int s = e / NP, k = e % NP;
                const double2 p = s < S ? args.psi0[s * NP + k] : make_double2(0, 0);
                m0[k * SP + s] = p;
                if (s < S) states_b[(size_t)s * NP + k] = p;
            }
        } else {  // resume: states, slot counter and partial cost left by the previous segment
            slot = offs_b[jb];
            cost = args.cost_out[b];
            load_matrix(states_b, (size_t)slot, m0);
        }
        lds_barrier();
    }
    // called behind a barrier: every state of `step` is in m0
    auto before_step = [&](int step) {
        const bool cost_step = step != 0 && args.has_step_costs && (step % args.cost_eval_step) == 0;
        if (cost_step) {
            mat_to_vecs(m0, vecs);
            lds_barrier();
            if (w == 0) cost += eval_costs<NB>(args, true, false, vecs, nullptr, h, i);
            lds_barrier();
        }
        if (args.step_states != nullptr)
            for (int e = tid; e < S * NP; e += 256)
                args.step_states[((size_t)b * (nsteps + 1) + step) * S * NP + e] =
                    m0[(e % NP) * SP + e / NP];
        if (tid == 0) offs_b[step] = slot;
    };

    // ---- forward sweep -------------------------------------------------------------------
    if (do_fwd) {
        issue_fetch(mbase + jb, 0, false);
        int nsub_next = 1 << step_squarings(args.s_arr[mbase + jb]);
        for (int step = jb; step < je; ++step) {
            const int par = (step - jb) & 1;
            const int nsub = nsub_next;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces have landed
            lds_barrier();                                     // ... and everybody else's
            if (step + 1 < je) {
                issue_fetch(mbase + step + 1, par ^ 1, false);
                nsub_next = 1 << step_squarings(args.s_arr[mbase + step + 1]);
            }
            before_step(step);
            const double2* qc = qbuf + par * MAT;
            const double2* pc = pibuf + par * MAT;
            for (int sub = 0; sub < nsub; ++sub) {
                if ((size_t)slot + 1 >= cap) {
                    overflow = true;
                    break;
                }
                d4 cre, cim;
                if (tile_on) {
                    gemm_tile<false>(qc, m0, ti, tj, cre, cim);
                    store_tile(cre, cim, m1, nullptr, 0);
                }
                lds_barrier();
                if (tile_on) {
                    gemm_tile<false>(pc, m1, ti, tj, cre, cim);
                    store_tile(cre, cim, m0, states_b, (size_t)slot + 1);
                }
                lds_barrier();
                ++slot;
            }
            if (overflow) break;
        }
    }
    if (overflow) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tid == 0) atomicOr(args.status, 4);
        return;
    }
    if (do_fwd) {
        if (je == nsteps) {
            before_step(nsteps);
            mat_to_vecs(m0, vecs);
            lds_barrier();
            if (w == 0) cost += eval_costs<NB>(args, false, true, vecs, nullptr, h, i);
            for (int e = tid; e < S * NP; e += 256) args.final_out[(size_t)b * S * NP + e] = vecs[e];
            lds_barrier();
        } else if (tid == 0) {
            offs_b[je] = slot;  // the next segment resumes from here
        }
        if (tid == 0) args.cost_out[b] = cost;
    }
    if (!do_bwd) return;

    // lambda (vectors) += host-supplied cotangent of the states at system step `step`
    auto inject = [&](int step) {
        if (args.inj_index == nullptr) return;
        const int row = args.inj_index[step];
        if (row < 0) return;
        for (int e = tid; e < S * NP; e += 256) {
            const double2 v = args.inj_bars[((size_t)b * args.inj_count + row) * S * NP + e];
            double2 l = lam[e];
            l.x += v.x;
            l.y += v.y;
            lam[e] = l;
        }
        lds_barrier();
    };

    // ---- adjoint sweep: Lambda lives in m0 ---------------------------------------------------
    if (je == nsteps) {
        if (!do_fwd) {  // final states of the forward segments
            slot = offs_b[nsteps];
            for (int e = tid; e < S * NP; e += 256) vecs[e] = states_b[(size_t)slot * S * NP + e];
        }
        for (int e = tid; e < S * NP; e += 256) lam[e] = make_double2(0, 0);
        lds_barrier();
        // cotangent seeds on the final states: non-step costs, and step costs if the final step
        // is a cost step (schroedingerdiscrete.py:412-416 evaluates them before the loop ends)
        if (w == 0)
            (void)eval_costs<NB>(args, (nsteps % args.cost_eval_step) == 0, true, vecs, lam, h, i);
        lds_barrier();
        inject(nsteps);
        vecs_to_mat(lam, m0);
    } else {  // resume the adjoint sweep below step je
        slot = offs_b[je];
        load_matrix(args.lam_buf + (size_t)b * S * NP, 0, m0);
    }
    lds_barrier();
    {
        issue_fetch(mbase + je - 1, 0, true);
        int nsub_next = 1 << step_squarings(args.s_arr[mbase + je - 1]);
        for (int step = je - 1, it = 0; step >= jb; --step, ++it) {
            const int par = it & 1;
            const int nsub = nsub_next;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            lds_barrier();
            if (step - 1 >= jb) {
                issue_fetch(mbase + step - 1, par ^ 1, true);
                nsub_next = 1 << step_squarings(args.s_arr[mbase + step - 1]);
            }
            const double2* qc = qbuf + par * MAT;   // Q^T: conj gives the rows of Q^H
            const double2* pc = pibuf + par * MAT;  // (P^-1)^T
            for (int sub = nsub - 1; sub >= 0; --sub) {
                if (slot <= 0) {
                    overflow = true;
                    break;
                }
                --slot;
                d4 cre, cim;
                if (tile_on) {  // X = P^-H Lambda'
                    gemm_tile<true>(pc, m0, ti, tj, cre, cim);
                    store_tile(cre, cim, m1, xs_b, (size_t)slot);
                }
                lds_barrier();
                if (tile_on) {  // Lambda = Q^H X
                    gemm_tile<true>(qc, m1, ti, tj, cre, cim);
                    store_tile(cre, cim, m0, nullptr, 0);
                }
                lds_barrier();
            }
            if (overflow) break;
            const bool cost_step = step != 0 && (step % args.cost_eval_step) == 0 && args.has_step_costs;
            const bool injected = step != 0 && args.inj_index != nullptr && args.inj_index[step] >= 0;
            if (cost_step || injected) {
                // step costs were evaluated on the states BEFORE evolving from `step`
                mat_to_vecs(m0, lam);
                if (cost_step)
                    for (int e = tid; e < S * NP; e += 256) vecs[e] = states_b[(size_t)slot * S * NP + e];
                lds_barrier();
                if (cost_step && w == 0) (void)eval_costs<NB>(args, true, false, vecs, lam, h, i);
                lds_barrier();
                inject(step);
                vecs_to_mat(lam, m0);
                lds_barrier();
            }
        }
    }
    if (overflow) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tid == 0) atomicOr(args.status, 4);
        return;
    }
    if (jb > 0)
        for (int e = tid; e < S * NP; e += 256)
            args.lam_buf[(size_t)b * S * NP + e] = m0[(e % NP) * SP + e / NP];
}

// ------------------------------------------------------------------------------------------
// K3d: the Krylov-chain adjoint of the Pade step with the states as GEMM columns
// ------------------------------------------------------------------------------------------
// krylov_grad_body (qocx_kernels.hip) spends 2 (M - 1) matrix-vector products and M rank-1 updates
// per (sub-step, state) on the vector unit. With X, Sigma = Psi + Psi', Delta = Psi - Psi' as
// 32 x S matrices the same sum abar = sum_i (a^H)^i X R_i^H, R_{M-1} = b_M Sigma,
// R_{i-1} = b_i W_i + a R_i (W_i = Sigma for odd i, Delta for even i) is evaluated by a second
// Horner recurrence, so that no chain has to be kept:
//     Y_{M-1} = X R_{M-1}^H,   Y_{i-1} = X R_{i-1}^H + a^H Y_i,   abar = Y_0
// - three complex 32 x 32 x {S, S, 32} products per order on the matrix cores, one LDS barrier per
// order, every matrix in LDS as a row-major image of pitch 33 (fragments along rows and along
// columns are both conflict free), the four waves owning one 16 x 16 tile of every result each.
namespace k3d {

constexpr int NP = 32, MAT = NP * NP, PM = 33;          // pitch of the row-major LDS matrices
constexpr int MBYTES = NP * PM * 16;
constexpr int A_OFF = 0;                                 // a, column-major pitch 32
constexpr int AH_OFF = A_OFF + MAT * 16;                 // a^H, column-major (general generators)
constexpr int X_OFF = AH_OFF + MAT * 16;                 // X [k][s]
constexpr int R_OFF = X_OFF + MBYTES;                    // two R matrices [k][s]
constexpr int Y_OFF = R_OFF + 2 * MBYTES;                // two Y matrices [r][c] (also Psi, Psi' on arrival)
constexpr int RED_OFF = Y_OFF + 2 * MBYTES;              // 4 partial sums per control
constexpr int LDS_BYTES = RED_OFF + 4 * 64 * 8;

struct Tile {
    d4 re, im;
};

// acc (3M partial sums) += A B over k = 0 .. 4 KS - 1; fa(kk) / fb(kk) give this lane's fragment
// elements A[16 ti + c][4 kk + q] and B[4 kk + q][16 tj + c]
// (fully unrolled: the fragment reads of all k-steps go out ahead of the first MFMA)
template <int KS, class FA, class FB>
__device__ __forceinline__ void gemm_acc_n(d4& t1, d4& t2, d4& t3, FA fa, FB fb) {
    double2 a[KS], bv[KS];
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
        a[kk] = fa(kk);
        bv[kk] = fb(kk);
    }
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
        t1 = mfma_f64(a[kk].x, bv[kk].x, t1);
        t2 = mfma_f64(a[kk].y, bv[kk].y, t2);
        t3 = mfma_f64(a[kk].x + a[kk].y, bv[kk].x + bv[kk].y, t3);
    }
}
template <class FA, class FB>
__device__ __forceinline__ void gemm_acc(d4& t1, d4& t2, d4& t3, int ks, FA fa, FB fb) {
    if (ks == 4) gemm_acc_n<4>(t1, t2, t3, fa, fb);
    else gemm_acc_n<8>(t1, t2, t3, fa, fb);
}

template <bool EXPLICIT, bool SKEW>
__global__ __launch_bounds__(256) void krylovd_kernel(KrylovArgs args) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double2* a_img = reinterpret_cast<double2*>(smem + A_OFF);
    double2* ah_img = reinterpret_cast<double2*>(smem + AH_OFF);
    double2* xm = reinterpret_cast<double2*>(smem + X_OFF);
    double2* rbase = reinterpret_cast<double2*>(smem + R_OFF);
    double2* ybase = reinterpret_cast<double2*>(smem + Y_OFF);
    auto rm_at = [&](int p) { return rbase + p * (NP * PM); };
    auto ym_at = [&](int p) { return ybase + p * (NP * PM); };
    double* red = reinterpret_cast<double*>(smem + RED_OFF);
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, c = lane & 15, ti = w & 1, tj = w >> 1;
    const int step = args.step0 + blockIdx.x, b = blockIdx.y;
    const int nsteps = args.nsteps, S = args.S, K = args.K;
    const size_t m = (size_t)b * nsteps + step;
    const int sq = step_squarings(args.s_arr[m]);
    const int M = step_order(args.s_arr[m]);
    const double* bt = pade_table(M);
    const double dts = args.dt * ldexp(1.0, -sq);
    const bool tile_on = 16 * tj < S;       // the state tiles of this wave hold states
    const int ks_states = S <= 16 ? 4 : 8;  // k-steps of a product whose inner index runs over the states

    // ---- the scaled generator and its adjoint as column-major images ------------------------
    if constexpr (EXPLICIT) {
        const double2* mm = args.m_rm + m * MAT;  // row-major, padded, unscaled
        const double sc = ldexp(1.0, -sq);
        for (int e = tid; e < MAT; e += 256) {
            const int k = e / NP, r = e % NP;
            const double2 v = mm[(size_t)r * NP + k];   // a[r][k]
            a_img[e] = make_double2(sc * v.x, sc * v.y);
            if constexpr (!SKEW) {
                const double2 t = mm[e];                // a[k][r] -> a^H[r][k] = conj
                ah_img[e] = make_double2(sc * t.x, -sc * t.y);
            }
        }
    } else {
        const StepInterp si = args.interp[step];
        const double* ctl_b = args.controls + (size_t)b * args.nc * K;
        const size_t tsel = (args.nt == 1) ? 0 : (size_t)step;
        const double2* h0r = args.h0_rimg + tsel * MAT;
        const double2* h0t = args.h0_timg + tsel * MAT;
        const double2* gr = args.g_rimg + tsel * K * MAT;
        const double2* gt = args.g_timg + tsel * K * MAT;
        for (int e = tid; e < MAT; e += 256) {
            double2 hv = h0r[e];                         // H[r][k], e = k * NP + r
            double2 tv = SKEW ? make_double2(0, 0) : h0t[e];  // H[k][r]
            for (int kc = 0; kc < K; ++kc) {
                const double uk = control_at(ctl_b, si, K, kc);
                const double2 g = gr[(size_t)kc * MAT + e];
                hv.x += uk * g.x;
                hv.y += uk * g.y;
                if constexpr (!SKEW) {
                    const double2 g2 = gt[(size_t)kc * MAT + e];
                    tv.x += uk * g2.x;
                    tv.y += uk * g2.y;
                }
            }
            a_img[e] = make_double2(dts * hv.y, -dts * hv.x);                   // a = -i dts H
            if constexpr (!SKEW) ah_img[e] = make_double2(dts * tv.y, dts * tv.x);  // conj(-i dts H[k][r])
        }
    }

    const size_t cap = args.slot_cap;
    const double2* states_b = args.states + (size_t)b * cap * S * NP;
    const double2* xs_b = args.xs + (size_t)b * cap * S * NP;
    const int t0 = args.offs[(size_t)b * (nsteps + 1) + step];
    const int nsub = 1 << sq;
    if (t0 < 0 || (size_t)t0 + (size_t)nsub >= cap) return;  // sweep overflowed (status bit 2); block-uniform

    // [slot][s][k] in HBM -> [k][s] in LDS, zero beyond the last state
    auto load_states = [&](const double2* src, double2* mat) {
        for (int e = tid; e < NP * NP; e += 256) {
            const int s = e / NP, k = e % NP;
            mat[k * PM + s] = s < S ? src[(size_t)s * NP + k] : make_double2(0, 0);
        }
    };
    auto store_tile = [&](const Tile& t, double2* mat) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            mat[(16 * ti + 4 * r + q) * PM + 16 * tj + c] = make_double2(t.re[r], t.im[r]);
    };
    auto finish = [&](const d4& t1, const d4& t2, const d4& t3, Tile& out) {
        out.re = t1 - t2;
        out.im = t3 - t1 - t2;
    };
    const d4 zero = {0, 0, 0, 0};
    Tile abar;
    abar.re = zero;
    abar.im = zero;
    for (int sub = 0; sub < nsub; ++sub) {
        const size_t t = (size_t)t0 + sub;
        __syncthreads();  // the images are complete / the previous sub-step has left the matrices
        load_states(xs_b + t * S * NP, xm);
        load_states(states_b + t * S * NP, ym_at(0));
        load_states(states_b + (t + 1) * S * NP, ym_at(1));
        __syncthreads();
        // Sigma and Delta tiles of this wave (rows = vector index, columns = states)
        Tile sg, dl;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int off = (16 * ti + 4 * r + q) * PM + 16 * tj + c;
            const double2 p0 = ym_at(0)[off], p1 = ym_at(1)[off];
            sg.re[r] = p0.x + p1.x;
            sg.im[r] = p0.y + p1.y;
            dl.re[r] = p0.x - p1.x;
            dl.im[r] = p0.y - p1.y;
        }
        Tile rt;  // R_{M-1} = b_M Sigma
        rt.re = bt[M] * sg.re;
        rt.im = bt[M] * sg.im;
        int rp = 0, yp = 0;
        store_tile(rt, rm_at(rp));
        __syncthreads();  // Psi, Psi' have been read (ym is free), R_{M-1} is complete
        Tile y;
        {   // Y_{M-1} = X R_{M-1}^H
            d4 t1 = zero, t2 = zero, t3 = zero;
            const double2* rc = rm_at(rp);
            gemm_acc(t1, t2, t3, ks_states,
                     [&](int kk) { return xm[(16 * ti + c) * PM + 4 * kk + q]; },
                     [&](int kk) {
                         const double2 v = rc[(16 * tj + c) * PM + 4 * kk + q];
                         return make_double2(v.x, -v.y);
                     });
            finish(t1, t2, t3, y);
        }
        for (int i = M - 1; i >= 1; --i) {
            // R_{i-1} = b_i W_i + a R_i ; Y_i to LDS
            store_tile(y, ym_at(yp));
            if (tile_on) {
                d4 t1 = zero, t2 = zero, t3 = zero;
                const double2* rc = rm_at(rp);
                gemm_acc(t1, t2, t3, 8,
                         [&](int kk) { return a_img[(4 * kk + q) * NP + 16 * ti + c]; },
                         [&](int kk) { return rc[(4 * kk + q) * PM + 16 * tj + c]; });
                finish(t1, t2, t3, rt);
                const double coef = bt[i];
                if (i & 1) {
                    rt.re += coef * sg.re;
                    rt.im += coef * sg.im;
                } else {
                    rt.re += coef * dl.re;
                    rt.im += coef * dl.im;
                }
                store_tile(rt, rm_at(rp ^ 1));
            }
            __syncthreads();
            rp ^= 1;
            // Y_{i-1} = X R_{i-1}^H + a^H Y_i
            d4 t1 = zero, t2 = zero, t3 = zero;
            {
                const double2* rc = rm_at(rp);
                gemm_acc(t1, t2, t3, ks_states,
                         [&](int kk) { return xm[(16 * ti + c) * PM + 4 * kk + q]; },
                         [&](int kk) {
                             const double2 v = rc[(16 * tj + c) * PM + 4 * kk + q];
                             return make_double2(v.x, -v.y);
                         });
            }
            {
                const double2* yc = ym_at(yp);
                if constexpr (SKEW) {  // a^H = -a
                    gemm_acc(t1, t2, t3, 8,
                             [&](int kk) {
                                 const double2 v = a_img[(4 * kk + q) * NP + 16 * ti + c];
                                 return make_double2(-v.x, -v.y);
                             },
                             [&](int kk) { return yc[(4 * kk + q) * PM + 16 * tj + c]; });
                } else {
                    gemm_acc(t1, t2, t3, 8,
                             [&](int kk) { return ah_img[(4 * kk + q) * NP + 16 * ti + c]; },
                             [&](int kk) { return yc[(4 * kk + q) * PM + 16 * tj + c]; });
                }
            }
            finish(t1, t2, t3, y);
            yp ^= 1;
        }
        abar.re += y.re;
        abar.im += y.im;
    }
    // this wave's tile of abar: rows 16 ti + 4 r + q, column 16 tj + c
    if constexpr (EXPLICIT) {  // Mbar = 2^-s abar, row-major
        double2* mb = args.mbar_rm + m * MAT;
        const double sc = ldexp(1.0, -sq);
#pragma unroll
        for (int r = 0; r < 4; ++r)
            mb[(size_t)(16 * ti + 4 * r + q) * NP + 16 * tj + c] =
                make_double2(sc * abar.re[r], sc * abar.im[r]);
    } else {
        // g_k = Re <abar, E_k>, E_k = d a / d u_k = -i dts G_k
        const size_t tsel = (args.nt == 1) ? 0 : (size_t)step;
        const double2* gr = args.g_rimg + tsel * K * MAT;
        for (int k0 = 0; k0 < K; k0 += 64) {
            const int kn = min(64, K - k0);
            __syncthreads();
            for (int kc = 0; kc < kn; ++kc) {
                double acc = 0;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double2 e = gr[(size_t)(k0 + kc) * MAT + (16 * tj + c) * NP + 16 * ti + 4 * r + q];
                    acc = fma(abar.im[r], -dts * e.x, fma(abar.re[r], dts * e.y, acc));
                }
                acc = wave_sum(acc);
                if (lane == 0) red[kc * 4 + w] = acc;
            }
            __syncthreads();
            if (tid < kn)
                args.gstep[m * K + k0 + tid] =
                    (red[tid * 4] + red[tid * 4 + 1]) + (red[tid * 4 + 2] + red[tid * 4 + 3]);
        }
    }
}

}  // namespace k3d

}  // namespace sweepd

bool sweepd_supports(int nb, int S) { return nb == 2 && S >= 8 && S <= 32; }

void launch_krylovd(const KrylovArgs& a, int nsteps, int batch, hipStream_t st) {
    using namespace sweepd::k3d;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(krylovd_kernel<true, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(krylovd_kernel<true, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(krylovd_kernel<false, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(krylovd_kernel<false, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        attr_set = true;
    }
    const dim3 grid(nsteps, batch), block(256);
    if (a.m_rm != nullptr && a.skew)
        hipLaunchKernelGGL((krylovd_kernel<true, true>), grid, block, LDS_BYTES, st, a);
    else if (a.m_rm != nullptr)
        hipLaunchKernelGGL((krylovd_kernel<true, false>), grid, block, LDS_BYTES, st, a);
    else if (a.skew)
        hipLaunchKernelGGL((krylovd_kernel<false, true>), grid, block, LDS_BYTES, st, a);
    else
        hipLaunchKernelGGL((krylovd_kernel<false, false>), grid, block, LDS_BYTES, st, a);
}

void launch_sweepd(const SweepArgs& a, int batch, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sweepd::sweepd_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, sweepd::LDS_BYTES);
        attr_set = true;
    }
    hipLaunchKernelGGL(sweepd::sweepd_kernel, dim3(batch), dim3(256), sweepd::LDS_BYTES, st, a);
}

}  // namespace qocx
