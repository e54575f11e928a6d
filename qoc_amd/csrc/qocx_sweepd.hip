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

}  // namespace sweepd

bool sweepd_supports(int nb, int S) { return nb == 2 && S >= 8 && S <= 32; }

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
