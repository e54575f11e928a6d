// qocx_big.hip - K1b (LU) and K3 (Krylov-chain adjoint) for 33 <= n <= 64 (padded to 64) as
// FOUR-wave workgroups.
//
// A 64 x 64 complex matrix is 256 registers per lane of a single wave (lane = row); with the
// second matrix both kernels carry (the multipliers' source and target / the generator and its
// cotangent) the one-wave forms of qocx_kernels.hip spill by the thousand. Here wave w of the
// workgroup owns a quarter of the COLUMNS and every lane still is a row:
//   K1b: columns 16 w .. 16 w + 15. The owner of column k finds the pivot, forms the multipliers
//        and publishes them (plus the pivot's position and reciprocal) through LDS; one workgroup
//        barrier per elimination step; every wave then updates its own columns, taking the pivot
//        row out of its own registers with a dynamic v_readlane; the owner of the NEXT column
//        factors it right after updating it, beside the other waves' updates (lookahead). Same pivot rule (first maximum of
//        |re| + |im|), same storage (original row order, U' = D^-1 U above the diagonal, perm /
//        iperm / 1/U_kk) as lu_kernel.
//   K3 : columns c = 4 cc + w (interleaved, so that the image loads of the four waves stay inside
//        the same cache lines). A matvec is a partial row sum per wave over its 16 columns, summed
//        across the waves through LDS (one barrier, two slots in turn); every wave ends up with
//        the whole vector (lane i = element i), so the NEXT matvec takes its x_c by v_readlane
//        and the thirteen tau_i stay in registers. Same chain as krylov_grad_body.
//
// Reference: the solve of expm_pade (qoc/standard/functions/expm.py:246-249) and the cotangent
// autograd derives for its input (tests/test_device_model.py).
#include "qocx_wave.h"

namespace qocx {

namespace big {

constexpr int NP = 64, WV = 4, CW = 16, MAT = NP * NP;

// ------------------------------------------------------------------------------------------
// K1b
// ------------------------------------------------------------------------------------------
struct LuLds {
    double2 mult[2][NP];  // multipliers of the step, by parity
    double2 rpiv[2];      // reciprocal pivot
    int prow[2];          // lane (= original row) of the pivot
};

template <class F, int... P>
__device__ __forceinline__ void for_each_index(F&& f, std::integer_sequence<int, P...>) {
    (f(std::integral_constant<int, P>()), ...);
}
template <class F>
__device__ __forceinline__ void for_each_step(F&& f) {
    for_each_index(f, std::make_integer_sequence<int, NP>{});
}

__global__ __launch_bounds__(256) void lu4_kernel(LuArgs args) {
    __shared__ __attribute__((aligned(16))) LuLds lds;
    const size_t m = (size_t)(blockIdx.x / args.seg_len) * args.nsteps + args.step0 +
                     blockIdx.x % args.seg_len;
    // behind the MFMA factorisation (qocx_lu4m.hip): only the matrices whose pivots left the diagonal
    if (args.redo != nullptr && (args.redo[m] == 0 || (QOCX_DBG_BITS(args.dbg) & 2))) return;  // (dbg: timing experiment)
    const int lane = lane_id();
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double2* img = args.lu_img + m * MAT;
    double pre[CW], pim[CW];
#pragma unroll
    for (int lc = 0; lc < CW; ++lc) {
        const double2 e = img[(CW * w + lc) * NP + lane];
        pre[lc] = e.x;
        pim[lc] = e.y;
    }
    int mypos = -1;
    bool singular = false;
    double my_dre = 0, my_dim = 0;
    // Column k is factored by its owner wave: pivot search (exact argmax of |re|+|im| over the
    // unpivoted rows as two u32 reductions of the monotonic bit pattern, first maximum wins -
    // LAPACK izamax - with the diagonal fast path of lu_kernel), reciprocal pivot, multipliers,
    // the final column to HBM, and the step's data to LDS (slot k & 1).
    auto factor_column = [&](auto KC) __attribute__((always_inline)) {
        constexpr int k = decltype(KC)::value;
        constexpr int lc = k & 15, par = k & 1;
        const bool mine = (mypos < 0);
        const double mag = fabs(pre[lc]) + fabs(pim[lc]);
        const unsigned long long bits =
            mine ? ((unsigned long long)__double_as_longlong(mag) + 1ull) : 0ull;
        const unsigned khi = (unsigned)(bits >> 32), klo = (unsigned)bits;
        const unsigned long long dbits =
            ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)khi, k) << 32) |
            (unsigned)__builtin_amdgcn_readlane((int)klo, k);
        int lp;
        if (dbits > 1ull && __ballot(bits > dbits) == 0ull) {  // wave-uniform
            lp = k;
        } else {
            const unsigned mh = wave_max_u32(khi);
            unsigned ml = 2u;
            unsigned long long ball = __ballot(khi == mh);
            if (__popcll(ball) > 1 || mh == 0u) {
                ml = wave_max_u32(khi == mh ? klo : 0u);
                ball = __ballot(khi == mh && klo == ml);
            }
            lp = __ffsll((long long)ball) - 1;
            singular = singular || (mh == 0u && ml <= 1u);
        }
        lp = min(max(lp, 0), NP - 1);
        const double pr = readlane_f64(pre[lc], lp), pi = readlane_f64(pim[lc], lp);
        const double rden = fast_rcp(pr * pr + pi * pi);
        const double rre = pr * rden, rim = -pi * rden;
        if (lane == k) args.dinv[m * NP + k] = make_double2(rre, rim);  // 1/U_kk
        const bool elim = mine && (lane != lp);
        const double mre = elim ? (pre[lc] * rre - pim[lc] * rim) : 0.0;
        const double mim = elim ? (pre[lc] * rim + pim[lc] * rre) : 0.0;
        // column k is final: multiplier L_ik (unpivoted rows), U_kk (the new pivot row),
        // U'_ik = U_ik / U_ii (rows pivoted earlier), in ORIGINAL row order
        {
            const bool done = (mypos >= 0);
            const double sre = pre[lc] * my_dre - pim[lc] * my_dim;
            const double sim = pre[lc] * my_dim + pim[lc] * my_dre;
            img[k * NP + lane] = make_double2(elim ? mre : (done ? sre : pre[lc]),
                                              elim ? mim : (done ? sim : pim[lc]));
        }
        lds.mult[par][lane] = make_double2(mre, mim);
        if (lane == 0) {
            lds.rpiv[par] = make_double2(rre, rim);
            lds.prow[par] = lp;
        }
    };
    // a[:, c] -= mult * a[p, c] for one of this wave's columns
    auto update_column = [&](int c, int p, const double2 mu) __attribute__((always_inline)) {
        const double vr = readlane_f64(pre[c], p), vi = readlane_f64(pim[c], p);
        pre[c] = fma(mu.y, vi, fma(-mu.x, vr, pre[c]));
        pim[c] = fma(-mu.y, vr, fma(-mu.x, vi, pim[c]));
    };
    // n <= 48 (the nine-tile K1a wrote the pad block b0 I): rows and columns 48..63 need no
    // elimination step - the pad rows are zero left of the diagonal, so they are never pivots and
    // their multipliers vanish; their columns in HBM are final as they stand.
    const int kmax = (args.n > 0 && args.n <= 48) ? 48 : NP;
    const bool live = __builtin_amdgcn_readfirstlane(CW * w < kmax ? 1 : 0) != 0;  // this wave has columns to update
    if (w == 0) factor_column(std::integral_constant<int, 0>());
    // Step k: one barrier, then every wave applies the step to its columns. LOOKAHEAD: the owner of
    // column k + 1 updates that column first, factors it and publishes step k + 1 (the other LDS
    // slot) BEFORE it updates the rest of its columns - so the serial part of step k + 1 (pivot
    // search, reciprocal, multipliers, ~half a step) runs beside the other waves' updates of step k
    // (measured: 4.06 -> 2.89 ms per 32 000 factorisations).
    for_each_step([&](auto KC) __attribute__((always_inline)) {
        constexpr int k = decltype(KC)::value;
        constexpr int par = k & 1;
        if (k >= kmax) return;  // workgroup-uniform
        __syncthreads();
        const int p = __builtin_amdgcn_readfirstlane(lds.prow[par]);
        {
            const double2 rp = lds.rpiv[par];
            my_dre = (lane == p) ? rp.x : my_dre;
            my_dim = (lane == p) ? rp.y : my_dim;
            mypos = (lane == p) ? k : mypos;
        }
        const double2 mu = lds.mult[par][lane];
        if constexpr (k + 1 < NP) {
            if (k + 1 < kmax && w == ((k + 1) >> 4)) {
                update_column((k + 1) & 15, p, mu);
                factor_column(std::integral_constant<int, k + 1>());
            }
        }
        // (n <= 48: the pivot rows are zero in the pad columns 48..63, nothing to update there)
        if (live && CW * w + CW - 1 > k + 1) {  // wave-uniform: columns beyond k + 1 are still active
#pragma unroll
            for (int c = 0; c < CW; ++c) {
                const bool on = (CW * w + c > k + 1);  // column k + 1 was done above
                double2 mc = mu;
                mc.x = on ? mu.x : 0.0;
                mc.y = on ? mu.y : 0.0;
                update_column(c, p, mc);
            }
        }
        __builtin_amdgcn_sched_barrier(0);  // keep the unrolled steps from interleaving
    });
    if (kmax < NP) {  // the pad rows stay where they are: position = row, 1 / U_kk = 1 / b0
        if (lane >= kmax) {
            mypos = lane;
            if (w == (lane >> 4)) {  // the owner of column `lane` holds the diagonal element b0 of
                double d = 1.0;      // the step's Pade order (qocx_wave.h), untouched by the elimination
#pragma unroll
                for (int c = 0; c < CW; ++c) d = ((lane & 15) == c) ? pre[c] : d;
                args.dinv[m * NP + lane] = make_double2(1.0 / d, 0.0);
            }
        }
    }
    if (singular && lane == 0) atomicOr(args.status, 1);
    if (mypos < 0 || mypos >= NP) {  // only reachable with non-finite input
        mypos = lane;
        atomicOr(args.status, 2);
    }
    if (w == 0) {
        args.perm[m * NP + mypos] = lane;   // row of P that ended at position mypos
        args.iperm[m * NP + lane] = mypos;  // position of row `lane`
    }
}

// ------------------------------------------------------------------------------------------
// K3
// ------------------------------------------------------------------------------------------
struct KrLds {
    double2 part[2][WV][NP];  // partial row sums, by parity
    double red[WV * 64];      // gradient partials: [k][w]
    double red_im[WV * 64];   // unit adjoint: the imaginary parts of gamma
};

// CC: columns per wave that can be non-zero - 16, or 12 for n <= 48 (c = 4 cc + w < 48)
template <bool EXPLICIT, bool SKEW, int CC>
__global__ __launch_bounds__(256) void krylov4_kernel(KrylovArgs args) {
    __shared__ __attribute__((aligned(16))) KrLds lds;
    constexpr int HC = SKEW ? 1 : CC;
    const int step = args.step0 + blockIdx.x, b = blockIdx.y;
    const int lane = lane_id();
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nsteps = args.nsteps, S = args.S, K = args.K;
    const size_t m = (size_t)b * nsteps + step;
    const int sq = step_squarings(args.s_arr[m]);
    // Pade order of the step (qocx_wave.h): chains of M terms; iterations of the unrolled loops
    // beyond M are skipped by branches every wave of the workgroup takes alike
    const int M = step_order(args.s_arr[m]);
    const double* bt = pade_table(M);
    const double dts = args.dt * ldexp(1.0, -sq);
    const StepInterp si = args.interp[step];
    const double* ctl_b = args.controls + (size_t)b * args.nc * K;
    const size_t tsel = (args.nt == 1) ? 0 : (size_t)step;
    const double2* h0r = args.h0_rimg + tsel * MAT;
    const double2* h0t = args.h0_timg + tsel * MAT;
    const double2* gr = args.g_rimg + tsel * K * MAT;
    const double2* gt = args.g_timg + tsel * K * MAT;

    // this wave's columns c = 4 cc + w of a (rows) and of a^H (rows) of the scaled generator
    double are[CC], aim[CC], hre[HC], him[HC];
    if constexpr (EXPLICIT) {
        const double2* mm = args.m_rm + m * MAT;  // row-major, padded
        const double sc = ldexp(1.0, -sq);
#pragma unroll
        for (int cc = 0; cc < CC; ++cc) {
            const int col = WV * cc + w;
            const double2 e = mm[(size_t)lane * NP + col];
            are[cc] = sc * e.x;
            aim[cc] = sc * e.y;
            if constexpr (!SKEW) {
                const double2 f = mm[(size_t)col * NP + lane];
                hre[cc] = sc * f.x;
                him[cc] = -sc * f.y;
            }
        }
    } else {
        double xr[CC], xi[CC], tr_[HC], ti_[HC];
#pragma unroll
        for (int cc = 0; cc < CC; ++cc) {
            const int col = WV * cc + w;
            const double2 e = h0r[col * NP + lane];
            xr[cc] = e.x;
            xi[cc] = e.y;
            if constexpr (!SKEW) {
                const double2 f = h0t[col * NP + lane];
                tr_[cc] = f.x;
                ti_[cc] = f.y;
            }
        }
        for (int k = 0; k < K; ++k) {
            const double uk = control_at(ctl_b, si, K, k);
#pragma unroll
            for (int cc = 0; cc < CC; ++cc) {
                const int col = WV * cc + w;
                const double2 e = gr[(size_t)k * MAT + col * NP + lane];
                xr[cc] += uk * e.x;
                xi[cc] += uk * e.y;
                if constexpr (!SKEW) {
                    const double2 f = gt[(size_t)k * MAT + col * NP + lane];
                    tr_[cc] += uk * f.x;
                    ti_[cc] += uk * f.y;
                }
            }
        }
#pragma unroll
        for (int cc = 0; cc < CC; ++cc) {
            are[cc] = dts * xi[cc];  // a = -i dts H
            aim[cc] = -dts * xr[cc];
            if constexpr (!SKEW) {
                hre[cc] = dts * ti_[cc];  // a^H[i][c] = conj(a[c][i])
                him[cc] = dts * tr_[cc];
            }
        }
    }

    int parity = 0;
    // (re, im) := sum over the four waves, the same value in every wave (fixed order)
    auto xsum = [&](double& re, double& im) __attribute__((always_inline)) {
        lds.part[parity][w][lane] = make_double2(re, im);
        __syncthreads();
        const double2 p0 = lds.part[parity][0][lane], p1 = lds.part[parity][1][lane];
        const double2 p2 = lds.part[parity][2][lane], p3 = lds.part[parity][3][lane];
        re = (p0.x + p1.x) + (p2.x + p3.x);
        im = (p0.y + p1.y) + (p2.y + p3.y);
        parity ^= 1;
    };

    const size_t cap = args.slot_cap;
    const double2* states_b = args.states + (size_t)b * cap * S * NP;
    const double2* xs_b = args.xs + (size_t)b * cap * S * NP;
    const int t0 = args.offs[(size_t)b * (nsteps + 1) + step];
    const int nsub = 1 << sq;
    if (t0 < 0 || (size_t)t0 + (size_t)nsub >= cap) return;  // sweep overflowed (status bit 2)
    // unit adjoint (krylov_grad_body, qocx_kernels.hip): x sits at slots of its own and is the
    // back-propagated target; gstep receives the complex gamma per (step, control)
    const bool unit = args.offs_x != nullptr;
    const int tx = unit ? args.offs_x[(size_t)b * (nsteps + 1) + step] : t0;
    if (tx < 0 || (size_t)tx + (size_t)nsub > cap) return;

    double abr[CC], abi[CC];
#pragma unroll
    for (int cc = 0; cc < CC; ++cc) {
        abr[cc] = 0;
        abi[cc] = 0;
    }
    for (int sub = 0; sub < nsub; ++sub)
        for (int s = 0; s < S; ++s) {
            const size_t t = (size_t)t0 + sub;
            const double2 x = xs_b[(((size_t)tx + sub) * S + s) * NP + lane];
            const double2 p0 = states_b[(t * S + s) * NP + lane];
            const double2 p1 = states_b[((t + 1) * S + s) * NP + lane];
            const double sgr = p0.x + p1.x, sgi = p0.y + p1.y;
            const double dlr = p0.x - p1.x, dli = p0.y - p1.y;
            // Phase A: tau_i = (a^H)^i x, kept in registers (every wave holds all of them)
            double tar[13], tai[13];
            tar[0] = x.x;
            tai[0] = x.y;
#pragma unroll
            for (int jj = 0; jj < 12; ++jj) {
                if (jj + 1 >= M) break;
                double s2r = 0, s2i = 0;
#pragma unroll
                for (int cc = 0; cc < CC; ++cc) {
                    const int col = WV * cc + w;
                    const double vx = readlane_f64(tar[jj], col), vy = readlane_f64(tai[jj], col);
                    if constexpr (SKEW) {  // a^H = -a
                        s2r = fma(aim[cc], vy, fma(-are[cc], vx, s2r));
                        s2i = fma(-aim[cc], vx, fma(-are[cc], vy, s2i));
                    } else {
                        s2r = fma(-him[cc], vy, fma(hre[cc], vx, s2r));
                        s2i = fma(him[cc], vx, fma(hre[cc], vy, s2i));
                    }
                }
                xsum(s2r, s2i);
                tar[jj + 1] = s2r;
                tai[jj + 1] = s2i;
            }
            // Phase B: rho_12 = b13 sigma, rho_{i-1} = b_i w_i + a rho_i (w_i = sigma for odd i,
            // delta for even i); abar += tau_i rho_i^H as each rho_i appears
            double rr = bt[M] * sgr, ri = bt[M] * sgi;
#pragma unroll
            for (int ii = 12; ii >= 0; --ii) {
                if (ii >= M) continue;
                double s0r = 0, s0i = 0;
#pragma unroll
                for (int cc = 0; cc < CC; ++cc) {
                    const int col = WV * cc + w;
                    const double rx = readlane_f64(rr, col), ry = readlane_f64(ri, col);
                    // tau * conj(rho)
                    abr[cc] = fma(tai[ii], ry, fma(tar[ii], rx, abr[cc]));
                    abi[cc] = fma(-tar[ii], ry, fma(tai[ii], rx, abi[cc]));
                    if (ii > 0) {
                        s0r = fma(-aim[cc], ry, fma(are[cc], rx, s0r));
                        s0i = fma(aim[cc], rx, fma(are[cc], ry, s0i));
                    }
                }
                if (ii > 0) {
                    xsum(s0r, s0i);
                    const double coef = bt[ii];
                    rr = fma(coef, (ii & 1) ? sgr : dlr, s0r);
                    ri = fma(coef, (ii & 1) ? sgi : dli, s0i);
                }
            }
        }
    if constexpr (EXPLICIT) {  // Mbar = 2^-s abar (the host or the Magnus reverse kernel finishes)
        double2* mb = args.mbar_rm + m * MAT;
        const double sc = ldexp(1.0, -sq);
#pragma unroll
        for (int cc = 0; cc < CC; ++cc)
            mb[(size_t)lane * NP + WV * cc + w] = make_double2(sc * abr[cc], sc * abi[cc]);
#pragma unroll
        for (int cc = CC; cc < 16; ++cc)  // the pad columns of the cotangent are zero
            mb[(size_t)lane * NP + WV * cc + w] = make_double2(0.0, 0.0);
    } else {
        // g_k = Re <abar, E_k>, E_k = d a / d u_k = -i dts G_k
        for (int k0 = 0; k0 < K; k0 += 64) {
            const int kn = min(64, K - k0);
            for (int k = 0; k < kn; ++k) {
                double acc = 0, acc_im = 0;
#pragma unroll
                for (int cc = 0; cc < CC; ++cc) {
                    const double2 e = gr[(size_t)(k0 + k) * MAT + (WV * cc + w) * NP + lane];
                    acc = fma(abi[cc], -dts * e.x, fma(abr[cc], dts * e.y, acc));
                    acc_im = fma(abi[cc], -dts * e.y, fma(abr[cc], -dts * e.x, acc_im));
                }
                acc = wave_sum(acc);
                if (unit) acc_im = wave_sum(acc_im);
                if (lane == 0) {
                    lds.red[k * WV + w] = acc;
                    lds.red_im[k * WV + w] = acc_im;
                }
            }
            __syncthreads();
            if (w == 0 && lane < kn) {
                const double re = (lds.red[lane * WV] + lds.red[lane * WV + 1]) +
                                  (lds.red[lane * WV + 2] + lds.red[lane * WV + 3]);
                if (unit) {
                    const double im = (lds.red_im[lane * WV] + lds.red_im[lane * WV + 1]) +
                                      (lds.red_im[lane * WV + 2] + lds.red_im[lane * WV + 3]);
                    args.gstep[(m * K + k0 + lane) * 2] = re;
                    args.gstep[(m * K + k0 + lane) * 2 + 1] = im;
                } else {
                    args.gstep[m * K + k0 + lane] = re;
                }
            }
            __syncthreads();
        }
    }
}

}  // namespace big

void launch_lu4(const LuArgs& a, size_t count, hipStream_t st) {
    hipLaunchKernelGGL(big::lu4_kernel, dim3((unsigned)count), dim3(256), 0, st, a);
}

void launch_krylov4(const KrylovArgs& a, int nsteps, int batch, hipStream_t st) {
    const dim3 grid(nsteps, batch), block(256);
    const bool small = a.n > 0 && a.n <= 48;  // the pad columns 48..63 are zero: 12 columns per wave
#define QOCX_K4(E, S)                                                                       \
    do {                                                                                    \
        if (small) hipLaunchKernelGGL((big::krylov4_kernel<E, S, 12>), grid, block, 0, st, a); \
        else hipLaunchKernelGGL((big::krylov4_kernel<E, S, 16>), grid, block, 0, st, a);       \
    } while (0)
    if (a.m_rm != nullptr && a.skew) QOCX_K4(true, true);
    else if (a.m_rm != nullptr) QOCX_K4(true, false);
    else if (a.skew) QOCX_K4(false, true);
    else QOCX_K4(false, false);
#undef QOCX_K4
}

}  // namespace qocx
