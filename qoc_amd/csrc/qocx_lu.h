// qocx_lu.h - K1b, the LU factorisation of the Pade denominator by one wavefront, as a device
// function: used stand-alone (lu_kernel, qocx_kernels.hip) and fused into the two-wave K1a
// (qocx_pade2.hip).
#ifndef QOCX_LU_H
#define QOCX_LU_H

#include "qocx_wave.h"

namespace qocx {

// ------------------------------------------------------------------------------------------
// K1b: LU with partial pivoting, in place on the column-major P image
// ------------------------------------------------------------------------------------------
// Rows are never moved: lane (h,i) keeps row i (columns cc*H+h) and remembers the step at which
// it became the pivot row. The factors are stored in ORIGINAL row order (column k as soon as it
// is final, which frees its registers); perm/iperm give the row order. Pivot choice = first
// maximum of |re|+|im| (LAPACK izamax). Only the pivot row goes through LDS; the pivot element
// comes from a dynamic v_readlane, the multipliers cross lane groups by ds_bpermute.
// value of lane group `g` (static) of a double, replicated to every lane group
template <int NB, int g>
__device__ __forceinline__ double from_group(double v, int i) {
    if (Geo<NB>::H == 2) {
        const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
        auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
        auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        return make_f64((int)a[g], (int)b[g]);
    }
    return __shfl(v, g * Geo<NB>::NP + i);
}

// One wave factors matrix m. `src` holds P column-major with `src_pitch` complex per column -
// the HBM image itself (the stand-alone K1b: src == args.lu_img + m * MAT, pitch NP) or an LDS copy
// (the two-wave K1a with the factorisation fused into it, qocx_pade2.hip: P never travels to HBM
// and back). The factors go to the HBM image args.lu_img + m * MAT either way. `prow`: NP complex
// of LDS owned by this wave.
template <int NB>
__device__ __forceinline__ void lu_body(const LuArgs& args, size_t m, const double2* src,
                                        int src_pitch, double2* prow) {
    typedef Geo<NB> G;
    constexpr int NP = G::NP, CPL = G::CPL, H = G::H;
    const int lane = lane_id(), i = lane % NP, h = lane / NP;
    double2* img = args.lu_img + m * G::MAT;
    double pre[CPL], pim[CPL];
#pragma unroll
    for (int cc = 0; cc < CPL; ++cc) {
        const double2 e = src[(cc * H + h) * src_pitch + i];
        pre[cc] = e.x;
        pim[cc] = e.y;
    }
    int mypos = -1;
    bool singular = false;
    double my_dre = 0, my_dim = 0;
    if (QOCX_DBG_BITS(args.dbg) & 1) {  // timing experiment: the memory traffic of K1b without its arithmetic
#pragma unroll
        for (int cc = 0; cc < CPL; ++cc) img[cc * 64 + lane] = make_double2(pre[cc] + 1.0, pim[cc]);
        if (h == 0) {
            args.perm[m * NP + i] = i;
            args.iperm[m * NP + i] = i;
            args.dinv[m * NP + i] = make_double2(1.0, 0.0);
        }
        return;
    }
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const int hk = k % H, ck = k / H;
        // ---- pivot search: exact argmax of |re|+|im| over the unpivoted rows, as two u32
        // reductions of the (monotonic) bit pattern; first maximum wins (LAPACK izamax).
        const bool mine = (h == hk) && (mypos < 0);
        const double mag = fabs(pre[ck]) + fabs(pim[ck]);
        const unsigned long long bits =
            mine ? ((unsigned long long)__double_as_longlong(mag) + 1ull) : 0ull;
        const unsigned khi = (unsigned)(bits >> 32), klo = (unsigned)bits;
        // Fast path: row k itself is the pivot row whenever it is still unpivoted and no other
        // candidate has a STRICTLY larger |re|+|im| (izamax takes the first maximum, and k is the
        // smallest unpivoted index then). One broadcast, one compare, one ballot instead of the
        // max-reduction; for the Pade denominators of well-scaled generators (P ~ b0 (I - a/2))
        // that is every step. Same pivot, same arithmetic: bit-identical factors.
        const int diag_lane = hk * NP + k;
        const unsigned long long dbits =
            ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)khi, diag_lane) << 32) |
            (unsigned)__builtin_amdgcn_readlane((int)klo, diag_lane);
        int lp;
        if (dbits > 1ull && __ballot(bits > dbits) == 0ull) {  // wave-uniform
            lp = diag_lane;
        } else {
            const unsigned mh = wave_max_u32(khi);
            unsigned ml = 2u;  // only compared against 1 below unless the low words were needed
            unsigned long long ball = __ballot(khi == mh);
            if (__popcll(ball) > 1 || mh == 0u) {  // wave-uniform; rare: the high words tie
                ml = wave_max_u32(khi == mh ? klo : 0u);
                ball = __ballot(khi == mh && klo == ml);
            }
            lp = __ffsll((long long)ball) - 1;  // a lane of group hk, never -1
            singular = singular || (mh == 0u && ml <= 1u);
        }
        const int p = lp % NP;
        // ---- reciprocal pivot, multipliers
        const double pr = readlane_f64(pre[ck], lp), pi = readlane_f64(pim[ck], lp);
        const double rden = fast_rcp(pr * pr + pi * pi);
        const double rre = pr * rden, rim = -pi * rden;
        if (lane == k) args.dinv[m * NP + k] = make_double2(rre, rim);  // 1/U_kk
        const bool elim = mine && (i != p);
        const double mre_own = elim ? (pre[ck] * rre - pim[ck] * rim) : 0.0;
        const double mim_own = elim ? (pre[ck] * rim + pim[ck] * rre) : 0.0;
        // this lane's reciprocal pivot, fixed from the step at which its row becomes the pivot
        // row (the empty asm keeps the compiler from deferring 32 selects to the end)
        my_dre = (i == p) ? rre : my_dre;
        my_dim = (i == p) ? rim : my_dim;
        asm volatile("" : "+v"(my_dre), "+v"(my_dim));
        // column k is final now: store it, in ORIGINAL row order. Unpivoted rows: multiplier
        // L_ik; the new pivot row: U_kk; rows pivoted earlier: U'_ik = U_ik / U_ii, so that
        // both triangular solves of the sweep have a unit diagonal.
        if (h == hk) {
            const bool done = (mypos >= 0);
            const double sre = pre[ck] * my_dre - pim[ck] * my_dim;
            const double sim = pre[ck] * my_dim + pim[ck] * my_dre;
            img[k * NP + i] = make_double2(elim ? mre_own : (done ? sre : pre[ck]),
                                           elim ? mim_own : (done ? sim : pim[ck]));
        }
        double mre = mre_own, mim = mim_own;
        if (H > 1) {
            if (hk == 0) { mre = from_group<NB, 0>(mre_own, i); mim = from_group<NB, 0>(mim_own, i); }
            if (hk == 1) { mre = from_group<NB, 1>(mre_own, i); mim = from_group<NB, 1>(mim_own, i); }
            if (hk == 2) { mre = from_group<NB, 2>(mre_own, i); mim = from_group<NB, 2>(mim_own, i); }
            if (hk == 3) { mre = from_group<NB, 3>(mre_own, i); mim = from_group<NB, 3>(mim_own, i); }
        }
        // ---- pivot row through LDS (only the columns still active), rank-1 update
        if (i == p) {
#pragma unroll
            for (int cc = 0; cc < CPL; ++cc)
                if (cc * H + (H - 1) > k)  // compile time: some lane group still needs it
                    if (cc * H + h > k) prow[cc * H + h] = make_double2(pre[cc], pim[cc]);
        }
        mypos = (i == p) ? k : mypos;
        asm volatile("" : "+v"(mypos));
        wave_sync();
#pragma unroll
        for (int cc = 0; cc < CPL; ++cc) {
            if (cc * H + (H - 1) > k) {
                const double2 pv = prow[cc * H + h];
                const bool on = (cc * H + h > k);
                const double ure = on ? mre : 0.0, uim = on ? mim : 0.0;
                pre[cc] = fma(uim, pv.y, fma(-ure, pv.x, pre[cc]));
                pim[cc] = fma(-uim, pv.x, fma(-ure, pv.y, pim[cc]));
            }
        }
        wave_sync();
        __builtin_amdgcn_sched_barrier(0);  // keep the unrolled steps from interleaving
    }
    if (singular && lane == 0) atomicOr(args.status, 1);
    if (mypos < 0 || mypos >= NP) {  // only reachable with non-finite input
        mypos = i;
        atomicOr(args.status, 2);
    }
    if (h == 0) {
        args.perm[m * NP + mypos] = i;   // row of P that ended at position mypos
        args.iperm[m * NP + i] = mypos;  // position of row i
    }
}

// The INVERSE of P instead of its factors (LuArgs::inverse; the dense-state sweep of
// qocx_sweepd.hip applies P^-1 and Q as MFMA GEMMs to all S states of a seed at once): in-place
// Gauss-Jordan elimination with the same pivot rule and the same data layout as lu_body - rows
// never move, the pivot row of a step goes through LDS, one complex FMA per element and step.
// Step k with pivot row p, r = 1 / A[p][k]: R = r A[p][:], R[k] = r; every other row i takes
// A[i][:] <- (A[i][:] with A[i][k] := 0) - A[i][k] R; row p becomes R. At the end the lane that was
// the pivot row of step t holds row t of (Pi P)^-1; P^-1 = (Pi P)^-1 Pi puts its column c at column
// perm[c]. The image args.lu_img + m * MAT receives P^-1 column-major. `prow`: NP complex of LDS.
template <int NB>
__device__ __forceinline__ void inv_body(const LuArgs& args, size_t m, const double2* src,
                                         int src_pitch, double2* prow) {
    typedef Geo<NB> G;
    constexpr int NP = G::NP, CPL = G::CPL, H = G::H;
    const int lane = lane_id(), i = lane % NP, h = lane / NP;
    double2* img = args.lu_img + m * G::MAT;
    double pre[CPL], pim[CPL];
#pragma unroll
    for (int cc = 0; cc < CPL; ++cc) {
        const double2 e = src[(cc * H + h) * src_pitch + i];
        pre[cc] = e.x;
        pim[cc] = e.y;
    }
    int mypos = -1;
    bool singular = false;
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const int hk = k % H, ck = k / H;
        // ---- pivot search: as lu_body (first maximum of |re| + |im| over the unpivoted rows)
        const bool mine = (h == hk) && (mypos < 0);
        const double mag = fabs(pre[ck]) + fabs(pim[ck]);
        const unsigned long long bits =
            mine ? ((unsigned long long)__double_as_longlong(mag) + 1ull) : 0ull;
        const unsigned khi = (unsigned)(bits >> 32), klo = (unsigned)bits;
        const int diag_lane = hk * NP + k;
        const unsigned long long dbits =
            ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)khi, diag_lane) << 32) |
            (unsigned)__builtin_amdgcn_readlane((int)klo, diag_lane);
        int lp;
        if (dbits > 1ull && __ballot(bits > dbits) == 0ull) {  // wave-uniform
            lp = diag_lane;
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
        const int p = lp % NP;
        const double pr = readlane_f64(pre[ck], lp), pi = readlane_f64(pim[ck], lp);
        const double rden = fast_rcp(pr * pr + pi * pi);
        const double rre = pr * rden, rim = -pi * rden;
        // this row's entry in column k (zero for the pivot row), known to every lane group
        const bool own = (h == hk) && (i != p);
        const double fre_own = own ? pre[ck] : 0.0, fim_own = own ? pim[ck] : 0.0;
        double fre = fre_own, fim = fim_own;
        if (H > 1) {
            if (hk == 0) { fre = from_group<NB, 0>(fre_own, i); fim = from_group<NB, 0>(fim_own, i); }
            if (hk == 1) { fre = from_group<NB, 1>(fre_own, i); fim = from_group<NB, 1>(fim_own, i); }
            if (hk == 2) { fre = from_group<NB, 2>(fre_own, i); fim = from_group<NB, 2>(fim_own, i); }
            if (hk == 3) { fre = from_group<NB, 3>(fre_own, i); fim = from_group<NB, 3>(fim_own, i); }
        }
        // ---- the scaled pivot row through LDS
        if (i == p) {
#pragma unroll
            for (int cc = 0; cc < CPL; ++cc) {
                const bool diag = (cc == ck) && (h == hk);
                const double sre = pre[cc] * rre - pim[cc] * rim, sim = pre[cc] * rim + pim[cc] * rre;
                prow[cc * H + h] = make_double2(diag ? rre : sre, diag ? rim : sim);
            }
        }
        mypos = (i == p) ? k : mypos;
        asm volatile("" : "+v"(mypos));
        wave_sync();
#pragma unroll
        for (int cc = 0; cc < CPL; ++cc) {
            const double2 rv = prow[cc * H + h];
            const bool diag = (cc == ck) && (h == hk);
            const double are = diag ? 0.0 : pre[cc], aim = diag ? 0.0 : pim[cc];
            const double nre = fma(fim, rv.y, fma(-fre, rv.x, are));
            const double nim = fma(-fim, rv.x, fma(-fre, rv.y, aim));
            pre[cc] = (i == p) ? rv.x : nre;
            pim[cc] = (i == p) ? rv.y : nim;
        }
        wave_sync();
        __builtin_amdgcn_sched_barrier(0);  // keep the unrolled steps from interleaving
    }
    if (singular && lane == 0) atomicOr(args.status, 1);
    if (mypos < 0 || mypos >= NP) {  // only reachable with non-finite input
        mypos = i;
        atomicOr(args.status, 2);
    }
    // perm[t] = the row that was the pivot row of step t
    int* ptab = reinterpret_cast<int*>(prow);
    if (h == 0) ptab[mypos] = i;
    wave_sync();
#pragma unroll
    for (int cc = 0; cc < CPL; ++cc) {
        const int col = min(max(ptab[cc * H + h], 0), NP - 1);
        img[(size_t)col * NP + mypos] = make_double2(pre[cc], pim[cc]);
    }
}

template <int NB>
__global__ __launch_bounds__(64) void inv_kernel(LuArgs args) {
    __shared__ __attribute__((aligned(16))) double2 prow[Geo<NB>::NP];
    const size_t m = (size_t)(blockIdx.x / args.seg_len) * args.nsteps + args.step0 +
                     blockIdx.x % args.seg_len;
    inv_body<NB>(args, m, args.lu_img + m * Geo<NB>::MAT, Geo<NB>::NP, prow);
}

template <int NB>
__global__ __launch_bounds__(64) void lu_kernel(LuArgs args) {
    __shared__ __attribute__((aligned(16))) double2 prow[Geo<NB>::NP];
    const size_t m = (size_t)(blockIdx.x / args.seg_len) * args.nsteps + args.step0 +
                     blockIdx.x % args.seg_len;
    lu_body<NB>(args, m, args.lu_img + m * Geo<NB>::MAT, Geo<NB>::NP, prow);
}


}  // namespace qocx

#endif
