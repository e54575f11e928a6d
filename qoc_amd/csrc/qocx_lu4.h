// qocx_lu4.h - K1b for 17 <= n <= 32 with its Schur updates on the matrix cores (round 4).
//
// The one-wave elimination of qocx_lu.h spends 3 400 vector and 500 LDS instructions per matrix,
// a third of them the rank-1 updates (64 multiply-adds per instruction) and the LDS broadcast of
// every pivot row; inside the two-wave K1a that is 44 000 cycles of one in-order wave - more than
// the whole GEMM chain (tools/k1a_stamps.py). Here the matrix stays in the accumulator layout of
// v_mfma_f64_16x16x4_f64 (four 16 x 16 tiles, 64 registers) and is eliminated in eight block steps
// of four pivots:
//
//   * the trailing update A22 -= L21 U12 of a block step is ONE k-step of the MFMA per tile
//     (k = 4 = the block size): four instructions per tile (plain complex product), 64 per matrix,
//     instead of ~1 500 vector multiply-adds;
//   * the four pivots of a block step are eliminated in a "panel" copy: lanes 0..31 hold the rows
//     of the column panel A[:, k0:k0+4], lanes 32..63 the columns of the row panel A[k0:k0+4, :]
//     (the column panel of the transpose), four entries per lane, and BOTH halves run the same
//     instructions - every lane carries the 4 x 4 pivot block (the upper half its transpose) and
//     eliminates it redundantly, so no value ever crosses lanes: the lower half ends with the
//     multipliers L21, the upper half with U12 and with U' = D^-1 U, the form the sweeps read;
//   * pivots are taken on the diagonal SPECULATIVELY: LAPACK's rule (first maximum of |re| + |im|
//     over the unpivoted rows) is checked at every pivot on exactly the numbers the elimination
//     has produced, and the first violation abandons the attempt - the caller then runs the
//     general kernel (qocx_lu.h lu_body) on the untouched LDS image of P. For the Pade
//     denominators of well-scaled generators (P ~ b0 (I - a / 2)) the diagonal always wins.
//
// Same factorisation, different summation order in the updates: factors agree with lu_body to
// rounding, not bit for bit (tests/test_gpu_engine.py::test_pade_factor_kernel runs both).
//
// Measured and not kept (round 4): the factors held in the tiles until the
// last block has passed the check and stored once in 64-byte runs (as lu9_kernel of qocx_lu4m.hip
// does, where a failed check must leave the HBM image of P intact) - the same K1a time, 0.729 ms per
// launch either way; and the products of the pivot-block update formed before the reciprocal they are
// scaled with (two levels off every pivot's dependent chain, twelve more multiplications per
// pivot) - 9 000 cycles MORE per matrix.
#ifndef QOCX_LU4_H
#define QOCX_LU4_H

#include "qocx_lu.h"

namespace qocx {
namespace lu4 {

struct Tiles {  // 32 x 32 complex, tile (ti, tj) in C-layout: lane (q, c), register r = element (16 ti + 4 r + q, 16 tj + c)
    d4 re[2][2], im[2][2];
};

// panel buffer [half][kk][index]; pitch 36 complex per kk so that the 16 lanes of a column-panel
// dump (four columns x four rows, 16 bytes each) fall into 16 different bank groups
constexpr int XP = 36;
constexpr int XB_COMPLEX = 2 * 4 * XP;

struct Cx {
    double re, im;
};
__device__ __forceinline__ Cx cmul(const Cx& a, const Cx& b) {
    return Cx{fma(a.re, b.re, -(a.im * b.im)), fma(a.re, b.im, a.im * b.re)};
}
// a - b * c
__device__ __forceinline__ Cx cfms(const Cx& a, const Cx& b, const Cx& c) {
    return Cx{fma(b.im, c.im, fma(-b.re, c.re, a.re)), fma(-b.im, c.re, fma(-b.re, c.im, a.im))};
}

// One block step J (pivots k0 .. k0 + 3, k0 = 4 J). Returns false (wave-uniform) if a pivot was
// not the diagonal.
template <int J, class Clk>
__device__ __forceinline__ bool block_step(Tiles& T, double2* xb, double2* img, double2* dinv, Clk& clk) {
    constexpr int k0 = 4 * J, t0 = J >> 2, r0 = J & 3, c0 = 4 * (J & 3);
    const int lane = lane_id(), q = lane >> 4, c = lane & 15, half = lane >> 5, idx = lane & 31;

    // ---- the two panels, from the accumulator layout: column panel -> xb[0][kk][row] (the 16
    // lanes that hold those columns), row panel -> xb[1][kk][col] (rows k0 + q are register r0 of
    // every lane)
    if ((c >> 2) == (J & 3)) {
#pragma unroll
        for (int ti = t0; ti < 2; ++ti)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                xb[(c - c0) * XP + 16 * ti + 4 * r + q] = make_double2(T.re[ti][t0][r], T.im[ti][t0][r]);
    }
#pragma unroll
    for (int tj = t0; tj < 2; ++tj)
        xb[(4 + q) * XP + 16 * tj + c] = make_double2(T.re[t0][tj][r0], T.im[t0][tj][r0]);
    wave_sync();

    // ---- this lane's four panel entries and the pivot block (upper half: its transpose - the
    // same address formula in the other panel)
    Cx x[4], dd[4][4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const double2 e = xb[(half * 4 + kk) * XP + idx];
        x[kk] = Cx{e.x, e.y};
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
            const double2 e = xb[(half * 4 + cc) * XP + k0 + r];
            dd[r][cc] = Cx{e.x, e.y};
        }

    clk.lap(5);  // (stamped build: panels out of the tiles and back as panel entries / pivot block)
    // ---- four pivots: the pivot block in every lane, the panel entries of this lane
    Cx f[4], rk[4];
    bool bad = false;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const Cx d = dd[kk][kk];
        // LAPACK izamax on the column as it stands: an unpivoted row strictly larger than the
        // diagonal in |re| + |im| (or a zero diagonal) ends the attempt
        const double magd = fabs(d.re) + fabs(d.im);
        const bool larger = (half == 0) && (idx > k0 + kk) && (fabs(x[kk].re) + fabs(x[kk].im) > magd);
        bad = bad || (__ballot(larger || !(magd > 0.0)) != 0ull);  // (wave-uniform)
        const double rden = fast_rcp(fma(d.re, d.re, d.im * d.im));
        rk[kk] = Cx{d.re * rden, -d.im * rden};
        Cx l[4];
#pragma unroll
        for (int r = kk + 1; r < 4; ++r) l[r] = cmul(dd[r][kk], rk[kk]);
#pragma unroll
        for (int r = kk + 1; r < 4; ++r)
#pragma unroll
            for (int cc = kk + 1; cc < 4; ++cc) dd[r][cc] = cfms(dd[r][cc], l[r], dd[kk][cc]);
        // the panel: rows below the pivot (lower half) / columns right of it (upper half)
        f[kk] = cmul(x[kk], rk[kk]);
        if (idx > k0 + kk) {
#pragma unroll
            for (int t = kk + 1; t < 4; ++t) x[t] = cfms(x[t], f[kk], dd[kk][t]);
        }
    }
    clk.lap(6);  // (stamped build: the four pivots)
    if (bad) return false;

    // ---- the finished entries leave for the image (column-major, rows in their original order:
    // the permutation is the identity). Lower half, column k0 + kk: multipliers below the diagonal,
    // the pivot itself on it; upper half, row k0 + kk: U' = U / U_kk right of the diagonal.
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const bool below = idx > k0 + kk;
        const Cx v = below ? f[kk] : x[kk];
        const int at = half ? (idx * 32 + k0 + kk) : ((k0 + kk) * 32 + idx);
        if (below || (half == 0 && idx == k0 + kk)) img[at] = make_double2(v.re, v.im);
    }
    if (lane == 0) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) dinv[k0 + kk] = make_double2(rk[kk].re, rk[kk].im);
    }

    if constexpr (J < 7) {
        // ---- A22 -= L21 U12 on the tiles that still hold unfinished rows and columns. The MFMA
        // operands come back through the panel buffer: A fragment = L21 (rows x 4, from the lower
        // half), B fragment = U12 (4 x columns, the upper half's entries BEFORE their scaling).
        if (half == 0) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) xb[kk * XP + idx] = make_double2(f[kk].re, f[kk].im);
        } else {
#pragma unroll
            for (int kk = 1; kk < 4; ++kk) xb[(4 + kk) * XP + idx] = make_double2(x[kk].re, x[kk].im);
        }
        wave_sync();
        constexpr int ta = (k0 + 4) >> 4;  // first tile with rows / columns beyond the block
        double are[2], aim[2], nre[2], nim[2], bre[2], bim[2];
#pragma unroll
        for (int t = ta; t < 2; ++t) {
            double2 a = xb[q * XP + 16 * t + c];
            double2 b = xb[(4 + q) * XP + 16 * t + c];
            if (t == t0 && c <= c0 + 3) {  // rows / columns of this block and of earlier ones
                a = make_double2(0.0, 0.0);
                b = make_double2(0.0, 0.0);
            }
            are[t] = a.x; aim[t] = a.y; nre[t] = -a.x; nim[t] = -a.y;
            bre[t] = b.x; bim[t] = b.y;
        }
#pragma unroll
        for (int ti = ta; ti < 2; ++ti)
#pragma unroll
            for (int tj = ta; tj < 2; ++tj) {
                T.re[ti][tj] = mfma_f64(nre[ti], bre[tj], T.re[ti][tj]);
                T.re[ti][tj] = mfma_f64(aim[ti], bim[tj], T.re[ti][tj]);
                T.im[ti][tj] = mfma_f64(nre[ti], bim[tj], T.im[ti][tj]);
                T.im[ti][tj] = mfma_f64(nim[ti], bre[tj], T.im[ti][tj]);
            }
        (void)are;
        wave_sync();  // the fragments have been read before the next block step rewrites the buffer
    }
    clk.lap(4);  // (stamped build: stores, fragments, MFMA issue)
    return true;
}

// One wave factors the 32 x 32 matrix whose column-major image (pitch `pitch` complex per column)
// sits at `src` in LDS; `xb`: XB_COMPLEX complex of LDS owned by this wave. Returns false if the
// diagonal-pivot attempt was abandoned: nothing the caller relies on has been written then (the
// image, 1/U_kk and the permutation are rewritten by the general kernel), `src` is untouched.
template <class Clk>
__device__ __forceinline__ bool lu_mfma_body(const LuArgs& args, size_t m, const double2* src, int pitch,
                                             double2* xb, Clk& clk) {
    const int lane = lane_id(), q = lane >> 4, c = lane & 15;
    Tiles T;
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double2 e = src[(16 * tj + c) * pitch + 16 * ti + 4 * r + q];
                T.re[ti][tj][r] = e.x;
                T.im[ti][tj][r] = e.y;
            }
    double2* img = args.lu_img + m * 1024;
    double2* dinv = args.dinv + m * 32;
    if (!block_step<0>(T, xb, img, dinv, clk)) return false;
    if (!block_step<1>(T, xb, img, dinv, clk)) return false;
    if (!block_step<2>(T, xb, img, dinv, clk)) return false;
    if (!block_step<3>(T, xb, img, dinv, clk)) return false;
    if (!block_step<4>(T, xb, img, dinv, clk)) return false;
    if (!block_step<5>(T, xb, img, dinv, clk)) return false;
    if (!block_step<6>(T, xb, img, dinv, clk)) return false;
    if (!block_step<7>(T, xb, img, dinv, clk)) return false;
    if (lane < 32) {
        args.perm[m * 32 + lane] = lane;
        args.iperm[m * 32 + lane] = lane;
    }
    return true;
}


}  // namespace lu4
}  // namespace qocx

#endif
