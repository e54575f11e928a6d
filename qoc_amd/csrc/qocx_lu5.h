// qocx_lu5.h - K1b for 17 <= n <= 32 on the vector unit with the broadcasts inside the multiply-adds
// (round 5), for Pade denominators whose pivots are PROVABLY the diagonal ones.
//
// When are they? P = b0 (I + E), E = sum_{j>=1} (-1)^j (b_j / b0) a^j, so ||E||_1 <= eps(theta) :=
// sum_{j>=1} (b_j / b0) theta^j for any bound theta >= ||a||_1 (the step table's, or the one K1a forms
// itself). A matrix I + E with column sums sum_i |E_ij| <= eps < 1 is column diagonally dominant, and
// Gaussian elimination keeps, for every column j of every Schur complement S, the margin |s_jj| -
// sum_{i != j} |s_ij| >= 1 - eps and the sum |s_jj| + sum_{i != j} |s_ij| <= 1 + eps (one elimination
// step takes |a_kj| / |a_kk| sum_{i > k} |a_ik| <= |a_kj| from either). So |s_jj| >= 1 - eps and every
// |s_ij| <= eps below the diagonal; LAPACK's izamax (zgetrf behind numpy.linalg.solve,
// qoc/standard/functions/expm.py:246) compares |re| + |im| <= sqrt(2) |.|, and sqrt(2) eps < 1 - eps
// as soon as eps < 1 / (1 + sqrt(2)) = 0.4142: the diagonal entry is the strict maximum of its column
// at every step, zgetrf never exchanges rows, and this kernel - which never looks - makes the same
// choice. pade_denominator_dominant() grants that for eps(theta) <= 0.40 (the margin covers rounding
// in P and in the bound); every other matrix takes the checked factorisations (qocx_lu4.h, qocx_lu.h).
// On the headline workload theta < theta_5 = 0.254, eps <= 0.135: every step.
// tests/test_oracle.py::test_dominant_pade_denominators_pivot_on_the_diagonal holds the claim against
// LAPACK on the CPU, tests/test_gpu_engine.py::test_pade_factor_kernel the factors on the GPU.
//
// The elimination. v_fmac_f64_dpp row_newbcast:k multiplies by the value LANE k of the reader's row of
// 16 lanes holds - a rank-1 update a_ij -= l_ik a_kj is four instructions per column j with the pivot
// row never leaving its lane - but reaches 16 lanes only, so the 32 x 32 matrix is eliminated as 2 x 2
// blocks of 16 and the four rows of 16 lanes each hold a different 16 x 16 block, a matrix row per
// lane, 16 complex registers:
//     lanes  0..15   A11          (row j of it)           32..47   A12        (row j)
//     lanes 16..31   A11^T        (column j of A11)       48..63   A21^T      (column j of A21)
// Sixteen pivots run over all four at once, the SAME instructions: rows of [A11 A12] are eliminated in
// the upper pair, rows of [A11^T A21^T] - the columns of [A11; A21] - in the lower pair (the row
// elimination of A^T yields U'^T and D L^T: the multipliers of the column panel come out scaled by the
// pivot, one multiplication by 1 / U_jj per entry at the end); the multipliers, formed in lanes 0..31
// from the pivot column of A11 / A11^T, are copied to lanes 32..63 (v_permlane32_swap), and the columns
// of A11 / A11^T at or left of the pivot are skipped by the DPP row mask while A12 / A21^T take every
// column. The Schur complement S = A22 - L21 U12 is sixteen MFMAs (operands through LDS), and S - a row
// per lane, every row of 16 lanes a copy - is eliminated by the same code. ~2 000 vector instructions and
// no scalar register in any chain, against 3 900 instructions, eight trips through LDS and eight blocks
// of serial pivots in qocx_lu4.h.
// Same factorisation as LAPACK's up to the order of the additions: factors agree with lu_body to rounding.
#ifndef QOCX_LU5_H
#define QOCX_LU5_H

#include "qocx_lu.h"

namespace qocx {

// eps(theta) <= 0.40 for the [order/order] denominator, theta an upper bound of the 1-norm of the SCALED
// generator (see above)
__device__ __forceinline__ bool pade_denominator_dominant(int order, double theta) {
    const double* b = pade_table(order);
    double eps = 0.0, tp = 1.0;
    const double ib0 = 1.0 / b[0];
    for (int j = 1; j <= order; ++j) {
        tp *= theta;
        eps += b[j] * ib0 * tp;
    }
    return eps <= 0.40;  // (false for NaN)
}

namespace lu5 {

constexpr int SP = 17;  // pitch (complex) of the 16 x 16 dumps in LDS

// x -= l * x[lane BC of the row], within the rows RM of 16 lanes (every lane on)
template <int BC, int RM>
__device__ __forceinline__ void elim(double& xr, double& xi, double lr, double li) {
    asm volatile(
        "v_fmac_f64_dpp %0, -%0, %2 row_newbcast:%4 row_mask:%5 bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %1, -%1, %2 row_newbcast:%4 row_mask:%5 bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %1, %3 row_newbcast:%4 row_mask:%5 bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %1, -%0, %3 row_newbcast:%4 row_mask:%5 bank_mask:0xf"
        : "+v"(xr), "+v"(xi)
        : "v"(lr), "v"(li), "i"(BC), "i"(RM));
}
// the value lane BC of the row holds (two wait states in front of a DPP read of a register the
// vector unit has just written)
template <int BC>
__device__ __forceinline__ void bcast(double& pr, double& pi, double xr, double xi) {
    asm volatile(
        "s_nop 1\n\t"
        "v_mov_b64_dpp %0, %2 row_newbcast:%4 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b64_dpp %1, %3 row_newbcast:%4 row_mask:0xf bank_mask:0xf"
        : "=&v"(pr), "=&v"(pi)
        : "v"(xr), "v"(xi), "i"(BC));
}
// l = x * r on the lanes of the mask (the rows below the pivot), zero elsewhere; the multiplier takes
// the place of the entry it eliminates
template <unsigned LO, unsigned HI>
__device__ __forceinline__ void multiplier(double& lr, double& li, double& xr, double& xi, double rr, double ri) {
    lr = 0.0;
    li = 0.0;
    asm volatile(
        "s_mov_b32 exec_lo, %6\n\t"
        "s_mov_b32 exec_hi, %7\n\t"
        "v_mul_f64 %0, %2, %4\n\t"
        "v_mul_f64 %1, %2, %5\n\t"
        "v_fma_f64 %0, -%3, %5, %0\n\t"
        "v_fma_f64 %1, %3, %4, %1\n\t"
        "v_mov_b64 %2, %0\n\t"
        "v_mov_b64 %3, %1\n\t"
        "s_mov_b64 exec, -1"
        : "+v"(lr), "+v"(li), "+v"(xr), "+v"(xi)
        : "v"(rr), "v"(ri), "i"(LO), "i"(HI));
}
// lanes 0..31 of a value, in lanes 32..63 as well
__device__ __forceinline__ double lower_half_everywhere(double v) {
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return make_f64((int)a[0], (int)b[0]);
}

struct Block {  // a row of a 16 x 16 block per lane
    double re[16], im[16];
};

// The reciprocal of the NEXT pivot, one instruction at a time (asm volatile keeps every step where it
// is written): an in-order wave stalls at a dependent instruction, so the ten steps of the chain - DPP
// read, |p|^2, v_rcp_f64, two Newton steps, r = conj(p) / |p|^2 - go between the column updates of the
// current pivot, which do not depend on them.
struct Recip {
    double pr, pi, t, d, r, e, rr, ri;
};
template <int KN, int STEP>
__device__ __forceinline__ void recip_step(Recip& s, const Block& x) {
    if constexpr (STEP == 0) bcast<KN>(s.pr, s.pi, x.re[KN], x.im[KN]);
    else if constexpr (STEP == 1) asm volatile("v_mul_f64 %0, %1, %1" : "=v"(s.t) : "v"(s.pi));
    else if constexpr (STEP == 2) asm volatile("v_fma_f64 %0, %1, %1, %2" : "=v"(s.d) : "v"(s.pr), "v"(s.t));
    else if constexpr (STEP == 3) asm volatile("v_rcp_f64 %0, %1" : "=v"(s.r) : "v"(s.d));
    else if constexpr (STEP == 4) asm volatile("s_nop 0\n\tv_fma_f64 %0, -%1, %2, 1.0" : "=v"(s.e) : "v"(s.d), "v"(s.r));
    else if constexpr (STEP == 5) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(s.r) : "v"(s.e));
    else if constexpr (STEP == 6) asm volatile("v_fma_f64 %0, -%1, %2, 1.0" : "=v"(s.e) : "v"(s.d), "v"(s.r));
    else if constexpr (STEP == 7) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(s.r) : "v"(s.e));
    else if constexpr (STEP == 8) asm volatile("v_mul_f64 %0, %1, %2" : "=v"(s.rr) : "v"(s.pr), "v"(s.r));
    else if constexpr (STEP == 9) asm volatile("v_mul_f64 %0, -%1, %2" : "=v"(s.ri) : "v"(s.pi), "v"(s.r));
}
constexpr int RECIP_STEPS = 10;
template <int KN, int... STEP>
__device__ __forceinline__ void recip_all(Recip& s, const Block& x, std::integer_sequence<int, STEP...>) {
    (recip_step<KN, STEP>(s, x), ...);
}

// Pivot K of sixteen; s.rr, s.ri: 1 / (pivot K) on entry, 1 / (pivot K + 1) on exit. FOUR: the four
// blocks of the first pass (see the head of the file); else one block, the same in every row of 16
// lanes. myr: this lane's 1 / U_jj once its row has been the pivot row.
template <int K, bool FOUR>
__device__ __forceinline__ void pivot(Block& x, double& myrr, double& myri, int j, Recip& s) {
    const double rr = s.rr, ri = s.ri;
    myrr = (j == K) ? rr : myrr;
    myri = (j == K) ? ri : myri;
    constexpr unsigned p16 = (0xffffu << (K + 1)) & 0xffffu, m32 = p16 | (p16 << 16);
    double lr, li;
    multiplier<m32, FOUR ? 0u : m32>(lr, li, x.re[K], x.im[K], rr, ri);
    if constexpr (FOUR) {
        lr = lower_half_everywhere(lr);
        li = lower_half_everywhere(li);
    }
    // the column right of the pivot first: the next pivot is its diagonal entry
    for_each_const(
        [&](auto C) __attribute__((always_inline)) {
            constexpr int idx = decltype(C)::value, c = (K + 1 + idx) % 16;
            if constexpr (idx >= 1 && idx - 1 < RECIP_STEPS && K + 1 < 16) recip_step<(K + 1) % 16, idx - 1>(s, x);
            if constexpr (c > K) elim<K, 0xf>(x.re[c], x.im[c], lr, li);
            else if constexpr (FOUR) elim<K, 0xc>(x.re[c], x.im[c], lr, li);
        },
        std::make_integer_sequence<int, 16>{});
}
template <bool FOUR, int... K>
__device__ __forceinline__ void pivots(Block& x, double& myrr, double& myri, int j, std::integer_sequence<int, K...>) {
    Recip s;
    recip_all<0>(s, x, std::make_integer_sequence<int, RECIP_STEPS>{});
    (pivot<K, FOUR>(x, myrr, myri, j, s), ...);
}

// One wave factors the 32 x 32 matrix whose column-major image (pitch `pitch` complex per column) sits
// at `src` in LDS; the image is overwritten (it serves as the exchange buffer). Only for matrices
// pade_denominator_dominant() has granted.
// (first half: the four blocks, columns 0..15 and rows 0..15 of the factors, and the Schur complement S
// as an accumulator tile - lane (q, c), register r: S[4 r + q][c])
__device__ __forceinline__ void lu_dpp_first(const LuArgs& args, size_t m, double2* src, int pitch, d4& sre,
                                             d4& sim) {
    const int lane = lane_id(), dr = lane >> 4, j = lane & 15, q = dr, c = j;
    double2* img = args.lu_img + m * 1024;
    double2* dinv = args.dinv + m * 32;
    // ---- the four blocks; A22 as an accumulator tile (lane (q, c), register r: row 16 + 4 r + q, column 16 + c)
    Block x;
    {
        const int base = (dr == 0) ? j : (dr == 2) ? 16 * pitch + j : (dr == 1) ? j * pitch : j * pitch + 16;
        const int stride = (dr & 1) ? 1 : pitch;
#pragma unroll
        for (int cc = 0; cc < 16; ++cc) {
            const double2 e = src[base + cc * stride];
            x.re[cc] = e.x;
            x.im[cc] = e.y;
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const double2 e = src[(16 + c) * pitch + 16 + 4 * r + q];
        sre[r] = e.x;
        sim[r] = e.y;
    }
    double myrr = 0.0, myri = 0.0;
    pivots<true>(x, myrr, myri, j, std::make_integer_sequence<int, 16>{});
    // 1 / U_jj of the lower lanes in the upper ones too (A12 scales like A11, A21^T like A11^T)
    myrr = lower_half_everywhere(myrr);
    myri = lower_half_everywhere(myri);

    // ---- first half of the factors: columns 0..15 (L11 \ U11, L21) and rows 0..15 right of them (U'12)
    // lanes  0..15 (row j, column cc): multiplier | pivot | U' = U / U_jj    -> img[cc][j]
    // lanes 32..47 (row j, column 16 + cc): U'                              -> img[16 + cc][j]
    // lanes 48..63 (row 16 + cc, column j): L21 = (d_j L21) / U_jj          -> img[j][16 + cc]
    wave_sync();  // (every read of the image is done: the dumps below reuse it)
    double2* lds_l = src;               // [k][row of L21]   16 x SP
    double2* lds_u = src + 16 * SP;     // [k][column of U12]
    {
        const int at0 = (dr == 0) ? j : (dr == 2) ? 16 * 32 + j : j * 32 + 16;
        const int step = (dr == 3) ? 1 : 32;
#pragma unroll
        for (int cc = 0; cc < 16; ++cc) {
            const double tr = x.re[cc] * myrr - x.im[cc] * myri, ti = x.re[cc] * myri + x.im[cc] * myrr;
            const bool scaled = (dr >= 2) || (cc > j);
            const double2 v = make_double2(scaled ? tr : x.re[cc], scaled ? ti : x.im[cc]);
            if (dr != 1) img[at0 + cc * step] = v;
            // operands of the Schur update: L21 (true multipliers) and U12 (unscaled), [k = j][index cc]
            if (dr == 3) lds_l[j * SP + cc] = v;
            if (dr == 2) lds_u[j * SP + cc] = make_double2(x.re[cc], x.im[cc]);
        }
    }
    if (dr == 0) dinv[j] = make_double2(myrr, myri);
    wave_sync();
    // ---- S = A22 - L21 U12: A fragment L21[16 + c][4 kk + q], B fragment U12[4 kk + q][16 + c]
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const double2 a = lds_l[(4 * kk + q) * SP + c];
        const double2 b = lds_u[(4 * kk + q) * SP + c];
        sre = mfma_f64(-a.x, b.x, sre);
        sre = mfma_f64(a.y, b.y, sre);
        sim = mfma_f64(-a.x, b.y, sim);
        sim = mfma_f64(-a.y, b.x, sim);
    }
    if (lane < 32) {
        args.perm[m * 32 + lane] = lane;
        args.iperm[m * 32 + lane] = lane;
    }
}

// (second half: S a row per lane - lane j of a row of 16 lanes holds row j - eliminated in every row of 16
// lanes at once, by the same instructions; the rows of lanes that `store` switches on write their factors
// to THEIR matrix `m`: one matrix in four copies (lu_dpp_body) or four matrices (qocx_pade3.hip))
__device__ __forceinline__ void lu_dpp_second(const LuArgs& args, size_t m, Block& x, bool store) {
    const int j = lane_id() & 15;
    double2* img = args.lu_img + m * 1024;
    double2* dinv = args.dinv + m * 32;
    double myrr = 0.0, myri = 0.0;
    pivots<false>(x, myrr, myri, j, std::make_integer_sequence<int, 16>{});
    if (store) {
#pragma unroll
        for (int cc = 0; cc < 16; ++cc) {
            const double tr = x.re[cc] * myrr - x.im[cc] * myri, ti = x.re[cc] * myri + x.im[cc] * myrr;
            const bool scaled = cc > j;
            img[(16 + cc) * 32 + 16 + j] = make_double2(scaled ? tr : x.re[cc], scaled ? ti : x.im[cc]);
        }
        dinv[16 + j] = make_double2(myrr, myri);
    }
}

__device__ __forceinline__ void lu_dpp_body(const LuArgs& args, size_t m, double2* src, int pitch) {
    const int lane = lane_id(), dr = lane >> 4, j = lane & 15, q = dr, c = j;
    d4 sre, sim;
    lu_dpp_first(args, m, src, pitch, sre, sim);
    // ---- S, a row per lane (every row of 16 lanes a copy), and its elimination
    wave_sync();
    double2* lds_s = src + 32 * SP;
#pragma unroll
    for (int r = 0; r < 4; ++r) lds_s[(4 * r + q) * SP + c] = make_double2(sre[r], sim[r]);
    wave_sync();
    Block x;
#pragma unroll
    for (int cc = 0; cc < 16; ++cc) {
        const double2 e = lds_s[j * SP + cc];
        x.re[cc] = e.x;
        x.im[cc] = e.y;
    }
    lu_dpp_second(args, m, x, dr == 0);
}

}  // namespace lu5

// ---- n <= 16: P^-1 of FOUR matrices per wave (round 5) -----------------------------------------------
// The batched evaluator applies P^-1 as a matrix at n <= 16 (qocx_sweepi.hip). inv_kernel (qocx_lu.h)
// spends a wave on ONE 16 x 16 Gauss-Jordan elimination with the pivot row through LDS; here every row
// of 16 lanes holds a different matrix, a matrix row per lane, and one stream of DPP multiply-adds
// eliminates all four: per pivot the multipliers f_j = a_jK / a_KK of every row but the pivot's,
// a_jc -= f_j a_Kc over the fifteen other columns (the pivot row is NOT scaled: it never leaves its
// lane), column K takes the identity part of the augmented matrix ( -f_j, and 1 in the pivot row);
// at the end row j is scaled by 1 / d_j. No pivot search: only for launches whose every matrix is
// diagonally dominant by the margin of pade_denominator_dominant() (the host decides from its bound of
// the step norm, LuArgs::all_dominant); others take inv_kernel.
namespace inv16 {

using lu5::Block;
using lu5::Recip;

// f = x * r on every lane of the rows but lane K of each (zero there); x := -f there, 1 on lane K
template <int K>
__device__ __forceinline__ void multiplier_gj(double& fr, double& fi, double& xr, double& xi, double rr, double ri) {
    constexpr unsigned one = 1u << K, others = (0xffffu & ~one), m_others = others | (others << 16), m_one = one | (one << 16);
    fr = 0.0;
    fi = 0.0;
    asm volatile(
        "s_mov_b32 exec_lo, %6\n\t"
        "s_mov_b32 exec_hi, %6\n\t"
        "v_mul_f64 %0, %2, %4\n\t"
        "v_mul_f64 %1, %2, %5\n\t"
        "v_fma_f64 %0, -%3, %5, %0\n\t"
        "v_fma_f64 %1, %3, %4, %1\n\t"
        "v_mul_f64 %2, %0, -1.0\n\t"
        "v_mul_f64 %3, %1, -1.0\n\t"
        "s_mov_b32 exec_lo, %7\n\t"
        "s_mov_b32 exec_hi, %7\n\t"
        "v_mov_b64 %2, 1.0\n\t"
        "v_mov_b64 %3, 0\n\t"
        "s_mov_b64 exec, -1"
        : "+v"(fr), "+v"(fi), "+v"(xr), "+v"(xi)
        : "v"(rr), "v"(ri), "i"(m_others), "i"(m_one));
}

template <int K>
__device__ __forceinline__ void pivot(Block& x, double& myrr, double& myri, int j, Recip& s) {
    const double rr = s.rr, ri = s.ri;
    myrr = (j == K) ? rr : myrr;
    myri = (j == K) ? ri : myri;
    double fr, fi;
    multiplier_gj<K>(fr, fi, x.re[K], x.im[K], rr, ri);
    // the column right of the pivot first (the next pivot is its diagonal entry), the reciprocal of the
    // next pivot between the others
    for_each_const(
        [&](auto C) __attribute__((always_inline)) {
            constexpr int idx = decltype(C)::value, c = (K + 1 + idx) % 16;
            if constexpr (idx >= 1 && idx - 1 < lu5::RECIP_STEPS && K + 1 < 16) lu5::recip_step<(K + 1) % 16, idx - 1>(s, x);
            if constexpr (c != K) lu5::elim<K, 0xf>(x.re[c], x.im[c], fr, fi);
        },
        std::make_integer_sequence<int, 16>{});
}
template <int... K>
__device__ __forceinline__ void pivots(Block& x, double& myrr, double& myri, int j, std::integer_sequence<int, K...>) {
    Recip s;
    lu5::recip_all<0>(s, x, std::make_integer_sequence<int, lu5::RECIP_STEPS>{});
    (pivot<K>(x, myrr, myri, j, s), ...);
}

// work item w -> matrix (w / seg_len) * nsteps + step0 + w % seg_len; the image (column-major 16 x 16)
// is inverted in place
// PACK8 (LuArgs::pack8, n <= 8): work item w is a PAIR of steps - seed w / pairs, steps step0 + 2 (w %
// pairs) and the next one, pairs = ceil(seg_len / 2) - whose denominators K1a left as the diagonal blocks
// of the tile in the even step's image; the inverse of a block-diagonal matrix is block diagonal, and its
// blocks go out as the two padded 16 x 16 images the sweeps read (pad block: the identity).
template <int NB, bool PACK8>  // (a template, NB = 1 only: the header goes into several translation units)
__global__ __launch_bounds__(64) void inv16_dpp_kernel(LuArgs args, unsigned count) {
    static_assert(NB == 1, "one MFMA tile");
    const int lane = lane_id(), d = lane >> 4, j = lane & 15;
    const unsigned w = min(4u * blockIdx.x + (unsigned)d, count - 1u);
    const bool live = 4u * blockIdx.x + (unsigned)d < count;
    const unsigned pairs = PACK8 ? (unsigned)(args.seg_len + 1) / 2u : 1u;
    const size_t m = PACK8 ? (size_t)(w / pairs) * args.nsteps + args.step0 + 2 * (w % pairs)
                           : (size_t)(w / args.seg_len) * args.nsteps + args.step0 + w % args.seg_len;
    double2* img = args.lu_img + m * 256;
    Block x;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const double2 e = img[c * 16 + j];
        x.re[c] = e.x;
        x.im[c] = e.y;
    }
    double myrr = 0.0, myri = 0.0;
    pivots(x, myrr, myri, j, std::make_integer_sequence<int, 16>{});
    if (live && !PACK8) {
#pragma unroll
        for (int c = 0; c < 16; ++c)
            img[c * 16 + j] = make_double2(x.re[c] * myrr - x.im[c] * myri, x.re[c] * myri + x.im[c] * myrr);
    }
    if (live && PACK8) {
        // lane j holds row j of the tile's inverse: row j % 8 of block j / 8, i.e. of step m + j / 8; it writes
        // that row and the pad row 8 + j % 8 of the step's image
        const int blk = j >> 3, lr = j & 7;
        const bool second = 2 * (int)(w % pairs) + 1 < args.seg_len;
        if (blk == 0 || second) {
            double2* out = args.lu_img + (m + blk) * 256;
            double2 row[8];
#pragma unroll
            for (int cc = 0; cc < 8; ++cc) {  // (blk is lane-dependent: a select per column of the block)
                const double xr = blk ? x.re[8 + cc] : x.re[cc], xi = blk ? x.im[8 + cc] : x.im[cc];
                row[cc] = make_double2(xr * myrr - xi * myri, xr * myri + xi * myrr);
            }
            // (block 0 overwrites the packed tile it has just read - every lane of this row of 16 lanes has
            // its row in registers - block 1 the image of the odd step)
#pragma unroll
            for (int cc = 0; cc < 16; ++cc) {
                out[cc * 16 + lr] = cc < 8 ? row[cc] : make_double2(0.0, 0.0);
                out[cc * 16 + 8 + lr] = make_double2(cc == 8 + lr ? 1.0 : 0.0, 0.0);
            }
        }
    }
}

}  // namespace inv16
}  // namespace qocx

#endif
