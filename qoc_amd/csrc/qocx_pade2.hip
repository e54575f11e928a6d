// qocx_pade2.hip - K1a for 17 <= n <= 32 as a TWO-wave workgroup per propagator step.
//
// The one-wave K1a (qocx_kernels.hip) needs ~410 registers, so a SIMD holds a single wave and
// nothing overlaps its VALU / LDS phases (3M recombination, staging the next A operand, the B
// combinations of the Paterson-Stockmeyer polynomials) with MFMA work: the matrix pipe is busy
// about half of the time (rocprofv3 SQ counters). Here wave w of the workgroup owns
// COLUMN BLOCK w (tiles (0,w), (1,w) of every 32 x 32 matrix, C-layout), which halves the
// registers per wave; two waves of different workgroups then share a SIMD and the hardware
// interleaves one wave's VALU / LDS phases with the other's MFMAs.
//
// A product C = A B needs all of A (from LDS, staged by both waves) and only B(:, w), which is
// the wave's own column block of an earlier product. HERM (Hermitian H: a^2, a^4, a^6, w2, v
// Hermitian, u skew-Hermitian): wave 0 computes tile (0,0) only, wave 1 tiles (0,1), (1,1) and
// hands tile (1,0) = +-conj(tile (0,1))^T to wave 0 through LDS.
//
// LDS (22 KiB per workgroup, so that four workgroups and a sweep wave share a CU): one planar
// A-operand slot holding a, a^2, a^6, w2 in turn, and M = one 16 x 16 complex tile for the
// mirror hand-over of a^4. The Q / P images are stored straight from the C-layout registers.
#include "qocx_wave.h"
#include "qocx_lu.h"
#include "qocx_lu4.h"
#include "qocx_lu5.h"

namespace qocx {

namespace pade2 {

constexpr int PITCH = Geo<2>::PITCH, PLANE = Geo<2>::PLANE, MAT = Geo<2>::MAT;
constexpr int SLOT_F64 = 2 * PLANE;             // re | im planes
constexpr int MPITCH = 18, MTILE_F64 = 2 * 16 * MPITCH;
constexpr int LDS_BYTES = (SLOT_F64 + MTILE_F64 + 6) * 8;  // + six norm words

struct Col {  // tiles (0,w), (1,w) of a complex matrix, C-layout
    d4 re[2], im[2];
};

// 3M accumulators of up to two tiles
struct Acc3 {
    d4 t1[2], t2[2], t3[2];
};

__device__ __forceinline__ void stage_tile(double* slot, int ti, int tj, const d4& re, const d4& im) {
    const int q = lane_id() >> 4, c = lane_id() & 15;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int off = (16 * ti + 4 * r + q) * PITCH + 16 * tj + c;
        slot[off] = re[r];
        slot[PLANE + off] = im[r];
    }
}
// tile (ti, tj) of the slot := sign * conj(src)^T, src being tile (tj, ti)
__device__ __forceinline__ void stage_mirror(double* slot, int ti, int tj, const d4& re, const d4& im,
                                             double sign) {
    const int q = lane_id() >> 4, c = lane_id() & 15;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int off = (16 * ti + c) * PITCH + 16 * tj + 4 * r + q;
        slot[off] = sign * re[r];
        slot[PLANE + off] = -sign * im[r];
    }
}
__device__ __forceinline__ void load_tile(const double* slot, int ti, int tj, d4& re, d4& im) {
    const int q = lane_id() >> 4, c = lane_id() & 15;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int off = (16 * ti + 4 * r + q) * PITCH + 16 * tj + c;
        re[r] = slot[off];
        im[r] = slot[PLANE + off];
    }
}
// the mirror tile M: written transposed-conjugated by wave 1, read in C-layout by wave 0
__device__ __forceinline__ void mirror_put(double* mt, const d4& re, const d4& im, double sign) {
    const int q = lane_id() >> 4, c = lane_id() & 15;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        mt[c * MPITCH + 4 * r + q] = sign * re[r];
        mt[16 * MPITCH + c * MPITCH + 4 * r + q] = -sign * im[r];
    }
}
__device__ __forceinline__ void mirror_get(const double* mt, d4& re, d4& im) {
    const int q = lane_id() >> 4, c = lane_id() & 15;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        re[r] = mt[(4 * r + q) * MPITCH + c];
        im[r] = mt[16 * MPITCH + (4 * r + q) * MPITCH + c];
    }
}

// acc(ti) += A(ti, :) B(:, w) for ti < NT, 3M scheme; A from the slot, B fragment from `bf`
template <int NT, class BFrag>
__device__ __forceinline__ void gemm3(Acc3& acc, const double* slot, BFrag bf) {
    const int q = lane_id() >> 4, c = lane_id() & 15;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
        double are[NT], aim[NT], asum[NT];
#pragma unroll
        for (int ti = 0; ti < NT; ++ti) {
            const int off = (16 * ti + c) * PITCH + 4 * kk + q;
            are[ti] = slot[off];
            aim[ti] = slot[PLANE + off];
            asum[ti] = are[ti] + aim[ti];
        }
        double bre, bim;
        bf(kk, bre, bim);
        const double bsum = bre + bim;
#pragma unroll
        for (int ti = 0; ti < NT; ++ti) {
            acc.t1[ti] = mfma_f64(are[ti], bre, acc.t1[ti]);
            acc.t2[ti] = mfma_f64(aim[ti], bim, acc.t2[ti]);
            acc.t3[ti] = mfma_f64(asum[ti], bsum, acc.t3[ti]);
        }
        // keep the k-steps apart: hoisting every A fragment and B combination to the top of
        // the product costs ~40 registers and spills
        if (kk & 1) __builtin_amdgcn_sched_barrier(0);
    }
}
template <int NT>
__device__ __forceinline__ void acc_zero(Acc3& a) {
#pragma unroll
    for (int ti = 0; ti < NT; ++ti) {
        a.t1[ti] = d4{0, 0, 0, 0};
        a.t2[ti] = d4{0, 0, 0, 0};
        a.t3[ti] = d4{0, 0, 0, 0};
    }
}
template <int NT>
__device__ __forceinline__ void acc_init(Acc3& a, const Col& c) {
#pragma unroll
    for (int ti = 0; ti < NT; ++ti) {
        a.t1[ti] = c.re[ti];
        a.t2[ti] = d4{0, 0, 0, 0};
        a.t3[ti] = c.re[ti] + c.im[ti];
    }
}
template <int NT>
__device__ __forceinline__ void acc_finish(Col& c, const Acc3& a) {
#pragma unroll
    for (int ti = 0; ti < NT; ++ti) {
        c.re[ti] = a.t1[ti] - a.t2[ti];
        c.im[ti] = a.t3[ti] - a.t1[ti] - a.t2[ti];
    }
}

struct Out {
    int pade_policy;  // FactorArgs::pade_policy
    double2* q_img;
    double2* p_img;
    int* s_out;
    int* status;
    // K1b fused in (FactorArgs::fuse_lu): P goes to an LDS image and wave 0 factors it from there
    LuArgs lu;
    bool fuse;
    size_t m;
    bool lu_mfma;                 // the factorisation's Schur updates on the matrix cores (qocx_lu4.h)
    bool lu_dpp;                  // provably diagonal pivots: the factorisation of qocx_lu5.h
    int dbg;                      // FactorArgs::dbg (diagnostic build)
    unsigned long long* stamps;   // FactorArgs::stamps
};

// barrier of the stamped build (qocx_diag.h): the time up to the barrier goes to phase `ph`, the
// wait itself to phase 2
#define K1A_SYNC(ph)          \
    do {                      \
        clk.lap(ph);          \
        __syncthreads();      \
        clk.lap(2);           \
    } while (0)

// LDS image of P for the fused factorisation: column-major, LP complex per column (the C-layout
// stores of a wave walk the columns: pitch 33 spreads them over the banks)
constexpr int LP = 33;
static_assert(32 * LP * 16 <= SLOT_F64 * 8, "the P image reuses the A-operand slot");
static_assert(lu4::XB_COMPLEX * 16 <= MTILE_F64 * 8, "the panel buffer of qocx_lu4.h reuses the mirror tile");

// W: the wave's column block. Every wave executes the same number of barriers.
template <bool HERM, int W, bool STAMP, bool PRESET, class Gen>
__device__ __forceinline__ void body(Gen gen, const Out& out, double* smem) {
    // stamped build: cycles per phase of this wave - 0 generator / norm / staging, 1 products,
    // 2 barrier waits, 3 P / Q images, 4 factorisation
    StampClock<STAMP> clk;
    clk.start();
    // tiles this wave computes: rows [0, NT) of column block W
    constexpr int NT = (HERM && W == 0) ? 1 : 2;
    constexpr bool GIVE = HERM && W == 1;   // hands tile (1,0) to wave 0
    constexpr bool TAKE = HERM && W == 0;
    double* sl = smem;
    double* mt = sl + SLOT_F64;
    double* nrm = mt + MTILE_F64;
    const int lane = lane_id();
    const int q = lane >> 4, c = lane & 15;

    // ---- generator, 1-norm, scaling (expm.py:116, :238-241) -----------------------------
    Col a;
    gen(a, W);
    // The squaring count needs ||a||_1 only up to the interval (theta 2^(s-1), theta 2^s] it
    // falls in. Two bounds without square roots, |z| <= |re| + |im| and |z| >= max(|re|, |im|),
    // usually name the same interval (always when the norm is well below theta, the common
    // case); only if they disagree is the exact norm formed (16 FP64 square roots per lane).
    int sq = 0, order = 13;
    bool dominant = false;  // LAPACK's pivots are provably the diagonal ones (qocx_lu5.h)
    if constexpr (PRESET) {
        // step table (launch_step_table): order and squaring count were decided from the bound
        // dt (||H0||_1 + sum |u_k| ||G_k||_1) when the controls of the step were interpolated
        const int entry = *out.s_out;
        sq = step_squarings(entry);
        order = step_order(entry);
        dominant = step_dominant(entry);
        if (sq > 0) {
            const double scale = ldexp(1.0, -sq);
#pragma unroll
            for (int ti = 0; ti < 2; ++ti) {
                a.re[ti] *= scale;
                a.im[ti] *= scale;
            }
        }
    } else {
    auto squarings = [](double v, bool& bad) {
        double th = QOCX_THETA13;
        int n = 0;
        while (v > th && n < 30) {
            th *= 2.0;
            ++n;
        }
        bad = !(v <= th);  // inf / nan / absurd
        return n;
    };
    auto column_max = [&](double v) {  // max over the wave's 16 columns of the column sums
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        return wave_max(v);
    };
    {
        double bu = 0, bl = 0;
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double ar = fabs(a.re[ti][r]), ai = fabs(a.im[ti][r]);
                bu += ar + ai;
                bl += fmax(ar, ai);
            }
        bu = column_max(bu);
        bl = column_max(bl);
        if (lane == 0) {
            nrm[W] = bu;
            nrm[2 + W] = bl;
        }
    }
    K1A_SYNC(0);  // 1
    bool bad = false, bad_lower = false;
    sq = squarings(fmax(nrm[0], nrm[1]), bad);
    const int sq_lower = squarings(fmax(nrm[2], nrm[3]), bad_lower);
    if (!bad && sq != sq_lower) {  // the same decision in both waves: they read the same numbers
        double e = 0;
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                e += sqrt(a.re[ti][r] * a.re[ti][r] + a.im[ti][r] * a.im[ti][r]);
        e = column_max(e);
        if (lane == 0) nrm[4 + W] = e;
        K1A_SYNC(1);
        sq = squarings(fmax(nrm[4], nrm[5]), bad);
    }
    if (bad) {
        if (W == 0 && lane == 0) atomicOr(out.status, 2);
        sq = 0;
    }
    if (sq > 0) {
        const double scale = ldexp(1.0, -sq);
#pragma unroll
        for (int ti = 0; ti < 2; ++ti) {
            a.re[ti] *= scale;
            a.im[ti] *= scale;
        }
    }
    // Pade order from the upper bound of the norm (qocx_wave.h); both waves read the same numbers
    order = bad ? 13 : pade_order_for(fmax(nrm[0], nrm[1]), out.pade_policy);
    dominant = !bad && pade_denominator_dominant(order, ldexp(fmax(nrm[0], nrm[1]), -sq));
    if (W == 0 && lane == 0) *out.s_out = step_entry(sq, order);
    }
    stage_tile(sl, 0, W, a.re[0], a.im[0]);
    stage_tile(sl, 1, W, a.re[1], a.im[1]);
    K1A_SYNC(0);  // 2

    Col u, v;
    Acc3 acc;
    if (order != 13) {
        // ---- orders 3, 5, 7, 9 (Higham 2005, (10.33); the reference's pade3..pade9 have this
        // shape, expm.py:119-150): x2 = a a, x4 = x2 x2, x6 = x2 x4, x8 = x2 x6 with x2 as the A
        // operand throughout; w = sum b_{2j+1} x_{2j}, v = sum b_{2j} x_{2j} + b0 I, u = w a + b1 a.
        // sq = 0 here (the norm is below theta_9). The finished products leave the registers as
        // soon as they have gone into w and v and served as the next B operand.
        const double* bt = pade_table(order);
        Col w, x;
        acc_zero<NT>(acc);
        gemm3<NT>(acc, sl, [&](int kk, double& bre, double& bim) {
            bre = a.re[kk >> 2][kk & 3];
            bim = a.im[kk >> 2][kk & 3];
        });
        acc_finish<NT>(x, acc);  // x2
#pragma unroll
        for (int ti = 0; ti < NT; ++ti) {
            w.re[ti] = bt[3] * x.re[ti];
            w.im[ti] = bt[3] * x.im[ti];
            v.re[ti] = bt[2] * x.re[ti];
            v.im[ti] = bt[2] * x.im[ti];
            if (ti == W) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (4 * r + q == c) v.re[ti][r] += bt[0];
            }
        }
        K1A_SYNC(1);  // L1: every read of a is done
        if (order >= 5) {
#pragma unroll
            for (int ti = 0; ti < NT; ++ti) stage_tile(sl, ti, W, x.re[ti], x.im[ti]);
            if (GIVE) stage_mirror(sl, 1, 0, x.re[0], x.im[0], 1.0);
            K1A_SYNC(1);  // L2
            if (TAKE) load_tile(sl, 1, 0, x.re[1], x.im[1]);
            for (int j = 2; 2 * j < order; ++j) {  // x_{2j} = x2 x_{2j-2}: j = 2 (x4), 3 (x6), 4 (x8)
                acc_zero<NT>(acc);
                gemm3<NT>(acc, sl, [&](int kk, double& bre, double& bim) {
                    bre = x.re[kk >> 2][kk & 3];
                    bim = x.im[kk >> 2][kk & 3];
                });
                acc_finish<NT>(x, acc);  // (the product is complete: its B operand may go)
                const double bw = bt[2 * j + 1], bv = bt[2 * j];
#pragma unroll
                for (int ti = 0; ti < NT; ++ti) {
                    w.re[ti] += bw * x.re[ti];
                    w.im[ti] += bw * x.im[ti];
                    v.re[ti] += bv * x.re[ti];
                    v.im[ti] += bv * x.im[ti];
                }
                if (2 * (j + 1) < order) {  // the next product takes it as its B operand
                    if (HERM) K1A_SYNC(1);  // the mirror tile's previous content has been read
                    if (GIVE) mirror_put(mt, x.re[0], x.im[0], 1.0);
                    if (HERM) K1A_SYNC(1);
                    if (TAKE) mirror_get(mt, x.re[1], x.im[1]);
                }
            }
            K1A_SYNC(1);  // L3: every read of x2 is done
        }
#pragma unroll
        for (int ti = 0; ti < NT; ++ti) stage_tile(sl, ti, W, w.re[ti], w.im[ti]);
        if (GIVE) stage_mirror(sl, 1, 0, w.re[0], w.im[0], 1.0);
        // (the generator stays in its 32 registers here - fewer products are alive than on the
        // [13/13] path, which rebuilds it from 24 KB of images)
#pragma unroll
        for (int ti = 0; ti < NT; ++ti) {
            u.re[ti] = bt[1] * a.re[ti];
            u.im[ti] = bt[1] * a.im[ti];
        }
        acc_init<NT>(acc, u);
        K1A_SYNC(1);  // L4
        gemm3<NT>(acc, sl, [&](int kk, double& bre, double& bim) {
            bre = a.re[kk >> 2][kk & 3];
            bim = a.im[kk >> 2][kk & 3];
        });
        acc_finish<NT>(u, acc);
    } else {
    // ---- a2 = a a ; a4 = a2 a2 ; a6 = a2 a4 (expm.py:154-156) ----------------------------
    Col x2, x4, x6;
    acc_zero<NT>(acc);
    gemm3<NT>(acc, sl, [&](int kk, double& bre, double& bim) {
        bre = a.re[kk >> 2][kk & 3];
        bim = a.im[kk >> 2][kk & 3];
    });
    acc_finish<NT>(x2, acc);
    K1A_SYNC(1);  // 3: every read of a is done
#pragma unroll
    for (int ti = 0; ti < NT; ++ti) stage_tile(sl, ti, W, x2.re[ti], x2.im[ti]);
    if (GIVE) stage_mirror(sl, 1, 0, x2.re[0], x2.im[0], 1.0);
    K1A_SYNC(1);  // 4
    if (TAKE) load_tile(sl, 1, 0, x2.re[1], x2.im[1]);

    acc_zero<NT>(acc);
    gemm3<NT>(acc, sl, [&](int kk, double& bre, double& bim) {
        bre = x2.re[kk >> 2][kk & 3];
        bim = x2.im[kk >> 2][kk & 3];
    });
    acc_finish<NT>(x4, acc);
    if (GIVE) mirror_put(mt, x4.re[0], x4.im[0], 1.0);
    K1A_SYNC(1);  // 5
    if (TAKE) mirror_get(mt, x4.re[1], x4.im[1]);

    acc_zero<NT>(acc);
    gemm3<NT>(acc, sl, [&](int kk, double& bre, double& bim) {
        bre = x4.re[kk >> 2][kk & 3];
        bim = x4.im[kk >> 2][kk & 3];
    });
    acc_finish<NT>(x6, acc);
    K1A_SYNC(1);  // 6: every read of a2 is done
#pragma unroll
    for (int ti = 0; ti < NT; ++ti) stage_tile(sl, ti, W, x6.re[ti], x6.im[ti]);
    if (GIVE) stage_mirror(sl, 1, 0, x6.re[0], x6.im[0], 1.0);
    K1A_SYNC(1);  // 7
    if (TAKE) load_tile(sl, 1, 0, x6.re[1], x6.im[1]);

    // ---- w2 = a6 (b13 a6 + b11 a4 + b9 a2) + b7 a6 + b5 a4 + b3 a2 (expm.py:157) ---------
    // ---- v  = a6 (b12 a6 + b10 a4 + b8 a2) + b6 a6 + b4 a4 + b2 a2 + b0 I (expm.py:158) --
    const double b0 = PADE_B[0], b1 = PADE_B[1], b2 = PADE_B[2], b3 = PADE_B[3], b4 = PADE_B[4],
                 b5 = PADE_B[5], b6 = PADE_B[6], b7 = PADE_B[7], b8 = PADE_B[8], b9 = PADE_B[9],
                 b10 = PADE_B[10], b11 = PADE_B[11], b12 = PADE_B[12], b13 = PADE_B[13];
    Col w2;
#pragma unroll
    for (int ti = 0; ti < NT; ++ti) {
        w2.re[ti] = b7 * x6.re[ti] + b5 * x4.re[ti] + b3 * x2.re[ti];
        w2.im[ti] = b7 * x6.im[ti] + b5 * x4.im[ti] + b3 * x2.im[ti];
    }
    acc_init<NT>(acc, w2);
    gemm3<NT>(acc, sl, [&](int kk, double& bre, double& bim) {
        const int tb = kk >> 2, r = kk & 3;
        bre = b13 * x6.re[tb][r] + b11 * x4.re[tb][r] + b9 * x2.re[tb][r];
        bim = b13 * x6.im[tb][r] + b11 * x4.im[tb][r] + b9 * x2.im[tb][r];
    });
    acc_finish<NT>(w2, acc);
#pragma unroll
    for (int ti = 0; ti < NT; ++ti) {
        v.re[ti] = b6 * x6.re[ti] + b4 * x4.re[ti] + b2 * x2.re[ti];
        v.im[ti] = b6 * x6.im[ti] + b4 * x4.im[ti] + b2 * x2.im[ti];
        if (ti == W) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (4 * r + q == c) v.re[ti][r] += b0;
        }
    }
    acc_init<NT>(acc, v);
    gemm3<NT>(acc, sl, [&](int kk, double& bre, double& bim) {
        const int tb = kk >> 2, r = kk & 3;
        bre = b12 * x6.re[tb][r] + b10 * x4.re[tb][r] + b8 * x2.re[tb][r];
        bim = b12 * x6.im[tb][r] + b10 * x4.im[tb][r] + b8 * x2.im[tb][r];
    });
    acc_finish<NT>(v, acc);
    K1A_SYNC(1);  // 8: every read of a6 is done
#pragma unroll
    for (int ti = 0; ti < NT; ++ti) stage_tile(sl, ti, W, w2.re[ti], w2.im[ti]);
    if (GIVE) stage_mirror(sl, 1, 0, w2.re[0], w2.im[0], 1.0);

    // ---- u = a w2 + b1 a (expm.py:157), evaluated as w2 a + b1 a: w2 is a polynomial in a, the
    // two commute, and this way the A operand is the product just finished while B is the
    // wave's own column block of the generator (rebuilt: keeping it costs 32 registers).
    gen(a, W);
    if (sq > 0) {
        const double scale = ldexp(1.0, -sq);
#pragma unroll
        for (int ti = 0; ti < 2; ++ti) {
            a.re[ti] *= scale;
            a.im[ti] *= scale;
        }
    }
#pragma unroll
    for (int ti = 0; ti < NT; ++ti) {
        u.re[ti] = b1 * a.re[ti];
        u.im[ti] = b1 * a.im[ti];
    }
    acc_init<NT>(acc, u);
    K1A_SYNC(1);  // 9
    gemm3<NT>(acc, sl, [&](int kk, double& bre, double& bim) {
        bre = a.re[kk >> 2][kk & 3];
        bim = a.im[kk >> 2][kk & 3];
    });
    acc_finish<NT>(u, acc);
    }  // order 13

    // ---- P = v - u ; Q = v + u (expm.py:246), straight from the C-layout registers: for a
    // fixed r the four q-lanes of a column hold rows 4r..4r+3, i.e. one 64-byte run of the
    // column-major image. HERM: Q = P^H, so wave 1 also writes tile (1,0) of each image as the
    // mirror of its tile (0,1) of the other one (256-byte runs).
    //
    // Fused K1b (out.fuse): P does not go to HBM at all. Both waves put their tiles of P
    // into an LDS image (the A-operand slot, free once every wave has finished the last product),
    // wave 1 leaves, and wave 0 runs the one-wave LU (qocx_lu.h) from that image; only the factors
    // travel. The stand-alone K1b moves 32 KB per matrix through HBM (P in, L\U out) and is bound
    // by exactly that: 0.31 ms per 32 000 matrices, 0.21 ms of it with the arithmetic removed
    // (profiles/r03_k1b_memory_floor.jsonl).
    const bool fused = out.fuse;
    double2* simg = reinterpret_cast<double2*>(sl);
    if (fused) K1A_SYNC(1);  // 10: every read of the slot (w2) is done
    else clk.lap(1);
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int idx = (16 * W + c) * 32 + 16 * ti + 4 * r + q;
            if (!QOCX_DBG_BITS(out.q_img == nullptr)) out.q_img[idx] = make_double2(v.re[ti][r] + u.re[ti][r], v.im[ti][r] + u.im[ti][r]);
            const double2 pe = make_double2(v.re[ti][r] - u.re[ti][r], v.im[ti][r] - u.im[ti][r]);
            if (fused) simg[(16 * W + c) * LP + 16 * ti + 4 * r + q] = pe;
            else out.p_img[idx] = pe;
        }
    if (GIVE) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int idx = (4 * r + q) * 32 + 16 + c;  // element (16 + c, 4r + q)
            if (!QOCX_DBG_BITS(out.q_img == nullptr)) out.q_img[idx] = make_double2(v.re[0][r] - u.re[0][r], -(v.im[0][r] - u.im[0][r]));
            const double2 pe = make_double2(v.re[0][r] + u.re[0][r], -(v.im[0][r] + u.im[0][r]));
            if (fused) simg[(4 * r + q) * LP + 16 + c] = pe;
            else out.p_img[idx] = pe;
        }
    }
    if (fused) {
        K1A_SYNC(3);  // 11: the image is complete
        if (W == 0 && !(QOCX_DBG_BITS(out.dbg) & 4)) {
            if (QOCX_DBG_BITS(out.dbg) & 3) {  // (experiment: the factorisation's wave goes first on its SIMD)
                if ((out.dbg & 3) == 1) __builtin_amdgcn_s_setprio(1);
                else if ((out.dbg & 3) == 2) __builtin_amdgcn_s_setprio(2);
                else __builtin_amdgcn_s_setprio(3);
            }
            // diagonal pivots and rank-4 updates on the matrix cores first (qocx_lu4.h); a matrix
            // whose pivots leave the diagonal takes the general elimination, from the same image
            bool done = false;
            if (out.lu_dpp && dominant) {
                lu5::lu_dpp_body(out.lu, out.m, simg, LP);
                done = true;
            } else if (out.lu_mfma) done = lu4::lu_mfma_body(out.lu, out.m, simg, LP, reinterpret_cast<double2*>(mt), clk);
            if (!done) {
                if (out.lu_mfma && out.lu.fallbacks != nullptr && lane_id() == 0) atomicAdd(out.lu.fallbacks, 1);
                lu_body<2>(out.lu, out.m, simg, LP, reinterpret_cast<double2*>(mt));
            }
        }
        clk.lap(4);
    } else {
        clk.lap(3);
    }
    if constexpr (STAMP) {
        clk.acc[7] = __builtin_amdgcn_s_memrealtime() - clk.acc[7];
        if (lane_id() == 0 && out.stamps != nullptr) {
#pragma unroll
            for (int k = 0; k < 8; ++k)  // 1024 sets of sums, or the atomics serialise the whole launch
                atomicAdd(out.stamps + ((blockIdx.x + 7 * blockIdx.y) & 1023) * 16 + W * 8 + k, clk.acc[k]);
        }
    }
}

template <bool HERM, bool STAMP = false, bool PRESET = false>
__global__ __launch_bounds__(128, 2) void pade_pq2_kernel(FactorArgs args) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* smem = reinterpret_cast<double*>(smem_raw);
    const int step = args.step0 + blockIdx.x, b = blockIdx.y;
    const int lane = lane_id();
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t m = (size_t)b * args.nsteps + step;
    if constexpr (PRESET) {
        // the steps at order 3 / 5 belong to the three-wave kernel (qocx_pade3.hip) when it runs too
        if (args.three_wave && step_order(args.s_arr[m]) <= 5) return;
    }
    Out out;
    out.q_img = QOCX_DBG_BITS(args.skip_q) ? nullptr : args.q_img + m * MAT;  // (skip_q: timing experiment)
    out.p_img = args.lu_img + m * MAT;
    out.s_out = args.s_arr + m;
    out.pade_policy = args.pade_policy;
    out.status = args.status;
    out.lu.lu_img = args.lu_img; out.lu.dinv = args.dinv; out.lu.perm = args.perm;
    out.lu.iperm = args.iperm; out.lu.status = args.status; out.lu.nsteps = args.nsteps;
    out.lu.step0 = args.step0; out.lu.seg_len = args.seg_len; out.lu.n = args.n; out.lu.dbg = 0;
    out.lu.fallbacks = args.lu_fallbacks;
    out.fuse = args.fuse_lu != 0;
    out.m = m;
    out.dbg = args.dbg; out.stamps = args.stamps; out.lu_mfma = args.lu_mfma != 0; out.lu_dpp = args.lu_dpp != 0;
    // (PRESET: the step table - controls is [B][nsteps][K], interpolated already)
    const StepInterp si = PRESET ? StepInterp{0, 0, 1.0, 0.0} : args.interp[step];
    const double* ctl_b = args.controls + (size_t)b * args.nc * args.K;
    const size_t tsel = (args.nt == 1) ? 0 : (size_t)step;
    const double2* h0 = args.h0_cimg + tsel * MAT;
    const double2* g = args.g_cimg + tsel * args.K * MAT;
    const double dt = args.dt;
    const int K = args.K;
    auto gen = [&](Col& a, int wcol) {
        // H = h0 + sum_k u_k g_k ; a = dt * (-i H)  (schroedingerdiscrete.py:485-486,
        // mathmethods.py:90-93); C-layout image index ((ti * 2 + tj) * 4 + r) * 64 + lane
        d4 hre[2], him[2];
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double2 e = h0[((ti * 2 + wcol) * 4 + r) * 64 + lane];
                hre[ti][r] = e.x;
                him[ti][r] = e.y;
            }
        for (int k = 0; k < K; ++k) {
            const double uk = PRESET ? ctl_b[(size_t)step * K + k] : control_at(ctl_b, si, K, k);
            const double2* gk = g + (size_t)k * MAT;
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double2 e = gk[((ti * 2 + wcol) * 4 + r) * 64 + lane];
                    hre[ti][r] += uk * e.x;
                    him[ti][r] += uk * e.y;
                }
        }
#pragma unroll
        for (int ti = 0; ti < 2; ++ti) {
            a.re[ti] = dt * him[ti];
            a.im[ti] = -dt * hre[ti];
        }
    };
    if (w == 0) body<HERM, 0, STAMP, PRESET>(gen, out, smem);
    else body<HERM, 1, STAMP, PRESET>(gen, out, smem);
}

// Explicit-generator variant: a[count][n][n] row-major complex in HBM (Magnus M4/M6, debug)
template <bool HERM>
__global__ __launch_bounds__(128, 2) void pade_pq2_explicit_kernel(const double2* a_in, int n,
                                                                   FactorArgs args) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* smem = reinterpret_cast<double*>(smem_raw);
    const size_t m = (size_t)(blockIdx.x / args.seg_len) * args.nsteps + args.step0 +
                     blockIdx.x % args.seg_len;
    const int lane = lane_id();
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q = lane >> 4, c = lane & 15;
    Out out;
    out.q_img = args.q_img + m * MAT;
    out.p_img = args.lu_img + m * MAT;
    out.s_out = args.s_arr + m;
    out.pade_policy = args.pade_policy;
    out.status = args.status;
    out.lu.lu_img = args.lu_img; out.lu.dinv = args.dinv; out.lu.perm = args.perm;
    out.lu.iperm = args.iperm; out.lu.status = args.status; out.lu.nsteps = args.nsteps;
    out.lu.step0 = args.step0; out.lu.seg_len = args.seg_len; out.lu.n = args.n; out.lu.dbg = 0;
    out.lu.fallbacks = args.lu_fallbacks;
    out.fuse = args.fuse_lu != 0;
    out.m = m;
    out.dbg = 0; out.stamps = nullptr; out.lu_mfma = args.lu_mfma != 0; out.lu_dpp = args.lu_dpp != 0;
    const double2* am = a_in + m * (size_t)n * n;
    auto gen = [&](Col& a, int wcol) {
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * ti + 4 * r + q, col = 16 * wcol + c;
                double2 e = make_double2(0, 0);
                if (row < n && col < n) e = am[(size_t)row * n + col];
                a.re[ti][r] = e.x;
                a.im[ti][r] = e.y;
            }
    };
    if (w == 0) body<HERM, 0, false, false>(gen, out, smem);
    else body<HERM, 1, false, false>(gen, out, smem);
}

}  // namespace pade2

void launch_pq2(const FactorArgs& a, int nsteps, int batch, hipStream_t st) {
#ifdef QOCX_DIAG
    if (a.stamps != nullptr && a.hermitian) {  // stamped build (qocx_diag.h)
        if (a.direct)
            hipLaunchKernelGGL((pade2::pade_pq2_kernel<true, true, true>), dim3(nsteps, batch), dim3(128),
                               pade2::LDS_BYTES, st, a);
        else
            hipLaunchKernelGGL((pade2::pade_pq2_kernel<true, true, false>), dim3(nsteps, batch), dim3(128),
                               pade2::LDS_BYTES, st, a);
        return;
    }
#endif
    if (a.hermitian && a.direct)
        hipLaunchKernelGGL((pade2::pade_pq2_kernel<true, false, true>), dim3(nsteps, batch), dim3(128),
                           pade2::LDS_BYTES, st, a);
    else if (a.hermitian)
        hipLaunchKernelGGL((pade2::pade_pq2_kernel<true, false, false>), dim3(nsteps, batch), dim3(128),
                           pade2::LDS_BYTES, st, a);
    else if (a.direct)
        hipLaunchKernelGGL((pade2::pade_pq2_kernel<false, false, true>), dim3(nsteps, batch), dim3(128),
                           pade2::LDS_BYTES, st, a);
    else
        hipLaunchKernelGGL((pade2::pade_pq2_kernel<false, false, false>), dim3(nsteps, batch), dim3(128),
                           pade2::LDS_BYTES, st, a);
}
void launch_pq2_explicit(const double2* a_in, int n, const FactorArgs& a, int count, hipStream_t st) {
    if (a.hermitian)
        hipLaunchKernelGGL(pade2::pade_pq2_explicit_kernel<true>, dim3(count), dim3(128),
                           pade2::LDS_BYTES, st, a_in, n, a);
    else
        hipLaunchKernelGGL(pade2::pade_pq2_explicit_kernel<false>, dim3(count), dim3(128),
                           pade2::LDS_BYTES, st, a_in, n, a);
}

}  // namespace qocx
