// qocx_kernels.hip - hand-written CDNA4 (gfx950) kernels of the GRAPE propagation engine.
//
// One 64-lane wavefront owns one Hilbert-space tile (n <= 32, padded to NP = 16 or 32):
//
//   K1 pade_factor : generator a = -i dt H(u_mid) 2^-s  ->  Pade-13 numerator/denominator
//                    Q = v+u, P = v-u  (6 complex GEMMs on v_mfma_f64_16x16x4_f64, A operand
//                    staged in LDS, B operand / accumulators in registers)  ->  LU(P) with
//                    partial pivoting.        reference: qoc/standard/functions/expm.py:153-159,
//                    :210-246, qoc/core/schroedingerdiscrete.py:483-489, mathmethods.py:36-67,:90-93
//   K2 sweep       : psi_{j+1} = (P^-1 Q)^(2^s) psi_j (serial in j, one wave per seed), state
//                    costs, then lambda_j = Q^H P^-H lambda_{j+1} backwards.
//                    reference: schroedingerdiscrete.py:393-436, expm.py:246-250, costs/*.py
//   K3 krylov_grad : d cost / d u_mid from Krylov chains of a, a^H (the hand-derived adjoint that
//                    replaces autograd's tape: autogradutil.py:26-30); SURVEY.md Appendix A.
//   K4 scatter     : transpose of the linear interpolation (mathmethods.py:33, :54-65).
//
// Register/LDS layouts
//   C-layout  : MFMA accumulator layout. lane = 16*q + c; tile (ti,tj), reg r holds element
//               (row 16 ti + 4 r + q, col 16 tj + c).  A C-layout tile IS the B operand of the
//               next MFMA (k-step kk <-> row block kk>>2, reg kk&3), so GEMM chains need no
//               shuffles; only the A operand goes through LDS.
//   R-layout  : lane = h*NP + i holds row i, columns h*CPL .. h*CPL+CPL-1 (CPL = NP*NP/64).
//               Used by LU, triangular solves and matvecs. Its HBM image stores element
//               (row i, col h*CPL+cc) at complex index cc*64 + lane: every load/store
//               instruction moves one contiguous KiB.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "qocx_device.h"

namespace qocx {

typedef double d4 __attribute__((ext_vector_type(4)));

__device__ __constant__ double PADE_B[14] = {
    64764752532480000.0, 32382376266240000.0, 7771770303897600.0, 1187353796428800.0,
    129060195264000.0,   10559470521600.0,    670442572800.0,     33522128640.0,
    1323241920.0,        40840800.0,          960960.0,           16380.0,
    182.0,               1.0};

#define QOCX_THETA13 5.371920351148152

// ------------------------------------------------------------------------------------------
// wave-level primitives
// ------------------------------------------------------------------------------------------

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

__device__ __forceinline__ double make_f64(int lo, int hi) { return __hiloint2double(hi, lo); }

template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
    return make_f64(lo, hi);
}

__device__ __forceinline__ double readlane_f64(double v, int lane) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return make_f64(lo, hi);
}

// DPP controls (gfx9): quad_perm [1,0,3,2] = 0xB1, [2,3,0,1] = 0x4E, row_half_mirror = 0x141,
// row_mirror = 0x140. Four symmetric exchanges leave every lane of a 16-lane row with the row
// result; the four rows are then combined through SGPRs.
#ifndef QOCX_SAFE_REDUCE
__device__ __forceinline__ double wave_max(double v) {
    v = fmax(v, dpp_f64<0xB1>(v));
    v = fmax(v, dpp_f64<0x4E>(v));
    v = fmax(v, dpp_f64<0x141>(v));
    v = fmax(v, dpp_f64<0x140>(v));
    double r0 = readlane_f64(v, 0), r1 = readlane_f64(v, 16);
    double r2 = readlane_f64(v, 32), r3 = readlane_f64(v, 48);
    return fmax(fmax(r0, r1), fmax(r2, r3));
}
__device__ __forceinline__ double wave_sum(double v) {
    v = v + dpp_f64<0xB1>(v);
    v = v + dpp_f64<0x4E>(v);
    v = v + dpp_f64<0x141>(v);
    v = v + dpp_f64<0x140>(v);
    double r0 = readlane_f64(v, 0), r1 = readlane_f64(v, 16);
    double r2 = readlane_f64(v, 32), r3 = readlane_f64(v, 48);
    return (r0 + r1) + (r2 + r3);
}
#else
__device__ __forceinline__ double wave_max(double v) {
    for (int m = 1; m < 64; m <<= 1) v = fmax(v, __shfl_xor(v, m));
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
    for (int m = 1; m < 64; m <<= 1) v = v + __shfl_xor(v, m);
    return v;
}
#endif

__device__ __forceinline__ d4 mfma_f64(double a, double b, d4 c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// A block is exactly one wavefront: this is an ordering point for its LDS traffic.
__device__ __forceinline__ void wave_sync() { __syncthreads(); }

// ------------------------------------------------------------------------------------------
// layouts
// ------------------------------------------------------------------------------------------

template <int NB>
struct Geo {
    static constexpr int NP = 16 * NB;        // padded Hilbert dimension
    static constexpr int PITCH = NP + 2;      // LDS row pitch (f64) of the planar A-operand slot
    static constexpr int H = 64 / NP;         // lane groups per row in R-layout
    static constexpr int CPL = NP / H;        // columns per lane in R-layout
    static constexpr int MAT = NP * NP;       // complex elements per matrix image
    static constexpr int PLANE = NP * PITCH;  // f64 per LDS plane
    static constexpr int TP = NP + 1;         // pitch (complex) of the LDS transpose buffer
};

template <int NB>
struct CMat {  // C-layout complex matrix in registers
    d4 re[NB][NB];
    d4 im[NB][NB];
};

template <int NB>
__device__ __forceinline__ void cmat_zero(CMat<NB>& m) {
#pragma unroll
    for (int ti = 0; ti < NB; ++ti)
#pragma unroll
        for (int tj = 0; tj < NB; ++tj) {
            m.re[ti][tj] = d4{0, 0, 0, 0};
            m.im[ti][tj] = d4{0, 0, 0, 0};
        }
}

// C-layout registers -> planar LDS slot (row-major, pitch PITCH). 16 consecutive lanes write 16
// consecutive f64: conflict free.
template <int NB>
__device__ __forceinline__ void cmat_to_lds(const CMat<NB>& m, double* lre, double* lim) {
    typedef Geo<NB> G;
    const int q = lane_id() >> 4, c = lane_id() & 15;
#pragma unroll
    for (int ti = 0; ti < NB; ++ti)
#pragma unroll
        for (int tj = 0; tj < NB; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int off = (16 * ti + 4 * r + q) * G::PITCH + 16 * tj + c;
                lre[off] = m.re[ti][tj][r];
                lim[off] = m.im[ti][tj][r];
            }
}

// acc += A * B. A from the planar LDS slot, B given per (k-step, column tile) by `bf`.
// A fragment of v_mfma_f64_16x16x4_f64: lane (q,c) holds A[16 ti + c][4 kk + q]; with
// PITCH = NP + 2 the 32 lanes of a ds_read_b64 group hit 32 distinct bank pairs.
template <int NB, class BFrag>
__device__ __forceinline__ void zgemm_acc(CMat<NB>& acc, const double* lre, const double* lim,
                                          BFrag bf) {
    typedef Geo<NB> G;
    const int q = lane_id() >> 4, c = lane_id() & 15;
#pragma unroll
    for (int kk = 0; kk < 4 * NB; ++kk) {
        double are[NB], aim[NB], nim[NB];
#pragma unroll
        for (int ti = 0; ti < NB; ++ti) {
            const int off = (16 * ti + c) * G::PITCH + 4 * kk + q;
            are[ti] = lre[off];
            aim[ti] = lim[off];
            nim[ti] = -aim[ti];
        }
#pragma unroll
        for (int tj = 0; tj < NB; ++tj) {
            double bre, bim;
            bf(kk, tj, bre, bim);
#pragma unroll
            for (int ti = 0; ti < NB; ++ti) {
                acc.re[ti][tj] = mfma_f64(are[ti], bre, acc.re[ti][tj]);
                acc.re[ti][tj] = mfma_f64(nim[ti], bim, acc.re[ti][tj]);
                acc.im[ti][tj] = mfma_f64(are[ti], bim, acc.im[ti][tj]);
                acc.im[ti][tj] = mfma_f64(aim[ti], bre, acc.im[ti][tj]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// generator assembly (K1 and K3 share it)
// ------------------------------------------------------------------------------------------

// u_k(t_mid): the reference's formula y1 + ((y2 - y1)/(x2 - x1)) * (x3 - x1), mathmethods.py:33.
__device__ __forceinline__ double control_at(const double* ctl_b, const StepInterp& si, int K,
                                              int k) {
    const double y1 = ctl_b[(size_t)si.i1 * K + k];
    const double y2 = ctl_b[(size_t)si.i2 * K + k];
    return y1 + (((y2 - y1) / si.dx) * si.off);
}

// ------------------------------------------------------------------------------------------
// K1: Pade-13 numerator / denominator and LU
// ------------------------------------------------------------------------------------------

struct FactorOut {
    double2* q_img;    // R-image of Q
    double2* lu_img;   // R-image of the row-permuted L\U
    double2* dinv;     // [NP] 1/U_kk
    int* perm;         // [NP] perm[pos] = original row
    int* s_out;        // squarings
    int* status;       // device status word (bit 0: singular pivot, bit 1: bad norm)
};

// LDS carve of K1 (bytes): planar A slot | pivot row | multipliers
template <int NB>
struct FactorLds {
    typedef Geo<NB> G;
    static constexpr int SLOT_BYTES = 2 * G::PLANE * 8;
    static constexpr int PROW_OFF = SLOT_BYTES;
    static constexpr int MULT_OFF = PROW_OFF + G::NP * 16;
    static constexpr int BYTES = MULT_OFF + G::NP * 16;
};

// The body of K1 after the (unscaled) generator has been built by `gen(a)` into C-layout.
template <int NB, class Gen>
__device__ __forceinline__ void pade_factor_body(Gen gen, const FactorOut& out, char* smem) {
    typedef Geo<NB> G;
    constexpr int NP = G::NP, CPL = G::CPL;
    double* lre = reinterpret_cast<double*>(smem);
    double* lim = lre + G::PLANE;
    double2* prow = reinterpret_cast<double2*>(smem + FactorLds<NB>::PROW_OFF);
    double2* mult = reinterpret_cast<double2*>(smem + FactorLds<NB>::MULT_OFF);
    const int lane = lane_id();

    // ---- generator, 1-norm, scaling (expm.py:116, :238-241) -----------------------------
    CMat<NB> a;
    gen(a);
    double colsum[NB];
#pragma unroll
    for (int tj = 0; tj < NB; ++tj) {
        double s = 0;
#pragma unroll
        for (int ti = 0; ti < NB; ++ti)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                s += sqrt(a.re[ti][tj][r] * a.re[ti][tj][r] + a.im[ti][tj][r] * a.im[ti][tj][r]);
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        colsum[tj] = s;
    }
    double norm1 = colsum[0];
#pragma unroll
    for (int tj = 1; tj < NB; ++tj) norm1 = fmax(norm1, colsum[tj]);
    norm1 = wave_max(norm1);
    int sq = 0;
    {
        double th = QOCX_THETA13;
        while (norm1 > th && sq < 30) {
            th *= 2.0;
            ++sq;
        }
        if (!(norm1 <= th)) {  // inf / nan
            if (lane == 0) atomicOr(out.status, 2);
            sq = 0;
        }
    }
    const double scale = ldexp(1.0, -sq);
    if (sq > 0) {
#pragma unroll
        for (int ti = 0; ti < NB; ++ti)
#pragma unroll
            for (int tj = 0; tj < NB; ++tj) {
                a.re[ti][tj] *= scale;
                a.im[ti][tj] *= scale;
            }
    }
    if (lane == 0) *out.s_out = sq;

    // ---- a2 = a a ; a4 = a2 a2 ; a6 = a2 a4 (expm.py:154-156) ----------------------------
    CMat<NB> x2, x4, x6;
    cmat_to_lds<NB>(a, lre, lim);
    wave_sync();
    cmat_zero<NB>(x2);
    zgemm_acc<NB>(x2, lre, lim, [&](int kk, int tj, double& bre, double& bim) {
        bre = a.re[kk >> 2][tj][kk & 3];
        bim = a.im[kk >> 2][tj][kk & 3];
    });
    wave_sync();
    cmat_to_lds<NB>(x2, lre, lim);
    wave_sync();
    cmat_zero<NB>(x4);
    zgemm_acc<NB>(x4, lre, lim, [&](int kk, int tj, double& bre, double& bim) {
        bre = x2.re[kk >> 2][tj][kk & 3];
        bim = x2.im[kk >> 2][tj][kk & 3];
    });
    cmat_zero<NB>(x6);
    zgemm_acc<NB>(x6, lre, lim, [&](int kk, int tj, double& bre, double& bim) {
        bre = x4.re[kk >> 2][tj][kk & 3];
        bim = x4.im[kk >> 2][tj][kk & 3];
    });
    wave_sync();

    // ---- w2 = a6 (b13 a6 + b11 a4 + b9 a2) + b7 a6 + b5 a4 + b3 a2 (expm.py:157) ---------
    // ---- v  = a6 (b12 a6 + b10 a4 + b8 a2) + b6 a6 + b4 a4 + b2 a2 + b0 I (expm.py:158) --
    cmat_to_lds<NB>(x6, lre, lim);
    wave_sync();
    const double b0 = PADE_B[0], b1 = PADE_B[1], b2 = PADE_B[2], b3 = PADE_B[3], b4 = PADE_B[4],
                 b5 = PADE_B[5], b6 = PADE_B[6], b7 = PADE_B[7], b8 = PADE_B[8], b9 = PADE_B[9],
                 b10 = PADE_B[10], b11 = PADE_B[11], b12 = PADE_B[12], b13 = PADE_B[13];
    CMat<NB> w2, v;
    const int q = lane >> 4, c = lane & 15;
#pragma unroll
    for (int ti = 0; ti < NB; ++ti)
#pragma unroll
        for (int tj = 0; tj < NB; ++tj) {
            w2.re[ti][tj] = b7 * x6.re[ti][tj] + b5 * x4.re[ti][tj] + b3 * x2.re[ti][tj];
            w2.im[ti][tj] = b7 * x6.im[ti][tj] + b5 * x4.im[ti][tj] + b3 * x2.im[ti][tj];
            v.re[ti][tj] = b6 * x6.re[ti][tj] + b4 * x4.re[ti][tj] + b2 * x2.re[ti][tj];
            v.im[ti][tj] = b6 * x6.im[ti][tj] + b4 * x4.im[ti][tj] + b2 * x2.im[ti][tj];
            if (ti == tj) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (4 * r + q == c) v.re[ti][tj][r] += b0;
            }
        }
    zgemm_acc<NB>(w2, lre, lim, [&](int kk, int tj, double& bre, double& bim) {
        const int tb = kk >> 2, r = kk & 3;
        bre = b13 * x6.re[tb][tj][r] + b11 * x4.re[tb][tj][r] + b9 * x2.re[tb][tj][r];
        bim = b13 * x6.im[tb][tj][r] + b11 * x4.im[tb][tj][r] + b9 * x2.im[tb][tj][r];
    });
    zgemm_acc<NB>(v, lre, lim, [&](int kk, int tj, double& bre, double& bim) {
        const int tb = kk >> 2, r = kk & 3;
        bre = b12 * x6.re[tb][tj][r] + b10 * x4.re[tb][tj][r] + b8 * x2.re[tb][tj][r];
        bim = b12 * x6.im[tb][tj][r] + b10 * x4.im[tb][tj][r] + b8 * x2.im[tb][tj][r];
    });
    wave_sync();

    // ---- u = a w2 + b1 a (expm.py:157) ; P = v - u ; Q = v + u (expm.py:246) ---------------
    gen(a);
    if (sq > 0) {
#pragma unroll
        for (int ti = 0; ti < NB; ++ti)
#pragma unroll
            for (int tj = 0; tj < NB; ++tj) {
                a.re[ti][tj] *= scale;
                a.im[ti][tj] *= scale;
            }
    }
    cmat_to_lds<NB>(a, lre, lim);
    wave_sync();
    CMat<NB> u;
#pragma unroll
    for (int ti = 0; ti < NB; ++ti)
#pragma unroll
        for (int tj = 0; tj < NB; ++tj) {
            u.re[ti][tj] = b1 * a.re[ti][tj];
            u.im[ti][tj] = b1 * a.im[ti][tj];
        }
    zgemm_acc<NB>(u, lre, lim, [&](int kk, int tj, double& bre, double& bim) {
        bre = w2.re[kk >> 2][tj][kk & 3];
        bim = w2.im[kk >> 2][tj][kk & 3];
    });
    wave_sync();

    const int i = lane % NP, h = lane / NP;
    // Q: C-layout -> LDS -> R-layout -> HBM image (one contiguous KiB per store instruction)
    {
        CMat<NB> t;
#pragma unroll
        for (int ti = 0; ti < NB; ++ti)
#pragma unroll
            for (int tj = 0; tj < NB; ++tj) {
                t.re[ti][tj] = v.re[ti][tj] + u.re[ti][tj];
                t.im[ti][tj] = v.im[ti][tj] + u.im[ti][tj];
            }
        cmat_to_lds<NB>(t, lre, lim);
        wave_sync();
#pragma unroll
        for (int cc = 0; cc < CPL; ++cc) {
            const int off = i * G::PITCH + h * CPL + cc;
            out.q_img[cc * 64 + lane] = make_double2(lre[off], lim[off]);
        }
        wave_sync();
#pragma unroll
        for (int ti = 0; ti < NB; ++ti)
#pragma unroll
            for (int tj = 0; tj < NB; ++tj) {
                t.re[ti][tj] = v.re[ti][tj] - u.re[ti][tj];
                t.im[ti][tj] = v.im[ti][tj] - u.im[ti][tj];
            }
        cmat_to_lds<NB>(t, lre, lim);
        wave_sync();
    }

    // ---- LU with partial pivoting in R-layout (numpy.linalg.solve == LAPACK zgesv, expm.py:246)
    // Rows are never moved: lane (h,i) keeps row i and remembers the step at which it became
    // the pivot row; the image is written row-permuted at the end. Pivot choice = first
    // maximum of |re|+|im| (LAPACK izamax).
    double pre[CPL], pim[CPL];
#pragma unroll
    for (int cc = 0; cc < CPL; ++cc) {
        const int off = i * G::PITCH + h * CPL + cc;
        pre[cc] = lre[off];
        pim[cc] = lim[off];
    }
    int mypos = -1;
    double dinv_re = 0, dinv_im = 0;
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const int hk = k / CPL, ck = k % CPL;
        const bool mine = (h == hk) && (mypos < 0);
        const double cand = mine ? (fabs(pre[ck]) + fabs(pim[ck])) : -1.0;
        const double mx = wave_max(cand);
        const unsigned long long ball = __ballot(cand == mx);
        const int p = (__ffsll((long long)ball) - 1) % NP;
        if (!(mx > 0.0) && lane == 0) atomicOr(out.status, 1);
        if (i == p) {
#pragma unroll
            for (int cc = 0; cc < CPL; ++cc) prow[h * CPL + cc] = make_double2(pre[cc], pim[cc]);
        }
        wave_sync();
        const double2 piv = prow[k];
        const double den = piv.x * piv.x + piv.y * piv.y;
        const double rden = 1.0 / den;
        const double rre = piv.x * rden, rim = -piv.y * rden;
        if (lane == k) {
            dinv_re = rre;
            dinv_im = rim;
        }
        if (h == hk) {
            double mre = 0, mim = 0;
            if (mypos < 0 && i != p) {
                mre = pre[ck] * rre - pim[ck] * rim;
                mim = pre[ck] * rim + pim[ck] * rre;
                pre[ck] = mre;
                pim[ck] = mim;
            }
            mult[i] = make_double2(mre, mim);
        }
        wave_sync();
        const double2 m = mult[i];
#pragma unroll
        for (int cc = 0; cc < CPL; ++cc) {
            if (h * CPL + cc > k) {
                const double2 pv = prow[h * CPL + cc];
                pre[cc] -= m.x * pv.x - m.y * pv.y;
                pim[cc] -= m.x * pv.y + m.y * pv.x;
            }
        }
        if (i == p) mypos = k;
        wave_sync();
    }
    if (mypos < 0 || mypos >= NP) {  // only reachable with non-finite input
        mypos = i;
        atomicOr(out.status, 2);
    }
#pragma unroll
    for (int cc = 0; cc < CPL; ++cc)
        out.lu_img[cc * 64 + h * NP + mypos] = make_double2(pre[cc], pim[cc]);
    if (h == 0) out.perm[mypos] = i;
    if (lane < NP) out.dinv[lane] = make_double2(dinv_re, dinv_im);
}

template <int NB>
__global__ __launch_bounds__(64) void pade_factor_kernel(FactorArgs args) {
    typedef Geo<NB> G;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int step = blockIdx.x, b = blockIdx.y;
    const int lane = lane_id();
    const size_t m = (size_t)b * args.nsteps + step;
    FactorOut out;
    out.q_img = args.q_img + m * G::MAT;
    out.lu_img = args.lu_img + m * G::MAT;
    out.dinv = args.dinv + m * G::NP;
    out.perm = args.perm + m * G::NP;
    out.s_out = args.s_arr + m;
    out.status = args.status;
    const StepInterp si = args.interp[step];
    const double* ctl_b = args.controls + (size_t)b * args.nc * args.K;
    const size_t tsel = (args.nt == 1) ? 0 : (size_t)step;
    const double2* h0 = args.h0_cimg + tsel * G::MAT;
    const double2* g = args.g_cimg + tsel * args.K * G::MAT;
    const double dt = args.dt;
    const int K = args.K;
    auto gen = [&](CMat<NB>& a) {
        // H = h0 + sum_k u_k g_k ; a = dt * (-i H)  (schroedingerdiscrete.py:485-486,
        // mathmethods.py:90-93)
        CMat<NB> hm;
#pragma unroll
        for (int ti = 0; ti < NB; ++ti)
#pragma unroll
            for (int tj = 0; tj < NB; ++tj)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double2 e = h0[((ti * NB + tj) * 4 + r) * 64 + lane];
                    hm.re[ti][tj][r] = e.x;
                    hm.im[ti][tj][r] = e.y;
                }
        for (int k = 0; k < K; ++k) {
            const double uk = control_at(ctl_b, si, K, k);
            const double2* gk = g + (size_t)k * G::MAT;
#pragma unroll
            for (int ti = 0; ti < NB; ++ti)
#pragma unroll
                for (int tj = 0; tj < NB; ++tj)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double2 e = gk[((ti * NB + tj) * 4 + r) * 64 + lane];
                        hm.re[ti][tj][r] += uk * e.x;
                        hm.im[ti][tj][r] += uk * e.y;
                    }
        }
#pragma unroll
        for (int ti = 0; ti < NB; ++ti)
#pragma unroll
            for (int tj = 0; tj < NB; ++tj) {
                a.re[ti][tj] = dt * hm.im[ti][tj];
                a.im[ti][tj] = -dt * hm.re[ti][tj];
            }
    };
    pade_factor_body<NB>(gen, out, smem);
}

// Debug / explicit-generator variant: a[count][n][n] row-major complex in HBM.
template <int NB>
__global__ __launch_bounds__(64) void pade_factor_explicit_kernel(const double2* a_in, int n,
                                                                  FactorArgs args) {
    typedef Geo<NB> G;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const size_t m = blockIdx.x;
    const int lane = lane_id();
    const int q = lane >> 4, c = lane & 15;
    FactorOut out;
    out.q_img = args.q_img + m * G::MAT;
    out.lu_img = args.lu_img + m * G::MAT;
    out.dinv = args.dinv + m * G::NP;
    out.perm = args.perm + m * G::NP;
    out.s_out = args.s_arr + m;
    out.status = args.status;
    const double2* am = a_in + m * (size_t)n * n;
    auto gen = [&](CMat<NB>& a) {
#pragma unroll
        for (int ti = 0; ti < NB; ++ti)
#pragma unroll
            for (int tj = 0; tj < NB; ++tj)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * ti + 4 * r + q, col = 16 * tj + c;
                    double2 e = make_double2(0, 0);
                    if (row < n && col < n) e = am[(size_t)row * n + col];
                    a.re[ti][tj][r] = e.x;
                    a.im[ti][tj][r] = e.y;
                }
    };
    pade_factor_body<NB>(gen, out, smem);
}

// ------------------------------------------------------------------------------------------
// K2: serial state sweep (forward), costs, adjoint sweep (backward)
// ------------------------------------------------------------------------------------------

// Triangular solves in R-layout, axpy form. z is replicated in every lane group h; in phase hk
// only group hk (which holds columns hk*CPL..) updates, then its copy is broadcast.
// LOWER: forward substitution (row k known before rows > k). UNIT: unit diagonal, otherwise the
// lane's diagonal reciprocal d is applied. CONJ: use conj of the stored coefficients.
template <int NB, bool LOWER, bool UNIT, bool CONJ>
__device__ __forceinline__ void tri_solve(const double (&tre)[Geo<NB>::CPL],
                                          const double (&tim)[Geo<NB>::CPL], double& zre,
                                          double& zim, double dre, double dim, int h, int i) {
    typedef Geo<NB> G;
    constexpr int NP = G::NP, CPL = G::CPL, H = G::H;
    if (CONJ) dim = -dim;
#pragma unroll
    for (int hh = 0; hh < H; ++hh) {
        const int hk = LOWER ? hh : (H - 1 - hh);
#pragma unroll
        for (int cs = 0; cs < CPL; ++cs) {
            const int ck = LOWER ? cs : (CPL - 1 - cs);
            const int k = hk * CPL + ck;
            double vre = zre, vim = zim;
            if (!UNIT) {
                vre = zre * dre - zim * dim;
                vim = zre * dim + zim * dre;
            }
            const double kre = readlane_f64(vre, hk * NP + k);
            const double kim = readlane_f64(vim, hk * NP + k);
            if (!UNIT) {
                if (i == k) {
                    zre = kre;
                    zim = kim;
                }
            }
            const double lre = tre[ck];
            const double lim = CONJ ? -tim[ck] : tim[ck];
            const bool upd = (h == hk) && (LOWER ? (i > k) : (i < k));
            if (upd) {
                zre -= lre * kre - lim * kim;
                zim -= lre * kim + lim * kre;
            }
        }
        if (H > 1) {
            zre = __shfl(zre, hk * NP + i);
            zim = __shfl(zim, hk * NP + i);
        }
    }
}

// y = M v (or conj(M) v) with M in R-layout registers, v in LDS; the result is replicated in all
// lane groups. `part` is an LDS scratch of H*NP complex.
template <int NB, bool CONJ>
__device__ __forceinline__ void matvec(const double (&mre)[Geo<NB>::CPL],
                                       const double (&mim)[Geo<NB>::CPL], const double2* v,
                                       double2* part, int h, int i, double& yre, double& yim) {
    typedef Geo<NB> G;
    constexpr int NP = G::NP, CPL = G::CPL, H = G::H;
    double sre = 0, sim = 0;
#pragma unroll
    for (int cc = 0; cc < CPL; ++cc) {
        const double2 x = v[h * CPL + cc];
        const double mr = mre[cc], mi = CONJ ? -mim[cc] : mim[cc];
        sre += mr * x.x - mi * x.y;
        sim += mr * x.y + mi * x.x;
    }
    part[h * NP + i] = make_double2(sre, sim);
    wave_sync();
    yre = 0;
    yim = 0;
#pragma unroll
    for (int hh = 0; hh < H; ++hh) {
        const double2 p = part[hh * NP + i];
        yre += p.x;
        yim += p.y;
    }
    wave_sync();
}

// <t|psi> over the first NP lanes (lane group 0), result uniform.
__device__ __forceinline__ void inner(const double2 t, const double2 p, bool active, double& re,
                                      double& im) {
    // conj(t) * p
    double pr = active ? (t.x * p.x + t.y * p.y) : 0.0;
    double pi = active ? (t.x * p.y - t.y * p.x) : 0.0;
    re = wave_sum(pr);
    im = wave_sum(pi);
}

// Evaluate the selected costs on the S states held in `vecs` (LDS, [S][NP]).
// If lam != nullptr also adds dC/dRe + i dC/dIm into lam (LDS, [S][NP]).
template <int NB>
__device__ __forceinline__ double eval_costs(const SweepArgs& args, bool step_pass,
                                             bool final_pass, const double2* vecs, double2* lam,
                                             int h, int i) {
    typedef Geo<NB> G;
    constexpr int NP = G::NP;
    const int S = args.S;
    const bool act = (h == 0);
    double total = 0;
    for (int ci = 0; ci < args.cost_count; ++ci) {
        const DevCost c = args.costs[ci];
        const bool on = c.step_cost ? step_pass : final_pass;
        if (!on) continue;
        const double2* pool = args.cost_vectors + (size_t)c.vec_offset * NP;
        if (c.kind == QOCX_DEV_COST_COHERENT) {
            double tre = 0, tim = 0;
            for (int s = 0; s < S; ++s) {
                double r, m;
                inner(pool[s * NP + i], vecs[s * NP + i], act, r, m);
                tre += r;
                tim += m;
            }
            total += c.scale * (1.0 - (tre * tre + tim * tim) / ((double)S * S));
            if (lam != nullptr && act) {
                const double f = -2.0 * c.scale / ((double)S * S);
                for (int s = 0; s < S; ++s) {
                    const double2 t = pool[s * NP + i];
                    double2 l = lam[s * NP + i];
                    l.x += f * (tre * t.x - tim * t.y);
                    l.y += f * (tre * t.y + tim * t.x);
                    lam[s * NP + i] = l;
                }
            }
        } else if (c.kind == QOCX_DEV_COST_INCOHERENT) {
            double fid = 0;
            const double f = -2.0 * c.scale / (double)S;
            for (int s = 0; s < S; ++s) {
                double r, m;
                const double2 t = pool[s * NP + i];
                inner(t, vecs[s * NP + i], act, r, m);
                fid += r * r + m * m;
                if (lam != nullptr && act) {
                    double2 l = lam[s * NP + i];
                    l.x += f * (r * t.x - m * t.y);
                    l.y += f * (r * t.y + m * t.x);
                    lam[s * NP + i] = l;
                }
            }
            total += c.scale * (1.0 - fid / (double)S);
        } else {  // QOCX_DEV_COST_FORBID
            int base = 0;
            double acc = 0;
            for (int s = 0; s < S; ++s) {
                const int fs = args.cost_counts[c.cnt_offset + s];
                const double w = 1.0 / (double)fs;
                for (int f = 0; f < fs; ++f) {
                    double r, m;
                    const double2 t = pool[(size_t)(base + f) * NP + i];
                    inner(t, vecs[s * NP + i], act, r, m);
                    acc += w * (r * r + m * m);
                    if (lam != nullptr && act) {
                        const double g = 2.0 * c.scale * w;
                        double2 l = lam[s * NP + i];
                        l.x += g * (r * t.x - m * t.y);
                        l.y += g * (r * t.y + m * t.x);
                        lam[s * NP + i] = l;
                    }
                }
                base += fs;
            }
            total += c.scale * acc;
        }
    }
    if (lam != nullptr) wave_sync();
    return total;
}

template <int NB>
struct SweepLds {
    typedef Geo<NB> G;
    static constexpr int TR_OFF = 0;                                  // transpose buffer
    static constexpr int TR_BYTES = G::NP * G::TP * 16;
    static constexpr int PART_OFF = TR_OFF + TR_BYTES;                // H*NP complex
    static constexpr int TMP_OFF = PART_OFF + 64 * 16;                // NP complex
    static constexpr int VEC_OFF = TMP_OFF + G::NP * 16;              // [S][NP] states
    static int bytes(int S) { return VEC_OFF + 2 * S * G::NP * 16; }  // + [S][NP] lambda
};

// R-image rows (registers) -> transposed R-layout registers through LDS.
template <int NB>
__device__ __forceinline__ void transpose_r(double (&mre)[Geo<NB>::CPL], double (&mim)[Geo<NB>::CPL],
                                            double2* tr, int h, int i) {
    typedef Geo<NB> G;
    constexpr int CPL = G::CPL, TP = G::TP;
#pragma unroll
    for (int cc = 0; cc < CPL; ++cc) tr[i * TP + h * CPL + cc] = make_double2(mre[cc], mim[cc]);
    wave_sync();
#pragma unroll
    for (int cc = 0; cc < CPL; ++cc) {
        const double2 e = tr[(h * CPL + cc) * TP + i];
        mre[cc] = e.x;
        mim[cc] = e.y;
    }
    wave_sync();
}

template <int NB>
__global__ __launch_bounds__(64) void sweep_kernel(SweepArgs args) {
    typedef Geo<NB> G;
    typedef SweepLds<NB> L;
    constexpr int NP = G::NP, CPL = G::CPL;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double2* tr = reinterpret_cast<double2*>(smem + L::TR_OFF);
    double2* part = reinterpret_cast<double2*>(smem + L::PART_OFF);
    double2* tmp = reinterpret_cast<double2*>(smem + L::TMP_OFF);
    double2* vecs = reinterpret_cast<double2*>(smem + L::VEC_OFF);
    const int S = args.S;
    double2* lam = vecs + S * NP;
    const int b = blockIdx.x;
    const int lane = lane_id(), i = lane % NP, h = lane / NP;
    const int nsteps = args.nsteps;
    const size_t cap = args.slot_cap;
    double2* states_b = args.states + (size_t)b * cap * S * NP;
    double2* xs_b = args.xs + (size_t)b * cap * S * NP;
    int* offs_b = args.offs + (size_t)b * (nsteps + 1);
    const bool g0 = (h == 0);

    for (int s = 0; s < S; ++s)
        if (g0) {
            const double2 p = args.psi0[s * NP + i];
            vecs[s * NP + i] = p;
            states_b[(size_t)s * NP + i] = p;
        }
    wave_sync();

    double cost = 0;
    int slot = 0;
    bool overflow = false;
    for (int step = 0; step <= nsteps; ++step) {
        if (step != 0 && (step % args.cost_eval_step) == 0)
            cost += eval_costs<NB>(args, true, false, vecs, nullptr, h, i);
        if (g0 && args.step_states != nullptr)
            for (int s = 0; s < S; ++s)
                args.step_states[(((size_t)b * (nsteps + 1) + step) * S + s) * NP + i] =
                    vecs[s * NP + i];
        if (lane == 0) offs_b[step] = slot;
        if (step == nsteps) break;
        const size_t m = (size_t)b * nsteps + step;
        const int nsub = 1 << min(max(args.s_arr[m], 0), 30);
        const double2* qi = args.q_img + m * G::MAT;
        const double2* li = args.lu_img + m * G::MAT;
        double qre[CPL], qim[CPL], lre[CPL], lim[CPL];
#pragma unroll
        for (int cc = 0; cc < CPL; ++cc) {
            const double2 e = qi[cc * 64 + lane];
            qre[cc] = e.x;
            qim[cc] = e.y;
            const double2 f = li[cc * 64 + lane];
            lre[cc] = f.x;
            lim[cc] = f.y;
        }
        const double2 dv = args.dinv[m * NP + i];
        const int pm = min(max(args.perm[m * NP + i], 0), NP - 1);
        for (int sub = 0; sub < nsub; ++sub) {
            if ((size_t)slot + 1 >= cap) {
                overflow = true;
                break;
            }
            for (int s = 0; s < S; ++s) {
                double yre, yim;
                matvec<NB, false>(qre, qim, vecs + s * NP, part, h, i, yre, yim);
                if (g0) tmp[i] = make_double2(yre, yim);
                wave_sync();
                const double2 zp = tmp[pm];  // z = Pi y
                wave_sync();
                double zre = zp.x, zim = zp.y;
                tri_solve<NB, true, true, false>(lre, lim, zre, zim, 0, 0, h, i);
                tri_solve<NB, false, false, false>(lre, lim, zre, zim, dv.x, dv.y, h, i);
                if (g0) {
                    const double2 p = make_double2(zre, zim);
                    vecs[s * NP + i] = p;
                    states_b[((size_t)(slot + 1) * S + s) * NP + i] = p;
                }
            }
            wave_sync();
            ++slot;
        }
        if (overflow) break;
    }
    if (overflow) {
        if (lane == 0) atomicOr(args.status, 4);
        return;
    }
    cost += eval_costs<NB>(args, false, true, vecs, nullptr, h, i);
    if (lane == 0) args.cost_out[b] = cost;
    if (g0)
        for (int s = 0; s < S; ++s) args.final_out[((size_t)b * S + s) * NP + i] = vecs[s * NP + i];
    if (!args.want_grad) return;

    // ---- adjoint sweep ---------------------------------------------------------------------
    for (int s = 0; s < S; ++s)
        if (g0) lam[s * NP + i] = make_double2(0, 0);
    wave_sync();
    // cotangent seeds on the final states: non-step costs, and step costs if the final step is
    // a cost step (schroedingerdiscrete.py:412-416 evaluates them before the loop ends).
    {
        const bool final_is_step = (nsteps % args.cost_eval_step) == 0;
        (void)eval_costs<NB>(args, final_is_step, true, vecs, lam, h, i);
    }
    for (int step = nsteps - 1; step >= 0; --step) {
        const size_t m = (size_t)b * nsteps + step;
        const int nsub = 1 << min(max(args.s_arr[m], 0), 30);
        const double2* qi = args.q_img + m * G::MAT;
        const double2* li = args.lu_img + m * G::MAT;
        double qre[CPL], qim[CPL], lre[CPL], lim[CPL];
#pragma unroll
        for (int cc = 0; cc < CPL; ++cc) {
            const double2 e = qi[cc * 64 + lane];
            qre[cc] = e.x;
            qim[cc] = e.y;
            const double2 f = li[cc * 64 + lane];
            lre[cc] = f.x;
            lim[cc] = f.y;
        }
        transpose_r<NB>(qre, qim, tr, h, i);
        transpose_r<NB>(lre, lim, tr, h, i);
        const double2 dv = args.dinv[m * NP + i];
        const int pm = min(max(args.perm[m * NP + i], 0), NP - 1);
        for (int sub = nsub - 1; sub >= 0; --sub) {
            --slot;
            for (int s = 0; s < S; ++s) {
                const double2 l0 = lam[s * NP + i];
                double zre = l0.x, zim = l0.y;
                // U^H w = lambda (lower, conj, diagonal 1/conj(U_kk)); L^H v = w (upper, unit, conj)
                tri_solve<NB, true, false, true>(lre, lim, zre, zim, dv.x, dv.y, h, i);
                tri_solve<NB, false, true, true>(lre, lim, zre, zim, 0, 0, h, i);
                if (g0) tmp[pm] = make_double2(zre, zim);  // x = Pi^T v
                wave_sync();
                if (g0) xs_b[((size_t)slot * S + s) * NP + i] = tmp[i];
                double yre, yim;
                matvec<NB, true>(qre, qim, tmp, part, h, i, yre, yim);  // lambda = Q^H x
                if (g0) lam[s * NP + i] = make_double2(yre, yim);
                wave_sync();
            }
        }
        if (step != 0 && (step % args.cost_eval_step) == 0 && args.has_step_costs) {
            // step costs were evaluated on the states *before* evolving from `step`
            if (g0)
                for (int s = 0; s < S; ++s)
                    vecs[s * NP + i] = states_b[((size_t)slot * S + s) * NP + i];
            wave_sync();
            (void)eval_costs<NB>(args, true, false, vecs, lam, h, i);
        }
    }
}

// ------------------------------------------------------------------------------------------
// K3: Krylov-chain adjoint of the Pade step
// ------------------------------------------------------------------------------------------
// For one sub-step with x = P^-H lambda', sigma = psi + psi', delta = psi - psi':
//   abar = sum_{i=0}^{12} tau_i rho_i^H,  tau_i = (a^H)^i x,
//   rho_i = sum_j cu_{i+j+1} a^j sigma + cv_{i+j+1} a^j delta,
// with cu/cv the odd/even Pade coefficients (u(a) = sum cu_m a^m, v(a) = b0 I + sum cv_m a^m).
// It equals the reverse-mode cotangent of expm_pade's input (tests/test_device_model.py).

template <int NB>
struct KrylovLds {
    typedef Geo<NB> G;
    static constexpr int V_OFF = 0;                        // 3 vectors (sigma, delta, tau)
    static constexpr int PART_OFF = V_OFF + 3 * G::NP * 16;  // 3 * 64 partial sums
    static constexpr int RHO_OFF = PART_OFF + 3 * 64 * 16;   // 13 rho vectors
    static constexpr int BYTES = RHO_OFF + 13 * G::NP * 16;
};

template <int NB>
__global__ __launch_bounds__(64) void krylov_grad_kernel(KrylovArgs args) {
    typedef Geo<NB> G;
    typedef KrylovLds<NB> L;
    constexpr int NP = G::NP, CPL = G::CPL, H = G::H;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double2* vv = reinterpret_cast<double2*>(smem + L::V_OFF);
    double2* part = reinterpret_cast<double2*>(smem + L::PART_OFF);
    double2* rho_l = reinterpret_cast<double2*>(smem + L::RHO_OFF);
    const int step = blockIdx.x, b = blockIdx.y;
    const int lane = lane_id(), i = lane % NP, h = lane / NP;
    const int nsteps = args.nsteps, S = args.S, K = args.K;
    const size_t m = (size_t)b * nsteps + step;
    const int sq = min(max(args.s_arr[m], 0), 30);
    const double dts = args.dt * ldexp(1.0, -sq);
    const StepInterp si = args.interp[step];
    const double* ctl_b = args.controls + (size_t)b * args.nc * K;
    const size_t tsel = (args.nt == 1) ? 0 : (size_t)step;
    const double2* h0r = args.h0_rimg + tsel * G::MAT;
    const double2* h0t = args.h0_timg + tsel * G::MAT;
    const double2* gr = args.g_rimg + tsel * K * G::MAT;
    const double2* gt = args.g_timg + tsel * K * G::MAT;

    // a (rows) and a^H (rows) of the scaled generator
    double are[CPL], aim[CPL], hre[CPL], him[CPL];
    {
        double xr[CPL], xi[CPL], tr_[CPL], ti_[CPL];
#pragma unroll
        for (int cc = 0; cc < CPL; ++cc) {
            const double2 e = h0r[cc * 64 + lane];
            xr[cc] = e.x;
            xi[cc] = e.y;
            const double2 f = h0t[cc * 64 + lane];
            tr_[cc] = f.x;
            ti_[cc] = f.y;
        }
        for (int k = 0; k < K; ++k) {
            const double uk = control_at(ctl_b, si, K, k);
#pragma unroll
            for (int cc = 0; cc < CPL; ++cc) {
                const double2 e = gr[(size_t)k * G::MAT + cc * 64 + lane];
                xr[cc] += uk * e.x;
                xi[cc] += uk * e.y;
                const double2 f = gt[(size_t)k * G::MAT + cc * 64 + lane];
                tr_[cc] += uk * f.x;
                ti_[cc] += uk * f.y;
            }
        }
#pragma unroll
        for (int cc = 0; cc < CPL; ++cc) {
            are[cc] = dts * xi[cc];   // a = -i dts H
            aim[cc] = -dts * xr[cc];
            hre[cc] = dts * ti_[cc];  // a^H[i][c] = conj(a[c][i]) = conj(-i dts H[c][i])
            him[cc] = dts * tr_[cc];
        }
    }

    double abr[CPL], abi[CPL];
#pragma unroll
    for (int cc = 0; cc < CPL; ++cc) {
        abr[cc] = 0;
        abi[cc] = 0;
    }

    const size_t cap = args.slot_cap;
    const double2* states_b = args.states + (size_t)b * cap * S * NP;
    const double2* xs_b = args.xs + (size_t)b * cap * S * NP;
    const int t0 = args.offs[(size_t)b * (nsteps + 1) + step];
    const int nsub = 1 << sq;
    if (t0 < 0 || (size_t)t0 + (size_t)nsub >= cap) return;  // sweep overflowed (status bit 2)
    for (int sub = 0; sub < nsub; ++sub) {
        for (int s = 0; s < S; ++s) {
            const size_t t = (size_t)t0 + sub;
            const double2 x = xs_b[(t * S + s) * NP + i];
            const double2 p0 = states_b[(t * S + s) * NP + i];
            const double2 p1 = states_b[((t + 1) * S + s) * NP + i];
            double sgr = p0.x + p1.x, sgi = p0.y + p1.y;
            double dlr = p0.x - p1.x, dli = p0.y - p1.y;
            double tar = x.x, tai = x.y;
            double rhr[13], rhi[13], tvr[13], tvi[13];
#pragma unroll
            for (int ii = 0; ii < 13; ++ii) {
                rhr[ii] = 0;
                rhi[ii] = 0;
            }
#pragma unroll
            for (int jj = 0; jj < 13; ++jj) {
#pragma unroll
                for (int ii = 0; ii + jj < 13; ++ii) {
                    const int mm = ii + jj + 1;
                    const double coef = PADE_B[mm];
                    if (mm & 1) {
                        rhr[ii] += coef * sgr;
                        rhi[ii] += coef * sgi;
                    } else {
                        rhr[ii] += coef * dlr;
                        rhi[ii] += coef * dli;
                    }
                }
                tvr[jj] = tar;
                tvi[jj] = tai;
                if (jj < 12) {
                    if (h == 0) {
                        vv[i] = make_double2(sgr, sgi);
                        vv[NP + i] = make_double2(dlr, dli);
                        vv[2 * NP + i] = make_double2(tar, tai);
                    }
                    wave_sync();
                    double s0r = 0, s0i = 0, s1r = 0, s1i = 0, s2r = 0, s2i = 0;
#pragma unroll
                    for (int cc = 0; cc < CPL; ++cc) {
                        const double2 v0 = vv[h * CPL + cc];
                        const double2 v1 = vv[NP + h * CPL + cc];
                        const double2 v2 = vv[2 * NP + h * CPL + cc];
                        s0r += are[cc] * v0.x - aim[cc] * v0.y;
                        s0i += are[cc] * v0.y + aim[cc] * v0.x;
                        s1r += are[cc] * v1.x - aim[cc] * v1.y;
                        s1i += are[cc] * v1.y + aim[cc] * v1.x;
                        s2r += hre[cc] * v2.x - him[cc] * v2.y;
                        s2i += hre[cc] * v2.y + him[cc] * v2.x;
                    }
                    part[lane] = make_double2(s0r, s0i);
                    part[64 + lane] = make_double2(s1r, s1i);
                    part[128 + lane] = make_double2(s2r, s2i);
                    wave_sync();
                    sgr = sgi = dlr = dli = tar = tai = 0;
#pragma unroll
                    for (int hh = 0; hh < H; ++hh) {
                        const double2 q0 = part[hh * NP + i];
                        const double2 q1 = part[64 + hh * NP + i];
                        const double2 q2 = part[128 + hh * NP + i];
                        sgr += q0.x;
                        sgi += q0.y;
                        dlr += q1.x;
                        dli += q1.y;
                        tar += q2.x;
                        tai += q2.y;
                    }
                }
            }
            // abar += sum_t tau_t rho_t^H
            if (h == 0) {
#pragma unroll
                for (int tt = 0; tt < 13; ++tt) rho_l[tt * NP + i] = make_double2(rhr[tt], rhi[tt]);
            }
            wave_sync();
#pragma unroll
            for (int tt = 0; tt < 13; ++tt) {
#pragma unroll
                for (int cc = 0; cc < CPL; ++cc) {
                    const double2 r = rho_l[tt * NP + h * CPL + cc];
                    // tau * conj(rho)
                    abr[cc] += tvr[tt] * r.x + tvi[tt] * r.y;
                    abi[cc] += tvi[tt] * r.x - tvr[tt] * r.y;
                }
            }
            wave_sync();
        }
    }

    // g_k = Re <abar, E_k>, E_k = d a / d u_k = -i dts G_k  (H-bar = i dt M-bar, Appendix A)
    for (int k = 0; k < K; ++k) {
        double acc = 0;
#pragma unroll
        for (int cc = 0; cc < CPL; ++cc) {
            const double2 e = gr[(size_t)k * G::MAT + cc * 64 + lane];
            acc += abr[cc] * (dts * e.y) + abi[cc] * (-dts * e.x);
        }
        acc = wave_sum(acc);
        if (lane == 0) args.gstep[m * K + k] = acc;
    }
}

// ------------------------------------------------------------------------------------------
// K4: transpose of the linear interpolation: grads[b][ic][k] = sum_j W[j][ic] gstep[b][j][k]
// ------------------------------------------------------------------------------------------
__global__ void scatter_kernel(ScatterArgs args) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)args.B * args.nc * args.K;
    if (idx >= total) return;
    const int k = (int)(idx % args.K);
    const int ic = (int)((idx / args.K) % args.nc);
    const size_t b = idx / ((size_t)args.K * args.nc);
    double acc = 0;
    for (int e = args.row_ptr[ic]; e < args.row_ptr[ic + 1]; ++e)
        acc += args.weight[e] * args.gstep[(b * args.nsteps + args.col_step[e]) * args.K + k];
    args.grads[idx] = acc;
}

// ------------------------------------------------------------------------------------------
// self test of the wave primitives
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void selftest_kernel(double* out) {
    const int lane = lane_id();
    const double v = (double)((lane * 37) % 64) + 0.25;
    out[lane] = wave_max(v);
    out[64 + lane] = wave_sum(v);
    // MFMA layout probe: C = A * B with A[i][k] = i + 16 k (16x4), B[k][j] = 100 k + j (4x16)
    const double a = (double)((lane & 15) + 16 * (lane >> 4));
    const double bb = (double)(100 * (lane >> 4) + (lane & 15));
    d4 c = {0, 0, 0, 0};
    c = mfma_f64(a, bb, c);
    for (int r = 0; r < 4; ++r) out[128 + lane * 4 + r] = c[r];
    out[384 + lane] = readlane_f64(v, 5);
    out[448 + lane] = dpp_f64<0x140>(v);
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
template <int NB>
static void launch_factor_t(const FactorArgs& a, int nsteps, int batch, hipStream_t st) {
    hipLaunchKernelGGL(pade_factor_kernel<NB>, dim3(nsteps, batch), dim3(64),
                       FactorLds<NB>::BYTES, st, a);
}
template <int NB>
static void launch_factor_explicit_t(const double2* a_in, int n, const FactorArgs& a, int count,
                                     hipStream_t st) {
    hipLaunchKernelGGL(pade_factor_explicit_kernel<NB>, dim3(count), dim3(64),
                       FactorLds<NB>::BYTES, st, a_in, n, a);
}
template <int NB>
static void launch_sweep_t(const SweepArgs& a, int batch, hipStream_t st) {
    const int bytes = SweepLds<NB>::bytes(a.S);
    if (bytes > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sweep_kernel<NB>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    hipLaunchKernelGGL(sweep_kernel<NB>, dim3(batch), dim3(64), bytes, st, a);
}
template <int NB>
static void launch_krylov_t(const KrylovArgs& a, int nsteps, int batch, hipStream_t st) {
    hipLaunchKernelGGL(krylov_grad_kernel<NB>, dim3(nsteps, batch), dim3(64),
                       KrylovLds<NB>::BYTES, st, a);
}

void launch_factor(int nb, const FactorArgs& a, int nsteps, int batch, hipStream_t st) {
    if (nb == 1) launch_factor_t<1>(a, nsteps, batch, st);
    else launch_factor_t<2>(a, nsteps, batch, st);
}
void launch_factor_explicit(int nb, const double2* a_in, int n, const FactorArgs& a, int count,
                            hipStream_t st) {
    if (nb == 1) launch_factor_explicit_t<1>(a_in, n, a, count, st);
    else launch_factor_explicit_t<2>(a_in, n, a, count, st);
}
void launch_sweep(int nb, const SweepArgs& a, int batch, hipStream_t st) {
    if (nb == 1) launch_sweep_t<1>(a, batch, st);
    else launch_sweep_t<2>(a, batch, st);
}
int sweep_lds_bytes(int nb, int S) { return nb == 1 ? SweepLds<1>::bytes(S) : SweepLds<2>::bytes(S); }
void launch_krylov(int nb, const KrylovArgs& a, int nsteps, int batch, hipStream_t st) {
    if (nb == 1) launch_krylov_t<1>(a, nsteps, batch, st);
    else launch_krylov_t<2>(a, nsteps, batch, st);
}
void launch_scatter(const ScatterArgs& a, hipStream_t st) {
    const size_t total = (size_t)a.B * a.nc * a.K;
    if (total == 0) return;
    hipLaunchKernelGGL(scatter_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a);
}
void launch_selftest(double* out, hipStream_t st) {
    hipLaunchKernelGGL(selftest_kernel, dim3(1), dim3(64), 0, st, out);
}

}  // namespace qocx
