// qocx_kernels.hip - hand-written CDNA4 (gfx950) kernels of the GRAPE propagation engine.
//
// One 64-lane wavefront owns one Hilbert-space tile (n <= 32, padded to NP = 16 or 32):
//
//   K1a pade_pq    : generator a = -i dt H(u_mid) 2^-s  ->  Pade-13 numerator/denominator
//                    Q = v+u, P = v-u: 6 complex GEMMs on v_mfma_f64_16x16x4_f64 (A operand
//                    staged in LDS, B operand / accumulators in registers).
//                    reference: qoc/standard/functions/expm.py:153-159, :210-246,
//                    qoc/core/schroedingerdiscrete.py:483-489, mathmethods.py:36-67, :90-93
//   K1b lu         : LU(P) with partial pivoting (numpy.linalg.solve = LAPACK zgesv,
//                    expm.py:246), in place, at several waves per SIMD to hide its latencies.
//   K2 sweep       : psi_{j+1} = (P^-1 Q)^(2^s) psi_j (serial in j, one wave per seed), state
//                    costs, then lambda_j = Q^H P^-H lambda_{j+1} backwards.
//                    reference: schroedingerdiscrete.py:393-436, expm.py:246-250, costs/*.py
//   K3 krylov_grad : d cost / d u_mid from Krylov chains of a, a^H (the hand-derived adjoint that
//                    replaces autograd's tape: autogradutil.py:26-30); SURVEY.md Appendix A.
//   K4 scatter     : transpose of the linear interpolation (mathmethods.py:33, :54-65).
//
// Register / memory layouts (H = 64/NP lane groups, CPL = NP/H columns per lane)
//   C-layout : MFMA accumulator layout. lane = 16*q + c; tile (ti,tj), reg r holds element
//              (row 16 ti + 4 r + q, col 16 tj + c).  A C-layout tile IS the B operand of the
//              next MFMA (k-step kk <-> row block kk>>2, reg kk&3), so GEMM chains need no
//              shuffles; only the A operand goes through LDS.
//   R-layout : lane = h*NP + i holds row i and the interleaved columns cc*H + h, cc < CPL
//              (balanced work for LU / triangular updates). Loading complex index cc*64 + lane
//              of a COLUMN-MAJOR NP x NP matrix lands directly in R-layout, one contiguous KiB
//              per instruction. Every matrix in HBM (Q, LU, H0, G_k) is such a column-major
//              image.
//   F-layout : lane l (any group) holds the full row l % NP (index c*NP + i of the image).
#include <algorithm>
#include <type_traits>

#include "qocx_wave.h"
#include "qocx_sweep_common.h"
#include "qocx_lu.h"
#include "qocx_lu5.h"

namespace qocx {

// ------------------------------------------------------------------------------------------
// K1a: Pade-13 numerator / denominator
// ------------------------------------------------------------------------------------------

struct PqOut {
    double2* q_img;   // column-major Q
    double2* p_img;   // column-major P (LU'd in place by K1b)
    int* s_out;       // squarings (and the Pade order: step_entry, qocx_wave.h)
    int* status;      // bit 1: non-finite norm
    int pade_policy;  // FactorArgs::pade_policy
    // pack8 (n <= 8): the tile holds TWO steps as its diagonal 8 x 8 blocks (pade_pq8_kernel). P goes out
    // as the packed tile (K1b unpacks the inverse), Q as the two padded images the sweeps read.
    int pack8 = 0;
    double2* q_img2 = nullptr;  // Q image of the second step, nullptr: there is none
    int* s_out2 = nullptr;
};

// pack8: block (0,0) of the tile in the LDS slot -> img0, block (1,1) -> img1 (column-major 16 x 16 images
// of 8 x 8 problems, the pad block the identity). Lane l: column l % 16, rows 4 (l / 16) .. + 3 of the image.
__device__ __forceinline__ void lds_to_images8(const double* lre, const double* lim, double2* img0, double2* img1) {
    typedef Geo<1> G;
    const int lane = lane_id(), col = lane & 15, r0 = 4 * (lane >> 4);
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
        double2* img = blk == 0 ? img0 : img1;
        if (img == nullptr) continue;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int row = r0 + rr;
            double2 v = make_double2(row == col ? 1.0 : 0.0, 0.0);
            if (row < 8 && col < 8) {
                const int off = (8 * blk + row) * G::PITCH + 8 * blk + col;
                v = make_double2(lre[off], lim[off]);
            }
            img[col * 16 + row] = v;
        }
    }
}

template <int NB>
struct PqLds {
    static constexpr int SLOT = 3 * Geo<NB>::PLANE * 8;   // planar A-operand slot: re | im | re+im
    static constexpr int BYTES = SLOT + 2 * 16 * 18 * 8;  // + one 16 x 16 complex tile (mirror)
};

// Complex products by the 3M scheme: with T1 = Ar Br, T2 = Ai Bi, T3 = (Ar + Ai)(Br + Bi),
//   Re(AB) = T1 - T2,  Im(AB) = T3 - T1 - T2,
// i.e. three real MFMA chains per complex product instead of four. The rounding error stays
// normwise O(eps ||A|| ||B||) (Higham, Accuracy and Stability, 23.2.4) - the parity tests hold
// the result to the same 1e-10 / 1e-8 gates as before.
template <int NB>
struct CAcc3 {
    d4 t1[NB][NB], t2[NB][NB], t3[NB][NB];
};

template <int NB>
__device__ __forceinline__ void acc3_init(CAcc3<NB>& a, const CMat<NB>& c) {
#pragma unroll
    for (int ti = 0; ti < NB; ++ti)
#pragma unroll
        for (int tj = 0; tj < NB; ++tj) {
            a.t1[ti][tj] = c.re[ti][tj];
            a.t2[ti][tj] = d4{0, 0, 0, 0};
            a.t3[ti][tj] = c.re[ti][tj] + c.im[ti][tj];
        }
}
template <int NB>
__device__ __forceinline__ void acc3_zero(CAcc3<NB>& a) {
#pragma unroll
    for (int ti = 0; ti < NB; ++ti)
#pragma unroll
        for (int tj = 0; tj < NB; ++tj) {
            a.t1[ti][tj] = d4{0, 0, 0, 0};
            a.t2[ti][tj] = d4{0, 0, 0, 0};
            a.t3[ti][tj] = d4{0, 0, 0, 0};
        }
}
template <int NB>
__device__ __forceinline__ void acc3_finish(CMat<NB>& c, const CAcc3<NB>& a) {
#pragma unroll
    for (int ti = 0; ti < NB; ++ti)
#pragma unroll
        for (int tj = 0; tj < NB; ++tj) {
            c.re[ti][tj] = a.t1[ti][tj] - a.t2[ti][tj];
            c.im[ti][tj] = a.t3[ti][tj] - a.t1[ti][tj] - a.t2[ti][tj];
        }
}

// C-layout registers -> three planes (re, im, re + im) of the planar A-operand slot
template <int NB>
__device__ __forceinline__ void cmat_to_lds3(const CMat<NB>& m, double* lre, double* lim,
                                             double* lsum) {
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
                lsum[off] = m.re[ti][tj][r] + m.im[ti][tj][r];
            }
}

// acc += A * B (3M). A from the three planes, B fragment (re, im) from `bf`. UPPER: only the
// tiles on and above the diagonal (the product is known to be Hermitian or skew-Hermitian).
template <int NB, bool UPPER, class BFrag>
__device__ __forceinline__ void zgemm3_acc(CAcc3<NB>& acc, const double* lre, const double* lim,
                                           const double* lsum, BFrag bf) {
    typedef Geo<NB> G;
    const int q = lane_id() >> 4, c = lane_id() & 15;
#pragma unroll
    for (int kk = 0; kk < 4 * NB; ++kk) {
        double are[NB], aim[NB], asum[NB];
#pragma unroll
        for (int ti = 0; ti < NB; ++ti) {
            const int off = (16 * ti + c) * G::PITCH + 4 * kk + q;
            are[ti] = lre[off];
            aim[ti] = lim[off];
            asum[ti] = lsum[off];
        }
#pragma unroll
        for (int tj = 0; tj < NB; ++tj) {
            double bre, bim;
            bf(kk, tj, bre, bim);
            const double bsum = bre + bim;
#pragma unroll
            for (int ti = 0; ti < NB; ++ti) {
                if (UPPER && ti > tj) continue;
                acc.t1[ti][tj] = mfma_f64(are[ti], bre, acc.t1[ti][tj]);
                acc.t2[ti][tj] = mfma_f64(aim[ti], bim, acc.t2[ti][tj]);
                acc.t3[ti][tj] = mfma_f64(asum[ti], bsum, acc.t3[ti][tj]);
            }
        }
    }
}

// Lower tiles of a (skew-)Hermitian product from the upper ones: tile (ti, tj), ti > tj, is
// sign * conj(tile (tj, ti))^T; the 16 x 16 transposition goes through an LDS scratch tile.
template <int NB>
__device__ __forceinline__ void mirror_lower(CMat<NB>& m, double sign, double* scratch) {
    const int q = lane_id() >> 4, c = lane_id() & 15;
    double* sre = scratch;
    double* sim = scratch + 16 * 18;
#pragma unroll
    for (int ti = 1; ti < NB; ++ti)
#pragma unroll
        for (int tj = 0; tj < ti; ++tj) {
            wave_sync();
#pragma unroll
            for (int r = 0; r < 4; ++r) {  // element (4r+q, c) of the upper tile -> slot [c][4r+q]
                sre[c * 18 + 4 * r + q] = m.re[tj][ti][r];
                sim[c * 18 + 4 * r + q] = m.im[tj][ti][r];
            }
            wave_sync();
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                m.re[ti][tj][r] = sign * sre[(4 * r + q) * 18 + c];
                m.im[ti][tj][r] = -sign * sim[(4 * r + q) * 18 + c];
            }
        }
}

// The body of K1a; `gen(a)` builds the unscaled generator in C-layout. HERM: the generator is
// exactly skew-Hermitian (Hermitian H), so a^2, a^4, a^6, w2 and v are Hermitian and u = a w2 is
// skew-Hermitian: only the upper tiles of the six products are computed.
template <int NB, bool HERM, class Gen>
__device__ __forceinline__ void pade_pq_body(Gen gen, const PqOut& out, char* smem) {
    typedef Geo<NB> G;
    constexpr bool UP = HERM && NB > 1;
    constexpr bool REGEN_A = true;
    double* lre = reinterpret_cast<double*>(smem);
    double* lim = lre + G::PLANE;
    double* lsum = lim + G::PLANE;
    double* mscr = lsum + G::PLANE;
    const int lane = lane_id();

    // ---- generator, 1-norm, scaling (expm.py:116, :238-241) -----------------------------
    CMat<NB> a;
    gen(a);
    double norm1 = 0;
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
        norm1 = (tj == 0) ? s : fmax(norm1, s);
    }
    norm1 = wave_max(norm1);
    int sq = 0;
    int order = pade_order_for(norm1, out.pade_policy);
    {
        double th = QOCX_THETA13;
        while (norm1 > th && sq < 30) {
            th *= 2.0;
            ++sq;
        }
        if (!(norm1 <= th)) {  // inf / nan / absurd
            if (lane == 0) atomicOr(out.status, 2);
            sq = 0;
            order = 13;
        }
    }
    const double scale = ldexp(1.0, -sq);
    if (sq > 0) cmat_scale<NB>(a, scale);
    if (lane == 0) {
        *out.s_out = step_entry(sq, order);
        if (out.s_out2 != nullptr) *out.s_out2 = step_entry(sq, order);
    }

    const int q = lane >> 4, c = lane & 15;
    CMat<NB> u, v;
    cmat_to_lds3<NB>(a, lre, lim, lsum);
    wave_sync();
    if (order != 13) {
        // ---- orders 3, 5, 7, 9 (qocx_wave.h; the shape of the reference's pade3..pade9,
        // expm.py:119-150): x2 = a a, x_{2j} = x2 x_{2j-2}; w = sum b_{2j+1} x_{2j},
        // v = sum b_{2j} x_{2j} + b0 I, u = a w + b1 a. No squarings at these norms.
        const double* bt = pade_table(order);
        CMat<NB> w, x;
        {
            CAcc3<NB> acc;
            acc3_zero<NB>(acc);
            zgemm3_acc<NB, UP>(acc, lre, lim, lsum, [&](int kk, int tj, double& bre, double& bim) {
                bre = a.re[kk >> 2][tj][kk & 3];
                bim = a.im[kk >> 2][tj][kk & 3];
            });
            acc3_finish<NB>(x, acc);
            if (UP) mirror_lower<NB>(x, 1.0, mscr);
        }
#pragma unroll
        for (int ti = 0; ti < NB; ++ti)
#pragma unroll
            for (int tj = 0; tj < NB; ++tj) {
                w.re[ti][tj] = bt[3] * x.re[ti][tj];
                w.im[ti][tj] = bt[3] * x.im[ti][tj];
                v.re[ti][tj] = bt[2] * x.re[ti][tj];
                v.im[ti][tj] = bt[2] * x.im[ti][tj];
                if (ti == tj) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (4 * r + q == c) v.re[ti][tj][r] += bt[0];
                }
            }
        wave_sync();
        if (order >= 5) {
            cmat_to_lds3<NB>(x, lre, lim, lsum);  // x2 is the A operand of every further product
            wave_sync();
            for (int j = 2; 2 * j < order; ++j) {
                CAcc3<NB> acc;
                acc3_zero<NB>(acc);
                zgemm3_acc<NB, UP>(acc, lre, lim, lsum, [&](int kk, int tj, double& bre, double& bim) {
                    bre = x.re[kk >> 2][tj][kk & 3];
                    bim = x.im[kk >> 2][tj][kk & 3];
                });
                acc3_finish<NB>(x, acc);  // (the product is complete: its B operand may go)
                if (UP) mirror_lower<NB>(x, 1.0, mscr);
                const double bw = bt[2 * j + 1], bv = bt[2 * j];
#pragma unroll
                for (int ti = 0; ti < NB; ++ti)
#pragma unroll
                    for (int tj = 0; tj < NB; ++tj) {
                        w.re[ti][tj] += bw * x.re[ti][tj];
                        w.im[ti][tj] += bw * x.im[ti][tj];
                        v.re[ti][tj] += bv * x.re[ti][tj];
                        v.im[ti][tj] += bv * x.im[ti][tj];
                    }
            }
            wave_sync();
            if constexpr (NB > 1) gen(a);  // (one tile: the generator stays in its 8 registers)
            cmat_to_lds3<NB>(a, lre, lim, lsum);
            wave_sync();
        }
#pragma unroll
        for (int ti = 0; ti < NB; ++ti)
#pragma unroll
            for (int tj = 0; tj < NB; ++tj) {
                u.re[ti][tj] = bt[1] * a.re[ti][tj];
                u.im[ti][tj] = bt[1] * a.im[ti][tj];
            }
        {
            CAcc3<NB> acc;
            acc3_init<NB>(acc, u);
            zgemm3_acc<NB, UP>(acc, lre, lim, lsum, [&](int kk, int tj, double& bre, double& bim) {
                bre = w.re[kk >> 2][tj][kk & 3];
                bim = w.im[kk >> 2][tj][kk & 3];
            });
            acc3_finish<NB>(u, acc);
            if (UP) mirror_lower<NB>(u, -1.0, mscr);
        }
        wave_sync();
    } else {
    // ---- a2 = a a ; a4 = a2 a2 ; a6 = a2 a4 (expm.py:154-156) ----------------------------
    CMat<NB> x2, x4, x6;
    {
        CAcc3<NB> acc;
        acc3_zero<NB>(acc);
        zgemm3_acc<NB, UP>(acc, lre, lim, lsum, [&](int kk, int tj, double& bre, double& bim) {
            bre = a.re[kk >> 2][tj][kk & 3];
            bim = a.im[kk >> 2][tj][kk & 3];
        });
        acc3_finish<NB>(x2, acc);
        if (UP) mirror_lower<NB>(x2, 1.0, mscr);
    }
    wave_sync();
    cmat_to_lds3<NB>(x2, lre, lim, lsum);
    wave_sync();
    {
        CAcc3<NB> acc;
        acc3_zero<NB>(acc);
        zgemm3_acc<NB, UP>(acc, lre, lim, lsum, [&](int kk, int tj, double& bre, double& bim) {
            bre = x2.re[kk >> 2][tj][kk & 3];
            bim = x2.im[kk >> 2][tj][kk & 3];
        });
        acc3_finish<NB>(x4, acc);
        if (UP) mirror_lower<NB>(x4, 1.0, mscr);
    }
    {
        CAcc3<NB> acc;
        acc3_zero<NB>(acc);
        zgemm3_acc<NB, UP>(acc, lre, lim, lsum, [&](int kk, int tj, double& bre, double& bim) {
            bre = x4.re[kk >> 2][tj][kk & 3];
            bim = x4.im[kk >> 2][tj][kk & 3];
        });
        acc3_finish<NB>(x6, acc);
        if (UP) mirror_lower<NB>(x6, 1.0, mscr);
    }
    wave_sync();

    // ---- w2 = a6 (b13 a6 + b11 a4 + b9 a2) + b7 a6 + b5 a4 + b3 a2 (expm.py:157) ---------
    // ---- v  = a6 (b12 a6 + b10 a4 + b8 a2) + b6 a6 + b4 a4 + b2 a2 + b0 I (expm.py:158) --
    cmat_to_lds3<NB>(x6, lre, lim, lsum);
    wave_sync();
    const double b0 = PADE_B[0], b1 = PADE_B[1], b2 = PADE_B[2], b3 = PADE_B[3], b4 = PADE_B[4],
                 b5 = PADE_B[5], b6 = PADE_B[6], b7 = PADE_B[7], b8 = PADE_B[8], b9 = PADE_B[9],
                 b10 = PADE_B[10], b11 = PADE_B[11], b12 = PADE_B[12], b13 = PADE_B[13];
    CMat<NB> w2;
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
    {
        CAcc3<NB> acc;
        acc3_init<NB>(acc, w2);
        zgemm3_acc<NB, UP>(acc, lre, lim, lsum, [&](int kk, int tj, double& bre, double& bim) {
            const int tb = kk >> 2, r = kk & 3;
            bre = b13 * x6.re[tb][tj][r] + b11 * x4.re[tb][tj][r] + b9 * x2.re[tb][tj][r];
            bim = b13 * x6.im[tb][tj][r] + b11 * x4.im[tb][tj][r] + b9 * x2.im[tb][tj][r];
        });
        acc3_finish<NB>(w2, acc);
        if (UP) mirror_lower<NB>(w2, 1.0, mscr);
    }
    {
        CAcc3<NB> acc;
        acc3_init<NB>(acc, v);
        zgemm3_acc<NB, UP>(acc, lre, lim, lsum, [&](int kk, int tj, double& bre, double& bim) {
            const int tb = kk >> 2, r = kk & 3;
            bre = b12 * x6.re[tb][tj][r] + b10 * x4.re[tb][tj][r] + b8 * x2.re[tb][tj][r];
            bim = b12 * x6.im[tb][tj][r] + b10 * x4.im[tb][tj][r] + b8 * x2.im[tb][tj][r];
        });
        acc3_finish<NB>(v, acc);
        if (UP) mirror_lower<NB>(v, 1.0, mscr);
    }
    wave_sync();

    // ---- u = a w2 + b1 a (expm.py:157) ; P = v - u ; Q = v + u (expm.py:246) ---------------
    if (REGEN_A) {  // rebuild the generator: keeping 64 more registers alive spills (measured +5 %)
        gen(a);
        if (sq > 0) cmat_scale<NB>(a, scale);
    }
    cmat_to_lds3<NB>(a, lre, lim, lsum);
    wave_sync();
#pragma unroll
    for (int ti = 0; ti < NB; ++ti)
#pragma unroll
        for (int tj = 0; tj < NB; ++tj) {
            u.re[ti][tj] = b1 * a.re[ti][tj];
            u.im[ti][tj] = b1 * a.im[ti][tj];
        }
    {
        CAcc3<NB> acc;
        acc3_init<NB>(acc, u);
        zgemm3_acc<NB, UP>(acc, lre, lim, lsum, [&](int kk, int tj, double& bre, double& bim) {
            bre = w2.re[kk >> 2][tj][kk & 3];
            bim = w2.im[kk >> 2][tj][kk & 3];
        });
        acc3_finish<NB>(u, acc);
        if (UP) mirror_lower<NB>(u, -1.0, mscr);
    }
    wave_sync();
    }  // order 13

    // C-layout -> LDS -> R-layout -> column-major images (one contiguous KiB per store)
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
    if constexpr (NB == 1) {
        if (out.pack8) lds_to_images8(lre, lim, out.q_img, out.q_img2);
        else lds_to_image<NB>(lre, lim, out.q_img);
    } else {
        lds_to_image<NB>(lre, lim, out.q_img);
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
    lds_to_image<NB>(lre, lim, out.p_img);
}

template <int NB, bool HERM>
__global__ __launch_bounds__(64) void pade_pq_kernel(FactorArgs args) {
    typedef Geo<NB> G;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int step = args.step0 + blockIdx.x, b = blockIdx.y;
    const int lane = lane_id();
    const size_t m = (size_t)b * args.nsteps + step;
    PqOut out;
    out.q_img = args.q_img + m * G::MAT;
    out.p_img = args.lu_img + m * G::MAT;
    out.s_out = args.s_arr + m;
    out.status = args.status;
    out.pade_policy = args.pade_policy;
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
    pade_pq_body<NB, HERM>(gen, out, smem);
}

// n <= 8: TWO consecutive steps of a seed per wave, as the diagonal 8 x 8 blocks of one 16 x 16 tile
// ("n = 8 packs several matrices per wavefront"). A block-diagonal generator stays block
// diagonal through every product of the Pade evaluation, its 1-norm is the larger of the two, so both
// steps take the order and the squaring count of the larger one (the smaller one is evaluated by a
// higher-order approximant than it needs: the same matrix to rounding) and the body runs unchanged - half
// the workgroups for the same steps. grid (ceil(seg_len / 2), seeds).
template <bool HERM>
__global__ __launch_bounds__(64) void pade_pq8_kernel(FactorArgs args) {
    typedef Geo<1> G;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int local = 2 * (int)blockIdx.x, b = blockIdx.y;
    const int step = args.step0 + local;
    const bool second = local + 1 < args.seg_len;
    const int lane = lane_id(), q = lane >> 4, c = lane & 15;
    const size_t m = (size_t)b * args.nsteps + step;
    PqOut out;
    out.q_img = args.q_img + m * G::MAT;
    out.p_img = args.lu_img + m * G::MAT;   // the packed tile, at the even step
    out.s_out = args.s_arr + m;
    out.status = args.status;
    out.pade_policy = args.pade_policy;
    out.pack8 = 1;
    out.q_img2 = second ? args.q_img + (m + 1) * G::MAT : nullptr;
    out.s_out2 = second ? args.s_arr + m + 1 : nullptr;
    const double* ctl_b = args.controls + (size_t)b * args.nc * args.K;
    const double dt = args.dt;
    const int K = args.K;
    auto gen = [&](CMat<1>& a) {
        // register r of lane (q, c): tile element (4 r + q, c); inside block blk = c / 8 it is element
        // (row % 8, c % 8) of the generator of step + blk, which the C-image holds in register (row % 8) / 4
        // of lane ((row % 8) % 4, c % 8); outside the diagonal blocks zero
        const int blk = c >> 3;
        const int st = step + ((blk == 1 && second) ? 1 : 0);
        const StepInterp si = args.interp[st];
        const size_t tsel = (args.nt == 1) ? 0 : (size_t)st;
        const double2* h0 = args.h0_cimg + tsel * G::MAT;
        const double2* g = args.g_cimg + tsel * K * G::MAT;
        double hre[4], him[4];
        int src[4];
        bool on[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * r + q;
            on[r] = (row >> 3) == blk && (blk == 0 || second);
            const int lr = row & 7;
            src[r] = (lr >> 2) * 64 + (lr & 3) * 16 + (c & 7);
            const double2 e = h0[src[r]];
            hre[r] = e.x;
            him[r] = e.y;
        }
        for (int k = 0; k < K; ++k) {
            const double uk = control_at(ctl_b, si, K, k);
            const double2* gk = g + (size_t)k * G::MAT;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double2 e = gk[src[r]];
                hre[r] += uk * e.x;
                him[r] += uk * e.y;
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            a.re[0][0][r] = on[r] ? dt * him[r] : 0.0;
            a.im[0][0][r] = on[r] ? -dt * hre[r] : 0.0;
        }
    };
    pade_pq_body<1, HERM>(gen, out, smem);
}

// U = P^-1 Q, in place of Q (round 5, one control set at a time: the sweep of such an evaluation is a chain
// of matrix-vector products and nothing else, qocx_sweepi.hip - with the propagator itself in the image a
// sub-step is ONE product instead of two). One wave per step; P^-1 is the left operand (LDS planes), Q the
// right one (C layout), both read from their column-major images.
template <int NB>
__global__ __launch_bounds__(64) void umul_kernel(LuArgs args, double2* q_all, double2* qt_all, unsigned count) {
    typedef Geo<NB> G;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* lre = reinterpret_cast<double*>(smem);
    double* lim = lre + G::PLANE;
    double* lsum = lim + G::PLANE;
    const unsigned w = blockIdx.x;
    if (w >= count) return;
    const size_t m = (size_t)(w / args.seg_len) * args.nsteps + args.step0 + w % args.seg_len;
    const double2* pinv = args.lu_img + m * G::MAT;
    double2* qimg = q_all + m * G::MAT;
    const int q = lane_id() >> 4, c = lane_id() & 15;
    CMat<NB> a, b;
#pragma unroll
    for (int ti = 0; ti < NB; ++ti)
#pragma unroll
        for (int tj = 0; tj < NB; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r) {  // element (16 ti + 4 r + q, 16 tj + c) of a column-major image
                const int at = (16 * tj + c) * G::NP + 16 * ti + 4 * r + q;
                const double2 e = pinv[at], f = qimg[at];
                a.re[ti][tj][r] = e.x;
                a.im[ti][tj][r] = e.y;
                b.re[ti][tj][r] = f.x;
                b.im[ti][tj][r] = f.y;
            }
    cmat_to_lds3<NB>(a, lre, lim, lsum);
    wave_sync();
    CAcc3<NB> acc;
    acc3_zero<NB>(acc);
    zgemm3_acc<NB, false>(acc, lre, lim, lsum, [&](int kk, int tj, double& bre, double& bim) {
        bre = b.re[kk >> 2][tj][kk & 3];
        bim = b.im[kk >> 2][tj][kk & 3];
    });
    CMat<NB> u;
    acc3_finish<NB>(u, acc);
    wave_sync();
    cmat_to_lds<NB>(u, lre, lim);
    wave_sync();
    lds_to_image<NB>(lre, lim, qimg);
    {   // U^T as well (the adjoint sweep's image): element (i, k) of the image is U[k][i]
        double2* timg = qt_all + m * G::MAT;
        const int lane = lane_id(), i = lane % G::NP, h = lane / G::NP;
#pragma unroll
        for (int cc = 0; cc < G::CPL; ++cc) {
            const int off = (cc * G::H + h) * G::PITCH + i;
            timg[cc * 64 + lane] = make_double2(lre[off], lim[off]);
        }
    }
}

// Explicit-generator variant: a[count][n][n] row-major complex in HBM (debug entry point; also
// the form a Magnus M4/M6 generator kernel would feed).
template <int NB, bool HERM>
__global__ __launch_bounds__(64) void pade_pq_explicit_kernel(const double2* a_in, int n,
                                                              FactorArgs args) {
    typedef Geo<NB> G;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const size_t m = (size_t)(blockIdx.x / args.seg_len) * args.nsteps + args.step0 +
                     blockIdx.x % args.seg_len;
    const int lane = lane_id();
    const int q = lane >> 4, c = lane & 15;
    PqOut out;
    out.q_img = args.q_img + m * G::MAT;
    out.p_img = args.lu_img + m * G::MAT;
    out.s_out = args.s_arr + m;
    out.status = args.status;
    out.pade_policy = args.pade_policy;
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
    pade_pq_body<NB, HERM>(gen, out, smem);
}

}  // namespace qocx

#include "qocx_sweep_core.h"

namespace qocx {

// W waves per seed: the S states of a seed are dealt out to the waves of its workgroup (state s to
// wave s % W). The operands of a step (Q, LU, 1/U_kk, perm) are fetched once, by wave 0, into LDS
// buffers all waves read; a workgroup barrier at every step start says "the operands have landed
// and every wave has left the previous step". The costs couple the states, so wave 0 evaluates
// them on all S states between two barriers. W = 1 is the single-state form: no barrier at all.
//
// LOADER: one more wave per seed does nothing but fetch: it issues the next step's 34 LDS-DMA
// pieces back to back, waits for them and meets the compute waves at the step barrier. An LDS-DMA
// costs its issuing wave 60-100 cycles of issue time each (MI355X_MICROARCH.md, 'LDS-DMA piece'),
// i.e. 2 000 - 3 400 cycles per step when the compute wave issues them from inside its dependent
// chains (the form without LOADER, kept for comparison: qocx_debug_set_knob "sweep_loader" 0).
//
// ONEBUF (one state, one wave per seed): ONE set of operand buffers instead of two, and TWO seeds
// per workgroup (two independent waves, no barrier between them). A step's LU image is copied to
// registers at the top of the step and its Q image is last read by the matvec of the last
// squaring sub-step (forward) - so the next step's operands can land in the SAME buffers while
// the solves run. The adjoint reads Q at the END of a sub-step (lambda = Q^H x): there the fetch
// of a step brings the step's OWN Q image (first solve, waited for with a counted vmcnt in front
// of the matvec) and the NEXT step's LU image. 35 KiB of LDS per seed instead of 68: two seeds
// share a CU, which halves the CUs on which the sweep displaces a K1a workgroup (a 221-register
// sweep wave leaves room for three two-wave K1a workgroups instead of four - whether the CU hosts
// one sweep wave or two).
template <int NB, int W, bool LOADER, bool ONEBUF = false, int NA = Geo<NB>::NP>
__global__ __launch_bounds__(ONEBUF ? 128 : 64 * (W + (LOADER ? 1 : 0))) void sweep_kernel(SweepArgs args) {
    typedef Geo<NB> G;
    static_assert(NA == G::NP || (NB == 4 && NA == 48), "NA < NP: the nine-tile images of 33 <= n <= 48");
    static_assert(!ONEBUF || (W == 1 && !LOADER && SweepPrefetch<NB>::value), "ONEBUF: one wave per seed");
    // With a loader wave the operands of TWO steps travel at once (ring of three buffers): the
    // fetch of step t+2 is issued while step t computes and has until the start of step t+2 to land.
    // NB = 4 (33 <= n <= 64): one Q and one LU image are 64 KiB each, so there is room for ONE
    // set of operands only - the step's fetch is issued at its start and waited for (PREFETCH off).
    constexpr bool PREFETCH = SweepPrefetch<NB>::value;
    constexpr bool LDSCOEF = SweepLdsCoef<NB>::value;
    // two states of a seed at a time on a wave (tri_solve2, lds_matvec2) where a wave has several:
    // the register-row solves of the multi-state forms
    constexpr bool PAIRS = !ONEBUF && !LDSCOEF && !LOADER && W > 1;
    static_assert(PREFETCH || !LOADER, "the loader variant needs a ring of buffers");
    constexpr int NBUF = LOADER ? 3 : ((PREFETCH && !ONEBUF) ? 2 : 1);
    typedef SweepLds<NB, NBUF, NA> L;
    constexpr int NP = G::NP, H = G::H, MAT = G::MAT;
    constexpr int LMAT = NA * NP;  // complex per image in LDS (NA of its NP columns)
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    // ONEBUF: wave v of the workgroup is seed 2 * blockIdx.x + v, with LDS of its own
    const int pack_wave = ONEBUF ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : 0;
    char* smem = smem_raw + (ONEBUF ? pack_wave * L::bytes_static(1) : 0);
    // the sweep is the serial chain of the evaluation: where it shares a SIMD with a wave of the
    // throughput kernels (two-wave K1a, K3) its instructions go first
    __builtin_amdgcn_s_setprio(3);
    double2* qbuf = reinterpret_cast<double2*>(smem + L::Q_OFF);
    double2* lbuf = reinterpret_cast<double2*>(smem + L::L_OFF);
    double2* dbuf = reinterpret_cast<double2*>(smem + L::D_OFF);
    int* pbuf = reinterpret_cast<int*>(smem + L::P_OFF);
    constexpr bool MULTI = (W > 1) || LOADER;  // more than one wave in the workgroup
    const int w = MULTI ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : 0;
    // the wave that fetches: the extra one, or compute wave 0 from inside its solves
    const bool fetcher = LOADER ? (w == W) : (w == 0);
    const bool computes = !LOADER || (w < W);
    double2* tmp = reinterpret_cast<double2*>(smem + L::TMP_OFF) + (computes ? w : 0) * L::TMPV * NP;
    double2* vecs = reinterpret_cast<double2*>(smem + L::VEC_OFF);
    auto block_sync = [&]() {
        if constexpr (LOADER) {
            // LDS hand-off only: a __syncthreads() also waits for vmcnt(0), i.e. for the fetch the
            // loader has just issued - which is what made the first loader variant slower
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        } else if constexpr (MULTI) {
            // (measured: a raw s_barrier without the vmcnt drain changes nothing here - at S > 1
            // the evaluation is bound by K3 and K1b on the compute stream, not by the sweep)
            __syncthreads();
        } else {
            wave_sync();
        }
    };
    // the loader's wait at the top of a step: everything but the fetch that is one step ahead
    auto wait_landed = [&](bool younger_in_flight) {
        if constexpr (LOADER) {
            if (younger_in_flight) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (G::MAT / 64) + 2) : "memory");
                return;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    // the loader wave runs the same control flow (every barrier) with empty state loops
    const int S = args.S, s0 = computes ? w : args.S;
    double2* lam = vecs + S * NP;
    const int b = ONEBUF ? (int)(blockDim.x >> 6) * blockIdx.x + pack_wave : blockIdx.x;
    if (ONEBUF && b >= args.batch) return;  // odd batch: the last workgroup has one seed
    const int lane = lane_id(), i = lane % NP, h = lane / NP;
    const int nsteps = args.nsteps;
    const size_t cap = args.slot_cap;
    double2* states_b = args.states + (size_t)b * cap * S * NP;
    double2* xs_b = args.xs + (size_t)b * cap * S * NP;
    int* offs_b = args.offs + (size_t)b * (nsteps + 1);
    const bool g0 = (h == 0);
    const size_t m0 = (size_t)b * nsteps;

    const int jb = args.j_begin, je = args.j_end;
    const bool do_fwd = (args.phase & 1) != 0, do_bwd = (args.phase & 2) != 0;
    if (jb > 0 || !do_fwd)
        if ((*(volatile int*)args.status) & 4) return;  // an earlier segment overflowed

    double cost = 0;
    int slot = 0;
    bool overflow = false;
    if (do_fwd) {
        if (jb == 0) {
            for (int s = s0; s < S; s += W)
                if (g0) {
                    const double2 p = args.psi0[s * NP + i];
                    vecs[s * NP + i] = p;
                    states_b[(size_t)s * NP + i] = p;
                }
        } else {  // resume: states, slot counter and partial cost left by the previous segment
            slot = offs_b[jb];
            cost = args.cost_out[b];
            for (int s = s0; s < S; s += W)
                if (g0) vecs[s * NP + i] = states_b[((size_t)slot * S + s) * NP + i];
        }
        wave_sync();
    }
    StepRegs<NB> r;
    if (QOCX_DBG_BITS(args.dbg) & 8192) {  // (timing experiment: the rows are never loaded)
#pragma unroll
        for (int c = 0; c < NP; ++c) r.lre[c] = r.lim[c] = 0.0;
    }
    const double2* qcur = qbuf;  // Q image of the step being computed
    const double2* lcur = lbuf;  // LU image of the step (read stage by stage when !PREFETCH)
    int permv = 0;               // perm[lane] of the step (adjoint, !PREFETCH)
    constexpr int MVB = Geo<NB>::CPL < 4 ? Geo<NB>::CPL : 4;  // (matrix, vector) LDS read pairs in flight

    // Everything a step needs arrives by LDS-DMA one step ahead: Q and LU images (the adjoint
    // gathers the transposed images), 1/U_kk, perm | iperm: 2*MAT/64 + 2 pieces. They are issued
    // one per column of the current step's triangular solves, into the bubbles of the
    // dependent readlane -> fma chain (an in-order wave cannot fill them otherwise).
    constexpr int IMG_PIECES = LMAT / 64, PIECES = 2 * IMG_PIECES + 2 + (NP > 32 ? 1 : 0);
    constexpr int PINTS = L::PINTS;
    // Per-lane global base addresses of the step being fetched; the pieces of an image are
    // reached through the instruction offset (dma16_imm), so a step needs six addresses, not one
    // per piece. Plain image: piece j is the KiB at j * 1024, two bases per image, each in the
    // middle of eight pieces (offsets -4096 .. 3072). Transposed image (adjoint): lane l of
    // piece j reads element ((l % NP) * NP + l / NP) + j * H, one base per image.
    constexpr int GROUP = IMG_PIECES >= 8 ? 8 : IMG_PIECES, CENTER = IMG_PIECES >= 8 ? 4 : 0;
    constexpr int NGROUP = IMG_PIECES / GROUP;
    const char* pf_q[NGROUP];
    const char* pf_l[NGROUP];
    const double2* pf_d = nullptr;
    const int* pf_p = nullptr;
    const int* pf_p2 = nullptr;  // NP = 64: perm and iperm are a piece each
    int pf_par = 0;
    bool pf_adjoint = false, pf_due = false;
    // what the hooks of the running (sub-step, state) issue: the Q pieces / everything else.
    // Two buffers: both follow pf_due in the first (sub-step, state) of a step. ONEBUF: see there.
    bool pf_fire_q = false, pf_fire_l = false;
    bool pf_qdue = false;  // ONEBUF adjoint: the step's own Q image is to be fetched
    // m: step whose LU image, 1/U_kk and permutation are fetched; mq: step of the Q image
    auto set_prefetch = [&](size_t m, int par, bool adjoint, size_t mq) __attribute__((always_inline)) {
        pf_par = par;
        pf_adjoint = adjoint;
        const size_t el = adjoint ? (size_t)(lane % NP) * NP + lane / NP : (size_t)lane;
#pragma unroll
        for (int g = 0; g < NGROUP; ++g) {
            const size_t mid = adjoint ? 0 : (size_t)(g * GROUP + CENTER) * 64;
            pf_q[g] = reinterpret_cast<const char*>(args.q_img + mq * MAT + el + mid);
            pf_l[g] = reinterpret_cast<const char*>(args.lu_img + m * MAT + el + mid);
        }
        pf_d = args.dinv + m * NP + i;
        if constexpr (NP > 32) {
            pf_p = args.perm + m * NP + lane;
            pf_p2 = args.iperm + m * NP + lane;
        } else {
            pf_p = (lane < 32 ? args.perm : args.iperm) + m * NP + (lane & 31) % NP;
        }
    };
    auto dma_image = [&](auto J, const char* const (&base)[NGROUP], double2* buf) __attribute__((always_inline)) {
        constexpr int j = decltype(J)::value;
        char* dst = reinterpret_cast<char*>(buf + pf_par * LMAT + j * 64);
        if (pf_adjoint) dma16_imm<j * H * 16>(base[0], dst);
        else dma16_imm<(j - ((j / GROUP) * GROUP + CENTER)) * 1024>(base[j / GROUP], dst);
    };
    auto dma_one = [&](auto PIECE) __attribute__((always_inline)) {
        constexpr int piece = decltype(PIECE)::value;
        if constexpr (piece < IMG_PIECES) {
            dma_image(std::integral_constant<int, piece>(), pf_q, qbuf);
        } else if constexpr (piece < 2 * IMG_PIECES) {
            dma_image(std::integral_constant<int, piece - IMG_PIECES>(), pf_l, lbuf);
        } else if constexpr (piece == 2 * IMG_PIECES) {
            dma16(pf_d, dbuf + pf_par * 64);
        } else if constexpr (piece == 2 * IMG_PIECES + 1) {
            dma4(pf_p, pbuf + pf_par * PINTS);
        } else if constexpr (piece == 2 * IMG_PIECES + 2 && NP > 32) {
            dma4(pf_p2, pbuf + pf_par * PINTS + PINTS / 2);
        }  // (NP = 16: hook_a also passes piece numbers >= PIECES, which fetch nothing)
    };
    // (always_inline: as a real call - what NB = 4 with its 131 pieces otherwise becomes - the closure
    // lives in scratch and the LDS-DMA destinations are no longer compile-time address-space known)
    auto issue_dma = [&](size_t m, int par, bool adjoint) __attribute__((always_inline)) {
        set_prefetch(m, par, adjoint, m);
        for_each_const(dma_one, std::make_integer_sequence<int, PIECES>{});
    };
    // a piece under the flags of the running (sub-step, state)
    auto dma_fire = [&](auto PIECE) __attribute__((always_inline)) {
        if constexpr (decltype(PIECE)::value < IMG_PIECES) {
            if (pf_fire_q && !(QOCX_DBG_BITS(args.dbg) & 1024)) dma_one(PIECE);  // (dbg: timing experiment)
        } else {
            if (pf_fire_l) dma_one(PIECE);
        }
    };
    auto issue_lu_only = [&](size_t m, bool adjoint) __attribute__((always_inline)) {  // ONEBUF adjoint
        set_prefetch(m, 0, adjoint, m);
        pf_fire_q = false;
        pf_fire_l = true;
        for_each_const(dma_fire, std::make_integer_sequence<int, PIECES>{});
        pf_fire_l = false;
    };
    auto hook_a = [&](auto KK) __attribute__((always_inline)) {  // first solve: pieces 0 .. NP-2
        if constexpr (!LOADER && PREFETCH) dma_fire(KK);
    };
    auto hook_b = [&](auto KK) __attribute__((always_inline)) {  // second solve: the remaining pieces
        constexpr int piece = NP - 1 + decltype(KK)::value;
        if constexpr (!LOADER && PREFETCH && piece < PIECES)
            dma_fire(std::integral_constant<int, piece>());
    };
    auto finish_prefetch = [&]() {  // pieces that did not fit into the two solves (NP = 16)
        constexpr int DONE = 2 * (NP - 1), REST = PIECES > DONE ? PIECES - DONE : 0;
        if constexpr (LOADER || !PREFETCH) return;
        for_each_const(
            [&](auto P) __attribute__((always_inline)) {
                dma_fire(std::integral_constant<int, DONE + decltype(P)::value>());
            },
            std::make_integer_sequence<int, REST>{});
        if (pf_fire_l || pf_fire_q) {
            pf_due = false;
            pf_qdue = false;
        }
        pf_fire_q = false;
        pf_fire_l = false;
    };
    auto issue_all_due = [&]() {  // LOADER: the whole step at once, from the wave that only fetches
        if (pf_due) for_each_const(dma_one, std::make_integer_sequence<int, PIECES>{});
        pf_due = false;
    };
    auto scalars = [&](int par, bool adjoint) {
        StepScalars sc;
        sc.dv = dbuf[par * 64 + i];
        sc.pm = min(max(pbuf[par * PINTS + (adjoint ? PINTS / 2 : 0) + i], 0), NP - 1);
        return sc;
    };

    // one propagator step on all S states with the operands in `r`
    auto forward_step = [&](const StepScalars& sc, int nsub) {
        for (int sub = 0; sub < nsub; ++sub) {
            if ((size_t)slot + 1 >= cap) {
                overflow = true;
                break;
            }
            int s = s0;
            if constexpr (PAIRS) {
                // two states at a time (tri_solve2): same arithmetic per state, bit for bit
                for (; s + W < S; s += 2 * W) {
                    pf_fire_q = pf_fire_l = pf_due;
                    double are, aim, bre, bim;
                    lds_matvec2<NB, false, MVB>(qcur, vecs + s * NP, vecs + (s + W) * NP,
                                                h * NP + sc.pm, h, are, aim, bre, bim);
                    tri_solve2<NB, true, false>(r.lre, r.lim, are, aim, bre, bim, hook_a);
                    cscale(are, aim, sc.dv);
                    cscale(bre, bim, sc.dv);
                    tri_solve2<NB, false, false>(r.lre, r.lim, are, aim, bre, bim, hook_b);
                    finish_prefetch();
                    wave_sync();
                    {
                        const double2 pa = make_double2(are, aim), pb = make_double2(bre, bim);
                        vecs[s * NP + i] = pa;
                        vecs[(s + W) * NP + i] = pb;
                        states_b[((size_t)(slot + 1) * S + s) * NP + i] = pa;
                        states_b[((size_t)(slot + 1) * S + s + W) * NP + i] = pb;
                    }
                    wave_sync();
                }
            }
            for (; s < S; s += W) {
                // the next step's operands: ONEBUF into the buffers of this step, once its last
                // matvec has read Q; else into the other set during the first (sub-step, state)
                pf_fire_q = pf_fire_l = pf_due && (!ONEBUF || sub == nsub - 1);
                // z = Pi (Q psi): the lane at position i takes row perm[i] of the Q image
                double zre, zim;
                if (QOCX_DBG_BITS(args.dbg) & 4096) {  // (timing experiment: no matrix-vector product)
                    const double2 e = vecs[s * NP + i];
                    zre = e.x; zim = e.y;
                } else {
                lds_matvec<NB, false, MVB, NA>(qcur, vecs + s * NP, h * NP + sc.pm, h, zre, zim);
                }
                if (!(QOCX_DBG_BITS(args.dbg) & 2048)) {  // (timing experiment: no solves)
                if constexpr (!LDSCOEF) tri_solve<NB, true, false>(r.lre, r.lim, zre, zim, hook_a);
                else tri_solve_lds<NB, true, false, false, NA>(lcur, sc.pm, i, permv, zre, zim, hook_a);
                }
                cscale(zre, zim, sc.dv);
                if (!(QOCX_DBG_BITS(args.dbg) & 2048)) {
                if constexpr (!LDSCOEF) tri_solve<NB, false, false>(r.lre, r.lim, zre, zim, hook_b);
                else tri_solve_lds<NB, false, false, false, NA>(lcur, sc.pm, i, permv, zre, zim, hook_b);
                }
                finish_prefetch();
                wave_sync();
                {   // every lane group holds the same z: all of them store (no exec-mask branch
                    // in the serial chain)
                    const double2 p = make_double2(zre, zim);
                    vecs[s * NP + i] = p;
                    states_b[((size_t)(slot + 1) * S + s) * NP + i] = p;
                }
                wave_sync();
            }
            ++slot;
        }
    };
    auto before_step = [&](int step) {  // called behind a barrier: every state of `step` is in vecs
        // (has_step_costs: without step costs eval_costs would still walk the cost table in HBM,
        // a dependent trip to memory in front of every step when cost_eval_step = 1)
        if (step != 0 && args.has_step_costs && (step % args.cost_eval_step) == 0) {
            if (w == 0) cost += eval_costs<NB>(args, true, false, vecs, nullptr, h, i);
            if constexpr (MULTI) __syncthreads();  // the other waves overwrite their states next
        }
        if (g0 && args.step_states != nullptr)
            for (int s = s0; s < S; s += W)
                args.step_states[(((size_t)b * (nsteps + 1) + step) * S + s) * NP + i] =
                    vecs[s * NP + i];
        if (w == 0 && lane == 0) offs_b[step] = slot;
    };

    // ---- forward sweep: the next step's operands stream into LDS while the current step's
    // dependent chains run -------------------------------------------------------------------
    if (do_fwd) {
        if (fetcher) issue_dma(m0 + jb, 0, false);
        if (LOADER && fetcher && jb + 1 < je) issue_dma(m0 + jb + 1, 1, false);
        int nsub_next = 1 << step_squarings(args.s_arr[m0 + jb]);
        for (int step = jb; step < je; ++step) {
            const int par = (step - jb) % NBUF;
            const int nsub = nsub_next;
            if constexpr (!PREFETCH) {
                if (step > jb) {
                    if constexpr (MULTI) __syncthreads();  // every wave has left the previous operands
                    if (fetcher) issue_dma(m0 + step, 0, false);
                }
            }
            if (fetcher) wait_landed(step + 1 < je);
            block_sync();
            const StepScalars sc = scalars(par, false);
            if constexpr (!LDSCOEF) {
                if (computes && !(QOCX_DBG_BITS(args.dbg) & 8192))  // (timing experiment: stale rows)
                    lds_to_regs<NB, false>(qbuf + par * LMAT, lbuf + par * LMAT, pbuf + par * PINTS, r,
                                           sc.pm, lane, i);
            } else {
                lcur = lbuf + par * LMAT;
            }
            qcur = qbuf + par * LMAT;
            wave_sync();
            if constexpr (LOADER) {
                pf_due = fetcher && (step + 2 < je);
                if (pf_due) set_prefetch(m0 + step + 2, (par + 2) % NBUF, false, m0 + step + 2);
                issue_all_due();
            } else if constexpr (PREFETCH) {
                pf_due = fetcher && (step + 1 < je) && !(QOCX_DBG_BITS(args.dbg) & 256);  // (dbg: timing experiment)
                if (pf_due) set_prefetch(m0 + step + 1, (par + 1) % NBUF, false, m0 + step + 1);
            }
            if (step + 1 < je) nsub_next = 1 << step_squarings(args.s_arr[m0 + step + 1]);
            before_step(step);
            forward_step(sc, nsub);
            if (overflow) break;
        }
    }
    if (overflow) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (w == 0 && lane == 0) atomicOr(args.status, 4);
        return;
    }
    if (do_fwd) {
        block_sync();  // every wave has finished the last step of the segment
        if (je == nsteps) {
            before_step(nsteps);
            if (w == 0) cost += eval_costs<NB>(args, false, true, vecs, nullptr, h, i);
            if (w == 0 && args.unit_adjoint && args.want_grad) unit_adjoint_scales<NB>(args, vecs, b, h, i);
            if (g0)
                for (int s = s0; s < S; s += W)
                    args.final_out[((size_t)b * S + s) * NP + i] = vecs[s * NP + i];
        } else if (w == 0 && lane == 0) {
            offs_b[je] = slot;  // the next segment resumes from here
        }
        if (w == 0 && lane == 0) args.cost_out[b] = cost;
    }
    if (!do_bwd) return;

    // lambda += host-supplied cotangent of the states at system step `step`, if there is one
    auto inject = [&](int step) {
        if (args.inj_index == nullptr) return;
        const int row = args.inj_index[step];
        if (row < 0) return;
        if (g0)
            for (int s = s0; s < S; s += W) {
                const double2 e = args.inj_bars[(((size_t)b * args.inj_count + row) * S + s) * NP + i];
                double2 l = lam[s * NP + i];
                l.x += e.x;
                l.y += e.y;
                lam[s * NP + i] = l;
            }
        wave_sync();
    };

    // ---- adjoint sweep ---------------------------------------------------------------------
    const bool unit = args.unit_adjoint != 0;
    int* offs_x = unit ? args.offs_x + (size_t)b * (nsteps + 1) : nullptr;
    if (je == nsteps && unit) {
        // lam = the targets. The forward sweep may not have numbered the sub-steps yet: the xs
        // slots are counted down from the capacity and recorded per step in offs_x
        unit_adjoint_seed<NB>(args, lam, s0, W, h, i);
        slot = (int)cap;
        block_sync();
    } else if (je == nsteps) {
        if (!do_fwd) {  // final states of the forward segments
            slot = offs_b[nsteps];
            for (int s = s0; s < S; s += W)
                if (g0) vecs[s * NP + i] = states_b[((size_t)slot * S + s) * NP + i];
        }
        for (int s = s0; s < S; s += W)
            if (g0) lam[s * NP + i] = make_double2(0, 0);
        block_sync();
        // cotangent seeds on the final states: non-step costs, and step costs if the final step
        // is a cost step (schroedingerdiscrete.py:412-416 evaluates them before the loop ends).
        if (w == 0)
            (void)eval_costs<NB>(args, (nsteps % args.cost_eval_step) == 0, true, vecs, lam, h, i);
        block_sync();
        inject(nsteps);
    } else {  // resume the adjoint sweep below step je
        slot = unit ? offs_x[je] : offs_b[je];
        for (int s = s0; s < S; s += W)
            if (g0) lam[s * NP + i] = args.lam_buf[((size_t)b * S + s) * NP + i];
        wave_sync();
    }

    auto adjoint_step = [&](const StepScalars& sc, int nsub, int step) {
        for (int sub = nsub - 1; sub >= 0; --sub) {
            if (slot <= 0) {  // (unit adjoint: nobody has checked the capacity before)
                overflow = true;
                break;
            }
            --slot;
            int s = s0;
            if constexpr (PAIRS) {
                for (; s + W < S; s += 2 * W) {
                    pf_fire_l = pf_fire_q = pf_due;
                    const double2 la = lam[s * NP + i], lb = lam[(s + W) * NP + i];
                    double are = la.x, aim = la.y, bre = lb.x, bim = lb.y;
                    tri_solve2<NB, true, true>(r.lre, r.lim, are, aim, bre, bim, hook_a);
                    cscale_conj(are, aim, sc.dv);
                    cscale_conj(bre, bim, sc.dv);
                    tri_solve2<NB, false, true>(r.lre, r.lim, are, aim, bre, bim, hook_b);
                    finish_prefetch();
                    const double xar = __shfl(are, sc.pm), xai = __shfl(aim, sc.pm);
                    const double xbr = __shfl(bre, sc.pm), xbi = __shfl(bim, sc.pm);
                    wave_sync();
                    {
                        const double2 xa = make_double2(xar, xai), xb = make_double2(xbr, xbi);
                        tmp[i] = xa;
                        tmp[NP + i] = xb;
                        xs_b[((size_t)slot * S + s) * NP + i] = xa;
                        xs_b[((size_t)slot * S + s + W) * NP + i] = xb;
                    }
                    wave_sync();
                    double yar, yai, ybr, ybi;
                    lds_matvec2<NB, true, MVB>(qcur, tmp, tmp + NP, lane, h, yar, yai, ybr, ybi);
                    wave_sync();
                    lam[s * NP + i] = make_double2(yar, yai);
                    lam[(s + W) * NP + i] = make_double2(ybr, ybi);
                    wave_sync();
                }
            }
            for (; s < S; s += W) {
                // (ONEBUF: the first sub-step processed fetches the step's own Q and the next
                // step's LU image; pf_due / pf_qdue are cleared by finish_prefetch)
                pf_fire_l = pf_due;
                pf_fire_q = ONEBUF ? pf_qdue : pf_due;
                const bool q_in_flight = ONEBUF && pf_fire_q, l_in_flight = pf_fire_l;
                const double2 l0 = lam[s * NP + i];
                double zre = l0.x, zim = l0.y;
                // P^H = U'^H D^H L^H Pi : U'^H a = lambda ; b = a / conj(U_kk) ; L^H v = b
                if (!(QOCX_DBG_BITS(args.dbg) & 2048)) {  // (timing experiment: no solves)
                if constexpr (!LDSCOEF) tri_solve<NB, true, true>(r.lre, r.lim, zre, zim, hook_a);
                else tri_solve_lds<NB, true, true, true, NA>(lcur, sc.pm, i, permv, zre, zim, hook_a);
                }
                cscale_conj(zre, zim, sc.dv);
                if (!(QOCX_DBG_BITS(args.dbg) & 2048)) {
                if constexpr (!LDSCOEF) tri_solve<NB, false, true>(r.lre, r.lim, zre, zim, hook_b);
                else tri_solve_lds<NB, false, true, true, NA>(lcur, sc.pm, i, permv, zre, zim, hook_b);
                }
                finish_prefetch();
                // x = Pi^T v : x_i = v[position of row i]
                const double xre = __shfl(zre, sc.pm), xim = __shfl(zim, sc.pm);
                wave_sync();
                {
                    const double2 x = make_double2(xre, xim);
                    tmp[i] = x;
                    xs_b[((size_t)slot * S + s) * NP + i] = x;
                }
                wave_sync();
                if constexpr (ONEBUF) {
                    // the Q pieces went out first (hook_a); younger than them: the LU image, 1/U_kk
                    // and the permutation of the next step, and the store of x just above
                    if (q_in_flight) {
                        if (l_in_flight) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IMG_PIECES + 3) : "memory");
                        else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
                    }
                }
                // lambda = Q^H x ; the LDS image is that of Q^T (lane (h,i): Q[cc*H+h][i])
                double yre, yim;
                if (QOCX_DBG_BITS(args.dbg) & 4096) {  // (timing experiment: no matrix-vector product)
                    yre = xre; yim = xim;
                } else {
                lds_matvec<NB, true, MVB, NA>(qcur, tmp, lane, h, yre, yim);
                }
                wave_sync();
                lam[s * NP + i] = make_double2(yre, yim);
                wave_sync();
            }
        }
        if (step != 0 && (step % args.cost_eval_step) == 0 && args.has_step_costs) {
            // step costs were evaluated on the states *before* evolving from `step`
            if (g0)
                for (int s = s0; s < S; s += W)
                    vecs[s * NP + i] = states_b[((size_t)slot * S + s) * NP + i];
            block_sync();
            if (w == 0) (void)eval_costs<NB>(args, true, false, vecs, lam, h, i);
            block_sync();
        }
        if (step != 0) inject(step);
        if (unit && w == 0 && lane == 0) offs_x[step] = slot;
    };
    {
        const size_t ml = m0 + je - 1;
        if constexpr (ONEBUF) issue_lu_only(ml, true);
        else if (fetcher) issue_dma(ml, 0, true);
        if (LOADER && fetcher && je - 2 >= jb) issue_dma(ml - 1, 1, true);
        int nsub_next = 1 << step_squarings(args.s_arr[ml]);
        for (int step = je - 1, it = 0; step >= jb; --step, ++it) {
            const int par = it % NBUF;
            const int nsub = nsub_next;
            if constexpr (!PREFETCH) {
                if (it > 0) {
                    if constexpr (MULTI) __syncthreads();
                    if (fetcher) issue_dma(m0 + step, 0, true);
                }
            }
            if (fetcher) wait_landed(step - 1 >= jb);
            block_sync();
            const StepScalars sc = scalars(par, true);
            if constexpr (!LDSCOEF) {
                if (computes && !(QOCX_DBG_BITS(args.dbg) & 8192))
                    lds_to_regs<NB, true>(qbuf + par * LMAT, lbuf + par * LMAT, pbuf + par * PINTS, r,
                                          sc.pm, lane, i);
            } else {
                lcur = lbuf + par * LMAT;
                permv = pbuf[par * PINTS + lane % NP];
            }
            qcur = qbuf + par * LMAT;
            wave_sync();
            if constexpr (LOADER) {
                pf_due = fetcher && (step - 2 >= jb);
                if (pf_due) set_prefetch(m0 + step - 2, (par + 2) % NBUF, true, m0 + step - 2);
                issue_all_due();
            } else if constexpr (ONEBUF) {
                pf_due = (step - 1 >= jb) && !(QOCX_DBG_BITS(args.dbg) & 512);
                pf_qdue = !(QOCX_DBG_BITS(args.dbg) & 512);
                set_prefetch(pf_due ? m0 + step - 1 : m0 + step, 0, true, m0 + step);
            } else if constexpr (PREFETCH) {
                pf_due = fetcher && (step - 1 >= jb);
                if (pf_due) set_prefetch(m0 + step - 1, par ^ 1, true, m0 + step - 1);
            }
            if (step - 1 >= jb) nsub_next = 1 << step_squarings(args.s_arr[m0 + step - 1]);
            adjoint_step(sc, nsub, step);
            if (overflow) break;
        }
    }
    if (overflow) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (w == 0 && lane == 0) atomicOr(args.status, 4);
        return;
    }
    if (jb > 0 && g0)
        for (int s = s0; s < S; s += W)
            args.lam_buf[((size_t)b * S + s) * NP + i] = lam[s * NP + i];
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
    static constexpr int TAU_OFF = 0;                            // 13 tau vectors
    static constexpr int V_OFF = TAU_OFF + 13 * G::NP * 16;      // 2 slots for the current rho
    static constexpr int BYTES = V_OFF + 2 * G::NP * 16;
};

// SKEW: every H0(t), G_k(t) is Hermitian, so a^H = -a exactly and the rows of a^H need no
// registers (the common physical case; the host checks it bit for bit).
//
// Measured and dropped: a variant that contracts abar with the G_k after every (sub-step, state)
// and rebuilds the generator, so that the two never coexist and three waves share a SIMD
// (<= 168 registers, 10 KiB LDS): no faster than two waves per SIMD of this form. What did pay
// (16 %) was removing the exec-mask branches around the LDS stores: after sum_groups every lane
// group holds the same values, so all of them store.
// NB = 4: a lane walks 64 columns per matvec. Without a fence the compiler hoists all 64 LDS reads
// of an unrolled loop above its FMAs (256 registers of operands on top of the 256 - 384 the
// generator and abar occupy) and spills; eight columns in flight are plenty.
template <int NB>
__device__ __forceinline__ void column_fence(int cc) {
    if constexpr (NB >= 4) {
        if ((cc & 7) == 7) asm volatile("" ::: "memory");
    }
}

// UMODE (KrylovArgs::umode, one control set at a time): a copy of its own, so that the kernels of the batched
// evaluation keep their registers (168: three waves to a SIMD; tests/test_build_resources.py)
template <int NB, bool EXPLICIT, bool SKEW, bool UMODE = false>
__device__ __forceinline__ void krylov_grad_body(const KrylovArgs& args, char* smem) {
    typedef Geo<NB> G;
    typedef KrylovLds<NB> L;
    constexpr int NP = G::NP, CPL = G::CPL, H = G::H;
    double2* vv = reinterpret_cast<double2*>(smem + L::V_OFF);
    double2* tau_l = reinterpret_cast<double2*>(smem + L::TAU_OFF);
    const int step = args.step0 + blockIdx.x, b = blockIdx.y;
    const int lane = lane_id(), i = lane % NP, h = lane / NP;
    const int nsteps = args.nsteps, S = args.S, K = args.K;
    const size_t m = (size_t)b * nsteps + step;
    const int sq = step_squarings(args.s_arr[m]);
    const int order = step_order(args.s_arr[m]);
    const double dts = args.dt * ldexp(1.0, -sq);
    // (direct: the step table - interpolated controls per step, no interpolation here)
    const StepInterp si = args.direct ? StepInterp{0, 0, 1.0, 0.0} : args.interp[step];
    const double* ctl_b = args.controls + (size_t)b * args.nc * K;
    const size_t tsel = (args.nt == 1) ? 0 : (size_t)step;
    const double2* h0r = args.h0_rimg + tsel * G::MAT;
    const double2* h0t = args.h0_timg + tsel * G::MAT;
    const double2* gr = args.g_rimg + tsel * K * G::MAT;
    const double2* gt = args.g_timg + tsel * K * G::MAT;

    // a (rows) and a^H (rows) of the scaled generator
    constexpr int HC = SKEW ? 1 : CPL;
    auto build = [&](double (&are)[CPL], double (&aim)[CPL], double (&hre)[HC],
                     double (&him)[HC]) __attribute__((always_inline)) {
        if constexpr (EXPLICIT) {
            // Magnus M4/M6: a = 2^-s M with M from magnus_fwd_kernel (row-major, padded)
            const double2* mm = args.m_rm + m * G::MAT;
            const double sc = ldexp(1.0, -sq);
#pragma unroll
            for (int cc = 0; cc < CPL; ++cc) {
                const double2 e = mm[(size_t)i * NP + cc * H + h];
                are[cc] = sc * e.x;
                aim[cc] = sc * e.y;
                if (!SKEW) {
                    const double2 f = mm[(size_t)(cc * H + h) * NP + i];
                    hre[cc] = sc * f.x;
                    him[cc] = -sc * f.y;
                }
            }
        } else {
            double xr[CPL], xi[CPL], tr_[HC], ti_[HC];
            const unsigned off = (unsigned)lane;
#pragma unroll
            for (int cc = 0; cc < CPL; ++cc) {
                const double2 e = (h0r + cc * 64)[off];
                xr[cc] = e.x;
                xi[cc] = e.y;
                if (!SKEW) {
                    const double2 f = (h0t + cc * 64)[off];
                    tr_[cc] = f.x;
                    ti_[cc] = f.y;
                }
                column_fence<NB>(cc);
            }
            for (int k = 0; k < K; ++k) {
                const double uk = args.direct ? ctl_b[(size_t)step * K + k] : control_at(ctl_b, si, K, k);
#pragma unroll
                for (int cc = 0; cc < CPL; ++cc) {
                    const double2 e = (gr + (size_t)k * G::MAT + cc * 64)[off];
                    xr[cc] += uk * e.x;
                    xi[cc] += uk * e.y;
                    if (!SKEW) {
                        const double2 f = (gt + (size_t)k * G::MAT + cc * 64)[off];
                        tr_[cc] += uk * f.x;
                        ti_[cc] += uk * f.y;
                    }
                    column_fence<NB>(cc);
                }
            }
#pragma unroll
            for (int cc = 0; cc < CPL; ++cc) {
                are[cc] = dts * xi[cc];   // a = -i dts H
                aim[cc] = -dts * xr[cc];
                if (!SKEW) {
                    hre[cc] = dts * ti_[cc];  // a^H[i][c] = conj(a[c][i]) = conj(-i dts H[c][i])
                    him[cc] = dts * tr_[cc];
                }
            }
        }
    };

    // One (sub-step, state). Phase A: tau_i = (a^H)^i x, i = 0..12, all kept in LDS. Phase B: the
    // rho_i by the Horner recurrence  rho_12 = b13 sigma,  rho_{i-1} = b_i w_i + a rho_i  with
    // w_i = sigma (i odd) or delta (i even) - one matvec per rho instead of the two chains
    // a^j sigma, a^j delta - and, as each rho_i appears, the rank-1 update abar += tau_i rho_i^H.
    // The matvec a rho_i and the rank-1 update read the same LDS broadcast of rho_i.
    // M: the Pade order K1a chose for this step (qocx_wave.h): chains of M terms, rho_{M-1} =
    // b_M sigma (M is odd), coefficients of the [M/M] approximant.
    // (Registers decide this kernel - three waves per SIMD at 168. A copy of the unrolled chains per
    // order costs 90 more, rolled loops with M as a run-time bound 60 more; so ONE unrolled copy
    // for order 13 whose iterations beyond M are skipped by wave-uniform branches.)
    const int M = order;
    const double* bt = pade_table(order);
    auto chains = [&](const double (&are)[CPL], const double (&aim)[CPL],
                      const double (&hre)[HC], const double (&him)[HC], double2 x, double2 p0,
                      double2 p1, double (&abr)[CPL], double (&abi)[CPL]) __attribute__((always_inline)) {
        const double sgr = p0.x + p1.x, sgi = p0.y + p1.y;
        const double dlr = p0.x - p1.x, dli = p0.y - p1.y;
        double tar = x.x, tai = x.y;
#pragma unroll
        for (int jj = 0; jj < 13; ++jj) {
            if (jj >= M) break;
            // (every lane group holds the same sums: all of them store, no exec-mask branch)
            tau_l[jj * NP + i] = make_double2(tar, tai);
            if (jj < M - 1) {
                wave_sync();
                double s2r = 0, s2i = 0;
#pragma unroll
                for (int cc = 0; cc < CPL; ++cc) {
                    const double2 v2 = tau_l[jj * NP + cc * H + h];
                    if (SKEW) {  // a^H = -a
                        s2r = fma(aim[cc], v2.y, fma(-are[cc], v2.x, s2r));
                        s2i = fma(-aim[cc], v2.x, fma(-are[cc], v2.y, s2i));
                    } else {
                        s2r = fma(-him[cc], v2.y, fma(hre[cc], v2.x, s2r));
                        s2i = fma(him[cc], v2.x, fma(hre[cc], v2.y, s2i));
                    }
                    column_fence<NB>(cc);
                }
                tar = sum_groups<NB>(s2r);
                tai = sum_groups<NB>(s2i);
            }
        }
        double rr = bt[M] * sgr, ri = bt[M] * sgi;
#pragma unroll
        for (int ii = 12; ii >= 0; --ii) {
            if (ii >= M) continue;
            double2* slot = vv + (ii & 1) * NP;  // two slots in turn: one sync per step
            slot[i] = make_double2(rr, ri);
            wave_sync();
            const double2 tv = tau_l[ii * NP + i];
            double s0r = 0, s0i = 0;
#pragma unroll
            for (int cc = 0; cc < CPL; ++cc) {
                const double2 r = slot[cc * H + h];
                // tau * conj(rho)
                abr[cc] = fma(tv.y, r.y, fma(tv.x, r.x, abr[cc]));
                abi[cc] = fma(-tv.x, r.y, fma(tv.y, r.x, abi[cc]));
                if (ii > 0) {
                    s0r = fma(-aim[cc], r.y, fma(are[cc], r.x, s0r));
                    s0i = fma(aim[cc], r.x, fma(are[cc], r.y, s0i));
                }
                column_fence<NB>(cc);
            }
            if (ii > 0) {
                const double coef = bt[ii];
                rr = fma(coef, (ii & 1) ? sgr : dlr, sum_groups<NB>(s0r));
                ri = fma(coef, (ii & 1) ? sgi : dli, sum_groups<NB>(s0i));
            }
        }
        wave_sync();  // tau_l and the slots are rewritten by the next (sub-step, state)
    };

    const size_t cap = args.slot_cap;
    const double2* states_b = args.states + (size_t)b * cap * S * NP;
    const double2* xs_b = args.xs + (size_t)b * cap * S * NP;
    const int t0 = args.offs[(size_t)b * (nsteps + 1) + step];
    const int nsub = 1 << sq;
    if (t0 < 0 || (size_t)t0 + (size_t)nsub >= cap) return;  // sweep overflowed (status bit 2)
    // unit adjoint: x sits at slots of its own and is scaled by the cost's scalar here
    const int tx = args.offs_x ? args.offs_x[(size_t)b * (nsteps + 1) + step] : t0;
    if (tx < 0 || (size_t)tx + (size_t)nsub > cap) return;

    double are[CPL], aim[CPL], hre[HC], him[HC];
    build(are, aim, hre, him);
    double abr[CPL], abi[CPL];
#pragma unroll
    for (int cc = 0; cc < CPL; ++cc) {
        abr[cc] = 0;
        abi[cc] = 0;
    }
    for (int sub = 0; sub < nsub; ++sub)
        for (int s = 0; s < S; ++s) {
            const size_t t = (size_t)t0 + sub;
            double2 x = xs_b[(((size_t)tx + sub) * S + s) * NP + i];
            if constexpr (UMODE) {
                // the adjoint sweep left lambda' (its cotangent BEFORE the step): x = P^-H lambda', from the
                // column-major image of P^-1 - lane (h, i): elements (cc H + h, i) of column i
                const double2* pin = args.pinv_img + m * G::MAT + (size_t)i * NP;
                vv[i] = x;
                wave_sync();
                double sr = 0, si_ = 0;
#pragma unroll
                for (int cc = 0; cc < CPL; ++cc) {
                    const double2 e = pin[cc * H + h];
                    const double2 l = vv[cc * H + h];
                    sr = fma(e.y, l.y, fma(e.x, l.x, sr));     // conj(e) * l
                    si_ = fma(-e.y, l.x, fma(e.x, l.y, si_));
                    column_fence<NB>(cc);
                }
                x = make_double2(sum_groups<NB>(sr), sum_groups<NB>(si_));
                wave_sync();
            }
            chains(are, aim, hre, him, x, states_b[(t * S + s) * NP + i],
                   states_b[((t + 1) * S + s) * NP + i], abr, abi);
        }
    if constexpr (EXPLICIT) {  // Mbar = 2^-s abar; the Magnus reverse kernel finishes the chain
        double2* mb = args.mbar_rm + m * G::MAT;
        const double sc = ldexp(1.0, -sq);
#pragma unroll
        for (int cc = 0; cc < CPL; ++cc)
            mb[(size_t)i * NP + cc * H + h] = make_double2(sc * abr[cc], sc * abi[cc]);
    } else {
        // g_k = Re <abar, E_k>, E_k = d a / d u_k = -i dts G_k  (H-bar = i dt M-bar, Appendix A)
        // unit adjoint: abar is that of the back-propagated target, abar = c abar_1 with the cost's
        // scalar c: g_k = Re(conj(c) gamma_k), gamma_k = sum conj(abar_1) E_k (real part as above)
        const bool unit = args.offs_x != nullptr;
        for (int k = 0; k < K; ++k) {
            double acc = 0, acc_im = 0;
#pragma unroll
            for (int cc = 0; cc < CPL; ++cc) {
                const double2 e = gr[(size_t)k * G::MAT + cc * 64 + lane];
                acc = fma(abi[cc], -dts * e.x, fma(abr[cc], dts * e.y, acc));
                acc_im = fma(abi[cc], -dts * e.y, fma(abr[cc], -dts * e.x, acc_im));
                column_fence<NB>(cc);
            }
            acc = wave_sum(acc);
            if (unit) {
                acc_im = wave_sum(acc_im);
                if (lane == 0) {
                    args.gstep[(m * K + k) * 2] = acc;
                    args.gstep[(m * K + k) * 2 + 1] = acc_im;
                }
            } else if (lane == 0) {
                args.gstep[m * K + k] = acc;
            }
        }
    }
}

template <int NB, bool EXPLICIT, bool UMODE = false>
__global__ __launch_bounds__(64) void krylov_grad_kernel(KrylovArgs args) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    krylov_grad_body<NB, EXPLICIT, false, UMODE>(args, smem);
}

// Hermitian-generator variant: half the generator registers, so two waves share a SIMD.
template <int NB, bool EXPLICIT, bool UMODE = false>
__global__ __launch_bounds__(64, NB < 4 ? 2 : 1) void krylov_grad_skew_kernel(KrylovArgs args) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    krylov_grad_body<NB, EXPLICIT, true, UMODE>(args, smem);
}

// ------------------------------------------------------------------------------------------
// Step table: one thread per (seed, step). u_k(t_mid) by the reference's formula (control_at), the
// bound dt (||H0||_1 + sum |u_k| ||G_k||_1) >= ||a||_1 of the step's generator, and from it the
// Pade order (Higham's thresholds) and the squaring count - decisions K1a used to take from the
// norm of the matrix it had just built, at the price of two reductions and a workgroup barrier in
// front of its first product.
__global__ __launch_bounds__(256) void step_table_kernel(StepTableArgs args) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)args.batch * args.nsteps) return;
    const int b = (int)(idx / args.nsteps), step = (int)(idx % args.nsteps);
    const StepInterp si = args.interp[step];
    const double* ctl_b = args.controls + (size_t)b * args.nc * args.K;
    double bound = args.h0_norm;
    for (int k = 0; k < args.K; ++k) {
        const double uk = control_at(ctl_b, si, args.K, k);
        args.ustep[idx * args.K + k] = uk;
        bound = fma(fabs(uk), args.g_norm[k], bound);
    }
    bound *= fabs(args.dt);
    int sq = 0, order = 13;
    bool dominant = false;
    if (!(bound < 1e300)) {  // inf / nan
        atomicOr(args.status, 2);
    } else {
        order = pade_order_for(bound, args.pade_policy);
        // (the host skips the launch of the two-wave K1a when ITS bound is below theta_5; the two
        // bounds differ by rounding at most)
        if (order > args.order_max) order = args.order_max;
        if (order == 13) {
            double th = QOCX_THETA13;
            // (never more squarings than the host sized the sub-step slots for: its bound is this one
            // up to the rounding of the interpolation)
            while (bound > th && sq < args.sq_max) {
                th *= 2.0;
                ++sq;
            }
        }
        dominant = pade_denominator_dominant(order, ldexp(bound, -sq));
    }
    args.s_arr[idx] = step_entry(sq, order) | (dominant ? QOCX_STEP_DOMINANT : 0);
}
void launch_step_table(const StepTableArgs& a, hipStream_t st) {
    const size_t total = (size_t)a.batch * a.nsteps;
    if (total == 0) return;
    hipLaunchKernelGGL(step_table_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a);
}

// ------------------------------------------------------------------------------------------
__global__ void scatter_kernel(ScatterArgs args) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)args.B * args.nc * args.K;
    if (idx >= total) return;
    const int k = (int)(idx % args.K);
    const int ic = (int)((idx / args.K) % args.nc);
    const size_t b = idx / ((size_t)args.K * args.nc);
    double acc = 0;
    if (args.lam_scale != nullptr) {  // unit adjoint: Re(conj(c) gamma)
        const double2 c = args.lam_scale[b * args.S];
        for (int e = args.row_ptr[ic]; e < args.row_ptr[ic + 1]; ++e) {
            const double* g = args.gstep + ((b * args.nsteps + args.col_step[e]) * args.K + k) * 2;
            acc += args.weight[e] * fma(c.y, g[1], c.x * g[0]);
        }
    } else {
        for (int e = args.row_ptr[ic]; e < args.row_ptr[ic + 1]; ++e)
            acc += args.weight[e] * args.gstep[(b * args.nsteps + args.col_step[e]) * args.K + k];
    }
    args.grads[idx] = acc;
}

// Sum over the seeds in a fixed order (deterministic): a block of 256 threads handles 16 gradient
// elements x 16 seed groups; every thread adds up its group's seeds (b = group, group + 16, ...),
// the 16 partial sums of an element are then added in group order. Block 0 also sums the costs.
// 16 KB of results from 4 MB of per-seed gradients.
__global__ __launch_bounds__(256) void reduce_results_kernel(const double* cost, const double* grads,
                                                             int batch, int per_seed, double* out) {
    __shared__ double part[16][17];
    const int jl = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int j = blockIdx.x * 16 + jl;
    double acc = 0;
    if (j < per_seed)
        for (int b = grp; b < batch; b += 16) acc += grads[(size_t)b * per_seed + j];
    part[grp][jl] = acc;
    __syncthreads();
    if (grp == 0 && j < per_seed) {
        double total = 0;
        for (int g = 0; g < 16; ++g) total += part[g][jl];
        out[1 + j] = total;
    }
    if (blockIdx.x == 0) {  // the costs, by the same scheme on 256 threads
        __syncthreads();
        double c = 0;
        for (int b = threadIdx.x; b < batch; b += 256) c += cost[b];
        // 256 partial sums -> 16 -> 1, in index order
        __shared__ double cpart[256];
        cpart[threadIdx.x] = c;
        __syncthreads();
        if (threadIdx.x < 16) {
            double t = 0;
            for (int k = 0; k < 16; ++k) t += cpart[threadIdx.x * 16 + k];
            part[0][threadIdx.x] = t;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0;
            for (int k = 0; k < 16; ++k) t += part[0][k];
            out[0] = t;
        }
    }
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

// Sustained FP64 MFMA rate: every wave issues `iters` rounds of 8 v_mfma_f64_16x16x4_f64 from registers,
// nothing else, all on ONE accumulator - the fastest pattern the pipe has: an MFMA that accumulates
// onto the result of the one issued just before it costs 73 cycles (two waves per SIMD; 68.5 TFLOP/s
// over the chip), three alternating accumulators 83 (61.0), eight 105 (48.2 - the figure of rounds
// 1-3, whose kernel used eight): profiles/r04_mfma_chains.jsonl. Calibrates the roofline peak.
__global__ __launch_bounds__(64) void mfma_peak_kernel(double* out, int iters) {
    const double a = 1.0 + 1e-9 * lane_id(), b = 1.0 - 1e-9 * lane_id();
    d4 c = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) c = mfma_f64(a, b, c);
    }
    if (c[0] + c[1] + c[2] + c[3] == -1.0) out[0] = 1.0;  // keeps the chain alive
}

#ifdef QOCX_DIAG
// Do FP64 MFMA and FP64 / FP32 vector work of two waves on one SIMD overlap? (knob "peak_mode" of
// the diagnostic build.) Every wave runs `iters` rounds of 512 cycles of issue: 8 MFMAs (64 cycles
// each) or 128 vector FMAs (4 cycles each). mode 1: all waves FP64 vector; 2: odd workgroups FP64
// vector, even MFMA; 3: all FP32 vector; 4: odd FP32 vector, even MFMA.
__global__ __launch_bounds__(64) void pipe_mix_kernel(double* out, int iters, int mode) {
    // mixed modes: the role follows the wave slot this wave landed in (HW_ID bits 3:0), so that with
    // two resident waves per SIMD every SIMD holds one wave of each kind
    const unsigned slot = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4);
    const bool vec = (mode == 1 || mode == 3) || ((mode == 2 || mode == 4) && (slot & 1));
    const bool f32 = mode >= 3;
    const double a = 1.0 + 1e-9 * lane_id(), b = 1.0 - 1e-9 * lane_id();
    if (mode >= 11 && mode <= 18) {
        // 8 MFMAs per round on mode - 10 accumulation chains (1: every MFMA accumulates onto the one
        // before it ... 8: the peak kernel's form): what a dependent FP64 MFMA costs
        const int chains = mode - 10;
        d4 c[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) c[k] = d4{0, 0, 0, 0};
        auto rounds = [&](auto tag) {
            constexpr int CH = decltype(tag)::value;
            for (int it = 0; it < iters; ++it)
#pragma unroll
                for (int m = 0; m < 8; ++m) c[m % CH] = mfma_f64(a, b, c[m % CH]);
        };
        if (chains == 1) rounds(std::integral_constant<int, 1>{});
        else if (chains == 2) rounds(std::integral_constant<int, 2>{});
        else if (chains == 3) rounds(std::integral_constant<int, 3>{});
        else if (chains == 4) rounds(std::integral_constant<int, 4>{});
        else if (chains == 6) rounds(std::integral_constant<int, 6>{});
        else rounds(std::integral_constant<int, 8>{});
        d4 s = c[0];
#pragma unroll
        for (int k = 1; k < 8; ++k) s += c[k];
        if (s[0] + s[1] + s[2] + s[3] == -1.0) out[0] = 1.0;
    } else if (!vec) {
        d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
        for (int it = 0; it < iters; ++it) {
            c0 = mfma_f64(a, b, c0); c1 = mfma_f64(a, b, c1); c2 = mfma_f64(a, b, c2); c3 = mfma_f64(a, b, c3);
            c4 = mfma_f64(a, b, c4); c5 = mfma_f64(a, b, c5); c6 = mfma_f64(a, b, c6); c7 = mfma_f64(a, b, c7);
        }
        const d4 s = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
        if (s[0] + s[1] + s[2] + s[3] == -1.0) out[0] = 1.0;
    } else if (!f32) {
        double c[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) c[k] = 1e-3 * k;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int k = 0; k < 16; ++k) c[k] = fma(c[k], a, b);
        }
        double s = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += c[k];
        if (s == -1.0) out[0] = 1.0;
    } else {
        float c[16];
        const float af = (float)a, bf = (float)b;
#pragma unroll
        for (int k = 0; k < 16; ++k) c[k] = 1e-3f * k;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int k = 0; k < 16; ++k) c[k] = fmaf(c[k], af, bf);
        }
        float s = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += c[k];
        if (s == -1.0f) out[0] = 1.0;
    }
}
void launch_pipe_mix(double* out, int blocks, int iters, int mode, hipStream_t st) {
    hipLaunchKernelGGL(pipe_mix_kernel, dim3(blocks), dim3(64), 0, st, out, iters, mode);
}
#endif

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
template <int NB>
static void launch_pq_t(const FactorArgs& a, int nsteps, int batch, hipStream_t st) {
    if (a.hermitian)
        hipLaunchKernelGGL((pade_pq_kernel<NB, true>), dim3(nsteps, batch), dim3(64),
                           PqLds<NB>::BYTES, st, a);
    else
        hipLaunchKernelGGL((pade_pq_kernel<NB, false>), dim3(nsteps, batch), dim3(64),
                           PqLds<NB>::BYTES, st, a);
}
template <int NB>
static void launch_pq_explicit_t(const double2* a_in, int n, const FactorArgs& a, int count,
                                 hipStream_t st) {
    if (a.hermitian)
        hipLaunchKernelGGL((pade_pq_explicit_kernel<NB, true>), dim3(count), dim3(64),
                           PqLds<NB>::BYTES, st, a_in, n, a);
    else
        hipLaunchKernelGGL((pade_pq_explicit_kernel<NB, false>), dim3(count), dim3(64),
                           PqLds<NB>::BYTES, st, a_in, n, a);
}
template <int NB, int W, bool LOADER>
static void launch_sweep_wl(const SweepArgs& a, int batch, hipStream_t st) {
    if constexpr (NB == 4 && !LOADER) {
        if (a.n > 0 && a.n <= 48) {  // nine-tile images: 48 of the 64 columns in LDS, 47 stages per solve
            const int bytes9 = SweepLds<4, 1, 48>::bytes(a.S);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sweep_kernel<4, W, false, false, 48>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, bytes9);
            hipLaunchKernelGGL((sweep_kernel<4, W, false, false, 48>), dim3(batch), dim3(64 * W), bytes9, st, a);
            return;
        }
    }
    const int bytes = LOADER ? SweepLds<NB, 3>::bytes(a.S) : SweepLds<NB>::bytes(a.S);
    if (bytes > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sweep_kernel<NB, W, LOADER>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    hipLaunchKernelGGL((sweep_kernel<NB, W, LOADER>), dim3(batch),
                       dim3(64 * (W + (LOADER ? 1 : 0))), bytes, st, a);
}
template <int NB, int W>
static void launch_sweep_w(const SweepArgs& a, int batch, hipStream_t st) {
    // (the loader variant's ring of three must fit the CU's LDS beside the state vectors)
    if constexpr (SweepPrefetch<NB>::value) {
        if (a.loader && SweepLds<NB, 3>::bytes(a.S) <= 160 * 1024) {
            launch_sweep_wl<NB, W, true>(a, batch, st);
            return;
        }
    }
    launch_sweep_wl<NB, W, false>(a, batch, st);
}
// one state, one wave per seed: one operand set in LDS, two seeds per workgroup
template <int NB>
static void launch_sweep_onebuf(const SweepArgs& a, int batch, hipStream_t st) {
    if constexpr (SweepPrefetch<NB>::value) {
        const int bytes = 2 * SweepLds<NB, 1>::bytes_static(1);
        static bool attr_set = false;
        if (bytes > 48 * 1024 && !attr_set) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sweep_kernel<NB, 1, false, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
            attr_set = true;
        }
        SweepArgs b = a;
        b.batch = batch;
        const int pack = a.onebuf >= 2 ? 2 : 1;  // seeds (waves) per workgroup
        hipLaunchKernelGGL((sweep_kernel<NB, 1, false, true>), dim3((batch + pack - 1) / pack),
                           dim3(64 * pack), bytes / 2 * pack, st, b);
    }
}
template <int NB>
static void launch_sweep_t(const SweepArgs& a, int batch, hipStream_t st) {
    if (SweepPrefetch<NB>::value && a.onebuf && a.S == 1 && !a.loader &&
        diag_getenv("QOCX_SWEEP_W") == nullptr) {
        if (a.one_state && sweep1_supports(NB, a.S)) {  // qocx_sweep1.hip: the same step, bit for bit
            launch_sweep1(NB, a, batch, a.onebuf, st);
            return;
        }
        launch_sweep_onebuf<NB>(a, batch, st);
        return;
    }
    // waves per seed: the states are independent chains, up to four of them run side by side
    // (eight were measured at S = 32: no faster, the adjoint phase is K3 bound by then)
    static const int forced = diag_getenv("QOCX_SWEEP_W") ? atoi(diag_getenv("QOCX_SWEEP_W")) : 0;  // experiments
    const int w = forced > 0 ? forced : (a.S >= 4 ? 4 : (a.S >= 2 ? 2 : 1));
    if (w >= 4) launch_sweep_w<NB, 4>(a, batch, st);
    else if (w >= 2) launch_sweep_w<NB, 2>(a, batch, st);
    else launch_sweep_w<NB, 1>(a, batch, st);
}
template <int NB>
static void launch_krylov_t(const KrylovArgs& a, int nsteps, int batch, hipStream_t st) {
    // a.skew: Hermitian H. The Magnus generators of skew-Hermitian node generators are
    // skew-Hermitian too (up to rounding), so the explicit path may use the same variant.
    if (a.umode) {  // one control set at a time: x = P^-H lambda' is formed in the kernel
        if (a.m_rm != nullptr && a.skew)
            hipLaunchKernelGGL((krylov_grad_skew_kernel<NB, true, true>), dim3(nsteps, batch), dim3(64), KrylovLds<NB>::BYTES, st, a);
        else if (a.m_rm != nullptr)
            hipLaunchKernelGGL((krylov_grad_kernel<NB, true, true>), dim3(nsteps, batch), dim3(64), KrylovLds<NB>::BYTES, st, a);
        else if (a.skew)
            hipLaunchKernelGGL((krylov_grad_skew_kernel<NB, false, true>), dim3(nsteps, batch), dim3(64), KrylovLds<NB>::BYTES, st, a);
        else
            hipLaunchKernelGGL((krylov_grad_kernel<NB, false, true>), dim3(nsteps, batch), dim3(64), KrylovLds<NB>::BYTES, st, a);
        return;
    }
    if (a.m_rm != nullptr && a.skew)
        hipLaunchKernelGGL((krylov_grad_skew_kernel<NB, true>), dim3(nsteps, batch), dim3(64),
                           KrylovLds<NB>::BYTES, st, a);
    else if (a.m_rm != nullptr)
        hipLaunchKernelGGL((krylov_grad_kernel<NB, true>), dim3(nsteps, batch), dim3(64),
                           KrylovLds<NB>::BYTES, st, a);
    else if (a.skew)
        hipLaunchKernelGGL((krylov_grad_skew_kernel<NB, false>), dim3(nsteps, batch), dim3(64),
                           KrylovLds<NB>::BYTES + (a.lds_pad > 0 && a.lds_pad <= 40 * 1024 ? a.lds_pad : 0), st, a);
    else
        hipLaunchKernelGGL((krylov_grad_kernel<NB, false>), dim3(nsteps, batch), dim3(64),
                           KrylovLds<NB>::BYTES, st, a);
}

static bool one_wave_pq() {
    static const bool v = diag_getenv("QOCX_PQ1") != nullptr;
    return v;
}
bool pq_second_pending(int nb, const FactorArgs& a, int nsteps) {
    return nb == 2 && !one_wave_pq() && a.three_wave && pq3_supports(a) && a.four_steps == 2 && pq3_parks(a, nsteps);
}
void launch_pq(int nb, const FactorArgs& a, int nsteps, int batch, hipStream_t st) {
    if (nb == 1 && a.pack8 && a.seg_len == nsteps) {
        if (a.hermitian)
            hipLaunchKernelGGL((pade_pq8_kernel<true>), dim3((nsteps + 1) / 2, batch), dim3(64), PqLds<1>::BYTES, st, a);
        else
            hipLaunchKernelGGL((pade_pq8_kernel<false>), dim3((nsteps + 1) / 2, batch), dim3(64), PqLds<1>::BYTES, st, a);
    } else if (nb == 1) launch_pq_t<1>(a, nsteps, batch, st);
    else if (nb == 4) launch_pq4(a, nsteps, batch, st);
    else if (one_wave_pq()) launch_pq_t<2>(a, nsteps, batch, st);
    else if (a.three_wave && pq3_supports(a)) {
        // steps at order 3 / 5 on the three-wave kernel, the others on the two-wave one (each workgroup
        // looks its step up in the step table); no step of the evaluation is above order 5 when the
        // host's bound says so
        launch_pq3(a, nsteps, batch, st);
        if (a.prefer_low != 2) launch_pq2(a, nsteps, batch, st);
    } else {
        FactorArgs b = a;
        b.three_wave = 0;
        launch_pq2(b, nsteps, batch, st);
    }
}
void launch_pq_explicit(int nb, const double2* a_in, int n, const FactorArgs& a, int count,
                        hipStream_t st) {
    if (nb == 1) launch_pq_explicit_t<1>(a_in, n, a, count, st);
    else if (nb == 4) launch_pq4_explicit(a_in, n, a, count, st);
    else if (one_wave_pq()) launch_pq_explicit_t<2>(a_in, n, a, count, st);
    else launch_pq2_explicit(a_in, n, a, count, st);
}
void launch_umul(int nb, const LuArgs& a, double2* q_img, double2* qt_img, size_t count, hipStream_t st) {
    if (count == 0) return;
    if (nb == 1)
        hipLaunchKernelGGL(umul_kernel<1>, dim3((unsigned)count), dim3(64), PqLds<1>::BYTES, st, a, q_img, qt_img, (unsigned)count);
    else
        hipLaunchKernelGGL(umul_kernel<2>, dim3((unsigned)count), dim3(64), PqLds<2>::BYTES, st, a, q_img, qt_img, (unsigned)count);
}
void launch_lu(int nb, const LuArgs& a, size_t count, hipStream_t st) {
    if (a.inverse && nb == 1 && a.all_dominant && a.pack8) {
        // (count: pairs of steps; FactorArgs::pack8)
        hipLaunchKernelGGL((inv16::inv16_dpp_kernel<1, true>), dim3((unsigned)((count + 3) / 4)), dim3(64), 0, st, a, (unsigned)count);
    } else if (a.inverse && nb == 1 && a.all_dominant)
        hipLaunchKernelGGL((inv16::inv16_dpp_kernel<1, false>), dim3((unsigned)((count + 3) / 4)), dim3(64), 0, st, a, (unsigned)count);
    else if (a.inverse && nb == 1) hipLaunchKernelGGL(inv_kernel<1>, dim3((unsigned)count), dim3(64), 0, st, a);
    else if (a.inverse && nb == 2) hipLaunchKernelGGL(inv_kernel<2>, dim3((unsigned)count), dim3(64), 0, st, a);
    else if (nb == 1) hipLaunchKernelGGL(lu_kernel<1>, dim3((unsigned)count), dim3(64), 0, st, a);
    else if (nb == 2) hipLaunchKernelGGL(lu_kernel<2>, dim3((unsigned)count), dim3(64), 0, st, a);
    else {
        if (a.redo != nullptr) launch_lu4m(a, count, a.redo, st);  // qocx_lu4m.hip: diagonal pivots, MFMA updates
        launch_lu4(a, count, st);                                  // qocx_big.hip: the general elimination
    }
}
void launch_sweep(int nb, const SweepArgs& a, int batch, hipStream_t st) {
    if (nb == 1) launch_sweep_t<1>(a, batch, st);
    else if (nb == 2) launch_sweep_t<2>(a, batch, st);
    else launch_sweep_t<4>(a, batch, st);
}
int sweep_lds_bytes(int nb, int S) {
    return nb == 1 ? SweepLds<1>::bytes(S) : (nb == 2 ? SweepLds<2>::bytes(S) : SweepLds<4>::bytes(S));
}
void launch_krylov(int nb, const KrylovArgs& a, int nsteps, int batch, hipStream_t st) {
    if (nb == 1) launch_krylov_t<1>(a, nsteps, batch, st);
    else if (nb == 2) launch_krylov_t<2>(a, nsteps, batch, st);
    else launch_krylov4(a, nsteps, batch, st);  // qocx_big.hip
}
void launch_scatter(const ScatterArgs& a, hipStream_t st) {
    const size_t total = (size_t)a.B * a.nc * a.K;
    if (total == 0) return;
    hipLaunchKernelGGL(scatter_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a);
}
void launch_reduce_results(const double* cost, const double* grads, int batch, int per_seed,
                           double* out, hipStream_t st) {
    const int blocks = std::max(1, (per_seed + 15) / 16);
    hipLaunchKernelGGL(reduce_results_kernel, dim3(blocks), dim3(256), 0, st, cost, grads, batch,
                       per_seed, out);
}
void launch_mfma_peak(double* out, int blocks, int iters, hipStream_t st) {
    hipLaunchKernelGGL(mfma_peak_kernel, dim3(blocks), dim3(64), 0, st, out, iters);
}
void launch_selftest(double* out, hipStream_t st) {
    hipLaunchKernelGGL(selftest_kernel, dim3(1), dim3(64), 0, st, out);
}

}  // namespace qocx
