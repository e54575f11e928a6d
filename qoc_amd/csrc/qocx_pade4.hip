// qocx_pade4.hip - K1a for 33 <= n <= 64 (padded to 64): a FOUR-wave workgroup per propagator step.
//
// Same construction as the two-wave kernel of qocx_pade2.hip, one size up: wave w of the
// workgroup owns COLUMN BLOCK w (tiles (0..3, w) of every 64 x 64 matrix, C-layout, 64 registers
// per matrix). A product C = A B needs all of A - from the planar LDS slot, staged by the four
// waves together - and only B(:, w), which is the wave's own column block of an earlier product;
// the C-layout tile is the B operand of v_mfma_f64_16x16x4_f64 as it stands. Complex products by
// the 3M scheme (three real MFMA chains, qocx_kernels.hip). A wave holds ~450 registers (a column
// block is 64, and x2, x4, x6, w2 / v and the 3M accumulators are alive together), so a SIMD takes
// one wave and a CU one workgroup: the kernel runs at 79 % of the sustained MFMA rate of its
// 4 608 instructions per matrix.
//
// NT = 3 (n <= 48): the same with three waves and three row tiles per column block - 9 of the 16
// tiles, 0.56 of the MFMA work; the images stay 64 x 64 (K1b, the sweep and K3 see the padded
// matrix), so the waves also write the pad block b0 I of P and Q.
//
// Reference: expm_pade (qoc/standard/functions/expm.py:153-252), always order 13, s from ||a||_1
// and theta13; the LU solve of expm.py:246-249 is K1b + the sweep (qocx_big.hip, qocx_kernels.hip).
#include "qocx_wave.h"

namespace qocx {

namespace pade4 {

constexpr int NP = Geo<4>::NP, MAT = Geo<4>::MAT;  // the images are always 64 x 64

template <int NT>  // row tiles of a column block = column blocks = waves per workgroup
struct Cfg {
    static constexpr int NA = 16 * NT;          // active size
    static constexpr int PITCH = NA + 2;        // LDS row pitch (f64) of the planar A-operand slot
    static constexpr int PLANE = NA * PITCH;
    static constexpr int SLOT_F64 = 2 * PLANE;  // re | im planes
    static constexpr int LDS_BYTES = (SLOT_F64 + 8) * 8;  // + a norm word per wave
};

template <int NT>
struct Col {  // tiles (0..NT-1, w) of a complex matrix, C-layout
    d4 re[NT], im[NT];
};
template <int NT>
struct Acc3 {
    d4 t1[NT], t2[NT], t3[NT];
};

template <int NT>
__device__ __forceinline__ void stage_col(double* slot, int w, const Col<NT>& m) {
    constexpr int PITCH = Cfg<NT>::PITCH, PLANE = Cfg<NT>::PLANE;
    const int q = lane_id() >> 4, c = lane_id() & 15;
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int off = (16 * ti + 4 * r + q) * PITCH + 16 * w + c;
            slot[off] = m.re[ti][r];
            slot[PLANE + off] = m.im[ti][r];
        }
}

// acc(ti) += A(ti, :) B(:, w), 3M scheme; A from the slot, B fragment from `bf`
template <int NT, class BFrag>
__device__ __forceinline__ void gemm3(Acc3<NT>& acc, const double* slot, BFrag bf) {
    constexpr int PITCH = Cfg<NT>::PITCH, PLANE = Cfg<NT>::PLANE;
    const int q = lane_id() >> 4, c = lane_id() & 15;
#pragma unroll
    for (int kk = 0; kk < 4 * NT; ++kk) {
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
        __builtin_amdgcn_sched_barrier(0);  // keep the k-steps apart (register pressure)
    }
}
template <int NT>
__device__ __forceinline__ void acc_zero(Acc3<NT>& a) {
#pragma unroll
    for (int ti = 0; ti < NT; ++ti) {
        a.t1[ti] = d4{0, 0, 0, 0};
        a.t2[ti] = d4{0, 0, 0, 0};
        a.t3[ti] = d4{0, 0, 0, 0};
    }
}
template <int NT>
__device__ __forceinline__ void acc_init(Acc3<NT>& a, const Col<NT>& c) {
#pragma unroll
    for (int ti = 0; ti < NT; ++ti) {
        a.t1[ti] = c.re[ti];
        a.t2[ti] = d4{0, 0, 0, 0};
        a.t3[ti] = c.re[ti] + c.im[ti];
    }
}
template <int NT>
__device__ __forceinline__ void acc_finish(Col<NT>& c, const Acc3<NT>& a) {
#pragma unroll
    for (int ti = 0; ti < NT; ++ti) {
        c.re[ti] = a.t1[ti] - a.t2[ti];
        c.im[ti] = a.t3[ti] - a.t1[ti] - a.t2[ti];
    }
}

struct Out {
    double2* q_img;
    double2* p_img;
    int* s_out;
    int* status;
    int pade_policy;  // FactorArgs::pade_policy
};

// P = v - u ; Q = v + u (expm.py:246), straight from the C-layout registers
template <int NT>
__device__ __forceinline__ void emit_pq(const Out& out, int w, double b0, const Col<NT>& u,
                                        const Col<NT>& v) {
    const int lane = lane_id();
    const int q = lane >> 4, c = lane & 15;
    // ---- P = v - u ; Q = v + u (expm.py:246), straight from the C-layout registers: for a
    // fixed r the four q-lanes of a column hold rows 4r..4r+3 of one tile, i.e. one 64-byte run
    // of the column-major image.
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int idx = (16 * w + c) * NP + 16 * ti + 4 * r + q;
            out.q_img[idx] = make_double2(v.re[ti][r] + u.re[ti][r], v.im[ti][r] + u.im[ti][r]);
            out.p_img[idx] = make_double2(v.re[ti][r] - u.re[ti][r], v.im[ti][r] - u.im[ti][r]);
        }
    if constexpr (NT < 4) {
        // the pad block of the 64 x 64 images: v = b0 I and u = 0 there, so P = Q = b0 I
        constexpr int NA = Cfg<NT>::NA;
#pragma unroll
        for (int r = 0; r < (NP - NA) / 4; ++r) {  // rows NA..63 of this wave's columns
            const int idx = (16 * w + c) * NP + NA + 4 * r + q;
            out.q_img[idx] = make_double2(0, 0);
            out.p_img[idx] = make_double2(0, 0);
        }
        for (int e = w * 64 + lane; e < (NP - NA) * NP; e += 64 * NT) {  // columns NA..63
            const int col = NA + e / NP, row = e % NP;
            const double2 val = make_double2(row == col ? b0 : 0.0, 0.0);
            out.q_img[col * NP + row] = val;
            out.p_img[col * NP + row] = val;
        }
    }
}

// Orders 3, 5, 7, 9 (qocx_wave.h). The generator is staged in the slot (barrier 2 has passed).
template <int NT, class Gen>
__device__ __forceinline__ void low_order_impl(Gen gen, const Out& out, double* smem, int w, int order,
                                               Col<NT>& a) {
    typedef Col<NT> Col;
    typedef Acc3<NT> Acc3;
    double* sl = smem;
    const int lane = lane_id();
    const int q = lane >> 4, c = lane & 15;
    const double b0 = pade_table(order)[0];
    Acc3 acc;
        Col u, v;
        // ---- orders 3, 5, 7, 9 (qocx_wave.h; the shape of the reference's pade3..pade9,
        // expm.py:119-150): x2 = a a, x_{2j} = x2 x_{2j-2}; wp = sum b_{2j+1} x_{2j},
        // v = sum b_{2j} x_{2j} + b0 I, u = wp a + b1 a. No squarings at these norms.
        const double* bt = pade_table(order);
        Col wp, x;
        acc_zero(acc);
        gemm3(acc, sl, [&](int kk, double& bre, double& bim) {
            bre = a.re[kk >> 2][kk & 3];
            bim = a.im[kk >> 2][kk & 3];
        });
        acc_finish(x, acc);
#pragma unroll
        for (int ti = 0; ti < NT; ++ti) {
            wp.re[ti] = bt[3] * x.re[ti];
            wp.im[ti] = bt[3] * x.im[ti];
            v.re[ti] = bt[2] * x.re[ti];
            v.im[ti] = bt[2] * x.im[ti];
            if (ti == w) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (4 * r + q == c) v.re[ti][r] += b0;
            }
        }
        __syncthreads();  // L1: every read of a is done
        if (order >= 5) {
            stage_col(sl, w, x);
            __syncthreads();  // L2
            for (int j = 2; 2 * j < order; ++j) {
                acc_zero(acc);
                gemm3(acc, sl, [&](int kk, double& bre, double& bim) {
                    bre = x.re[kk >> 2][kk & 3];
                    bim = x.im[kk >> 2][kk & 3];
                });
                acc_finish(x, acc);  // (the product is complete: its B operand may go)
                const double bw = bt[2 * j + 1], bv = bt[2 * j];
#pragma unroll
                for (int ti = 0; ti < NT; ++ti) {
                    wp.re[ti] += bw * x.re[ti];
                    wp.im[ti] += bw * x.im[ti];
                    v.re[ti] += bv * x.re[ti];
                    v.im[ti] += bv * x.im[ti];
                }
            }
            __syncthreads();  // L3: every read of x2 is done
        }
        stage_col(sl, w, wp);
        gen(a, w);
#pragma unroll
        for (int ti = 0; ti < NT; ++ti) {
            u.re[ti] = bt[1] * a.re[ti];
            u.im[ti] = bt[1] * a.im[ti];
        }
        acc_init(acc, u);
        __syncthreads();  // L4
        gemm3(acc, sl, [&](int kk, double& bre, double& bim) {
            bre = a.re[kk >> 2][kk & 3];
            bim = a.im[kk >> 2][kk & 3];
        });
        acc_finish(u, acc);
        emit_pq<NT>(out, w, b0, u, v);
}
template <int NT, class Gen>
__device__ __attribute__((noinline)) void low_order_call(Gen gen, const Out& out, double* smem, int w,
                                                         int order) {
    Col<NT> a;
    gen(a, w);
    low_order_impl<NT>(gen, out, smem, w, order, a);
}

// Orders 3 to 9 for HERMITIAN generators (a skew-Hermitian; x2, x4, wp, v Hermitian; u skew-Hermitian;
// Q = P^H). Wave w computes only the row tiles w, w + 1 (, w + 2 for waves 0 and 1 of four) - cyclically -
// of its column block: 6 of the 9 tiles (10 of 16), two per wave (3, 3, 2, 2), instead of three (four)
// per wave; the other tiles of every matrix are their mirrors, written into the A-operand slot next to
// the computed ones, and the B operand of x4 = x2 x2 is read from that slot like the A operand.
template <int NT>
struct Herm {
    static constexpr int NR = NT == 3 ? 2 : 3;  // row tiles of a wave (the last one: waves 0, 1 of four)
};
template <int NT>
struct HCol {  // tile d <-> row tile (w + d) % NT of column block w
    d4 re[Herm<NT>::NR], im[Herm<NT>::NR];
};
template <int NT>
struct HAcc {
    d4 t1[Herm<NT>::NR], t2[Herm<NT>::NR], t3[Herm<NT>::NR];
};
template <int NT>
__device__ __forceinline__ int herm_row(int w, int d) {
    const int r = w + d;
    return r >= NT ? r - NT : r;
}
template <int NT>
__device__ __forceinline__ int herm_count(int w) {
    return NT == 3 ? 2 : (w < 2 ? 3 : 2);
}
// acc(d) += A(row(d), :) B(:, w), d < nr
template <int NT, class BFrag>
__device__ __forceinline__ void gemm3h(HAcc<NT>& acc, const double* slot, int w, int nr, BFrag bf) {
    constexpr int PITCH = Cfg<NT>::PITCH, PLANE = Cfg<NT>::PLANE, NR = Herm<NT>::NR;
    const int q = lane_id() >> 4, c = lane_id() & 15;
    int rowoff[NR];
#pragma unroll
    for (int d = 0; d < NR; ++d) rowoff[d] = (16 * herm_row<NT>(w, d) + c) * PITCH + q;
#pragma unroll
    for (int kk = 0; kk < 4 * NT; ++kk) {
        double are[NR], aim[NR], asum[NR];
#pragma unroll
        for (int d = 0; d < NR; ++d)
            if (d < 2 || nr == 3) {
                are[d] = slot[rowoff[d] + 4 * kk];
                aim[d] = slot[PLANE + rowoff[d] + 4 * kk];
                asum[d] = are[d] + aim[d];
            }
        double bre, bim;
        bf(kk, bre, bim);
        const double bsum = bre + bim;
#pragma unroll
        for (int d = 0; d < NR; ++d)
            if (d < 2 || nr == 3) {
                acc.t1[d] = mfma_f64(are[d], bre, acc.t1[d]);
                acc.t2[d] = mfma_f64(aim[d], bim, acc.t2[d]);
                acc.t3[d] = mfma_f64(asum[d], bsum, acc.t3[d]);
            }
        __builtin_amdgcn_sched_barrier(0);
    }
}
// the wave's tiles and their mirrors (tile (w, row(d)) = sign conj(tile (row(d), w))^T, d >= 1) into the slot
template <int NT>
__device__ __forceinline__ void stage_herm(double* slot, int w, int nr, const HCol<NT>& m, double sign) {
    constexpr int PITCH = Cfg<NT>::PITCH, PLANE = Cfg<NT>::PLANE, NR = Herm<NT>::NR;
    const int q = lane_id() >> 4, c = lane_id() & 15;
#pragma unroll
    for (int d = 0; d < NR; ++d)
        if (d < 2 || nr == 3) {
            const int row = herm_row<NT>(w, d);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int off = (16 * row + 4 * r + q) * PITCH + 16 * w + c;
                slot[off] = m.re[d][r];
                slot[PLANE + off] = m.im[d][r];
                if (d > 0) {
                    const int moff = (16 * w + c) * PITCH + 16 * row + 4 * r + q;
                    slot[moff] = sign * m.re[d][r];
                    slot[PLANE + moff] = -sign * m.im[d][r];
                }
            }
        }
}
// MAXORD: the highest order the build contains - 5 when the host's norm bound is below theta_5 (the
// sixteen-tile build then has no spills), else 9
template <int NT, int MAXORD, class Gen>
__device__ __forceinline__ void low_order_herm(Gen gen, const Out& out, double* smem, int w, int order,
                                               Col<NT>& a) {
    constexpr int PITCH = Cfg<NT>::PITCH, PLANE = Cfg<NT>::PLANE, NR = Herm<NT>::NR;
    typedef HCol<NT> HCol;
    double* sl = smem;
    const int lane = lane_id();
    const int q = lane >> 4, c = lane & 15;
    const int nr = herm_count<NT>(w);
    const double* bt = pade_table(order);
    const double b0 = bt[0];
    HAcc<NT> acc;
    auto zero = [&]() {
#pragma unroll
        for (int d = 0; d < NR; ++d) {
            acc.t1[d] = d4{0, 0, 0, 0};
            acc.t2[d] = d4{0, 0, 0, 0};
            acc.t3[d] = d4{0, 0, 0, 0};
        }
    };
    auto finish = [&](HCol& m) {
#pragma unroll
        for (int d = 0; d < NR; ++d) {
            m.re[d] = acc.t1[d] - acc.t2[d];
            m.im[d] = acc.t3[d] - acc.t1[d] - acc.t2[d];
        }
    };
    // x2 = a a: A from the slot (the generator, staged by body()), B = this wave's column block of it
    HCol x, wp, v, u;
    zero();
    gemm3h<NT>(acc, sl, w, nr, [&](int kk, double& bre, double& bim) {
        bre = a.re[kk >> 2][kk & 3];
        bim = a.im[kk >> 2][kk & 3];
    });
    finish(x);
    auto add_power = [&](const HCol& y, double bw, double bv) {
#pragma unroll
        for (int d = 0; d < NR; ++d) {
            wp.re[d] += bw * y.re[d];
            wp.im[d] += bw * y.im[d];
            v.re[d] += bv * y.re[d];
            v.im[d] += bv * y.im[d];
        }
    };
    auto slot_column = [&](int kk, double& bre, double& bim) {  // B = column block w of the staged matrix
        const int off = (4 * kk + q) * PITCH + 16 * w + c;
        bre = sl[off];
        bim = sl[PLANE + off];
    };
    __syncthreads();  // H1: every read of a is done
    if (order >= 5) {
        stage_herm<NT>(sl, w, nr, x, 1.0);
        __syncthreads();  // H2
        // x4 = x2 x2, both operands from the slot
        zero();
        gemm3h<NT>(acc, sl, w, nr, slot_column);
        HCol x4;
        finish(x4);
        // (wp and v only now: x2 and the accumulators were all that lived through the product)
#pragma unroll
        for (int d = 0; d < NR; ++d) {
            wp.re[d] = bt[3] * x.re[d] + bt[5] * x4.re[d];
            wp.im[d] = bt[3] * x.im[d] + bt[5] * x4.im[d];
            v.re[d] = bt[2] * x.re[d] + bt[4] * x4.re[d];
            v.im[d] = bt[2] * x.im[d] + bt[4] * x4.im[d];
        }
        if (MAXORD >= 7 && order >= 7) {
            // x6 = x4 x2 (the powers commute): this wave's column block of x2 leaves the slot for the
            // registers, x4 takes the slot; x8 = x4 x4 with both operands from the slot again
            Col<NT> c2;
#pragma unroll
            for (int ti = 0; ti < NT; ++ti)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int off = (16 * ti + 4 * r + q) * PITCH + 16 * w + c;
                    c2.re[ti][r] = sl[off];
                    c2.im[ti][r] = sl[PLANE + off];
                }
            __syncthreads();  // every read of x2 is done
            stage_herm<NT>(sl, w, nr, x4, 1.0);
            __syncthreads();
            zero();
            gemm3h<NT>(acc, sl, w, nr, [&](int kk, double& bre, double& bim) {
                bre = c2.re[kk >> 2][kk & 3];
                bim = c2.im[kk >> 2][kk & 3];
            });
            HCol x6;
            finish(x6);
            add_power(x6, bt[7], bt[6]);
            if (order == 9) {
                zero();
                gemm3h<NT>(acc, sl, w, nr, slot_column);
                HCol x8;
                finish(x8);
                add_power(x8, bt[9], bt[8]);
            }
        }
        __syncthreads();  // H3: every read of the slot is done
    } else {
#pragma unroll
        for (int d = 0; d < NR; ++d) {
            wp.re[d] = bt[3] * x.re[d];
            wp.im[d] = bt[3] * x.im[d];
            v.re[d] = bt[2] * x.re[d];
            v.im[d] = bt[2] * x.im[d];
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)  // tile 0 is the diagonal tile (w, w)
        if (4 * r + q == c) v.re[0][r] += b0;
    stage_herm<NT>(sl, w, nr, wp, 1.0);
    // u = wp a + b1 a, B = the generator's column block (rebuilt)
    gen(a, w);
#pragma unroll
    for (int d = 0; d < NR; ++d) {
        const int row = herm_row<NT>(w, d);
        d4 are = a.re[0], aim = a.im[0];
#pragma unroll
        for (int t = 1; t < NT; ++t)
            if (row == t) {
                are = a.re[t];
                aim = a.im[t];
            }
        acc.t1[d] = bt[1] * are;
        acc.t2[d] = d4{0, 0, 0, 0};
        acc.t3[d] = bt[1] * (are + aim);
    }
    __syncthreads();  // H4
    gemm3h<NT>(acc, sl, w, nr, [&](int kk, double& bre, double& bim) {
        bre = a.re[kk >> 2][kk & 3];
        bim = a.im[kk >> 2][kk & 3];
    });
    finish(u);
    // ---- P = v - u ; Q = v + u (expm.py:246): the wave's tiles, and their mirrors Q = P^H
#pragma unroll
    for (int d = 0; d < NR; ++d)
        if (d < 2 || nr == 3) {
            const int row = herm_row<NT>(w, d);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int idx = (16 * w + c) * NP + 16 * row + 4 * r + q;
                const double qr = v.re[d][r] + u.re[d][r], qi = v.im[d][r] + u.im[d][r];
                const double pr = v.re[d][r] - u.re[d][r], pi = v.im[d][r] - u.im[d][r];
                out.q_img[idx] = make_double2(qr, qi);
                out.p_img[idx] = make_double2(pr, pi);
                if (d > 0) {  // element (16 w + c, 16 row + 4 r + q)
                    const int midx = (16 * row + 4 * r + q) * NP + 16 * w + c;
                    out.q_img[midx] = make_double2(pr, -pi);
                    out.p_img[midx] = make_double2(qr, -qi);
                }
            }
        }
    if constexpr (NT < 4) {
        // the pad block of the 64 x 64 images: v = b0 I and u = 0 there, so P = Q = b0 I
        constexpr int NA = Cfg<NT>::NA;
#pragma unroll
        for (int r = 0; r < (NP - NA) / 4; ++r) {  // rows NA..63 of this wave's columns
            const int idx = (16 * w + c) * NP + NA + 4 * r + q;
            out.q_img[idx] = make_double2(0, 0);
            out.p_img[idx] = make_double2(0, 0);
        }
        for (int e = w * 64 + lane; e < (NP - NA) * NP; e += 64 * NT) {  // columns NA..63
            const int col = NA + e / NP, row = e % NP;
            const double2 val = make_double2(row == col ? b0 : 0.0, 0.0);
            out.q_img[col * NP + row] = val;
            out.p_img[col * NP + row] = val;
        }
    }
}

// Order 13 (expm.py:153-159). The generator is staged in the slot (barrier 2 has passed); `a` is
// this wave's column block of it, scaled by 2^-sq.
template <int NT, class Gen>
__device__ __forceinline__ void high_order_impl(Gen gen, const Out& out, double* smem, int w, int sq,
                                                Col<NT>& a) {
    typedef Col<NT> Col;
    typedef Acc3<NT> Acc3;
    double* sl = smem;
    const int lane = lane_id();
    const int q = lane >> 4, c = lane & 15;
    const double b0 = PADE_B[0];
    const double scale = ldexp(1.0, -sq);
    Acc3 acc;
    // ---- a2 = a a ; a4 = a2 a2 ; a6 = a2 a4 (expm.py:154-156) ----------------------------
    Col u, v;
    Col x2, x4, x6;
    acc_zero(acc);
    gemm3(acc, sl, [&](int kk, double& bre, double& bim) {
        bre = a.re[kk >> 2][kk & 3];
        bim = a.im[kk >> 2][kk & 3];
    });
    acc_finish(x2, acc);
    __syncthreads();  // 3: every read of a is done
    stage_col(sl, w, x2);
    __syncthreads();  // 4
    acc_zero(acc);
    gemm3(acc, sl, [&](int kk, double& bre, double& bim) {
        bre = x2.re[kk >> 2][kk & 3];
        bim = x2.im[kk >> 2][kk & 3];
    });
    acc_finish(x4, acc);
    acc_zero(acc);
    gemm3(acc, sl, [&](int kk, double& bre, double& bim) {
        bre = x4.re[kk >> 2][kk & 3];
        bim = x4.im[kk >> 2][kk & 3];
    });
    acc_finish(x6, acc);
    __syncthreads();  // 5: every read of a2 is done
    stage_col(sl, w, x6);
    __syncthreads();  // 6

    // ---- w2 = a6 (b13 a6 + b11 a4 + b9 a2) + b7 a6 + b5 a4 + b3 a2 (expm.py:157) ---------
    // ---- v  = a6 (b12 a6 + b10 a4 + b8 a2) + b6 a6 + b4 a4 + b2 a2 + b0 I (expm.py:158) --
    const double b1 = PADE_B[1], b2 = PADE_B[2], b3 = PADE_B[3], b4 = PADE_B[4],
                 b5 = PADE_B[5], b6 = PADE_B[6], b7 = PADE_B[7], b8 = PADE_B[8], b9 = PADE_B[9],
                 b10 = PADE_B[10], b11 = PADE_B[11], b12 = PADE_B[12], b13 = PADE_B[13];
    Col w2;
#pragma unroll
    for (int ti = 0; ti < NT; ++ti) {
        w2.re[ti] = b7 * x6.re[ti] + b5 * x4.re[ti] + b3 * x2.re[ti];
        w2.im[ti] = b7 * x6.im[ti] + b5 * x4.im[ti] + b3 * x2.im[ti];
    }
    acc_init(acc, w2);
    gemm3(acc, sl, [&](int kk, double& bre, double& bim) {
        const int tb = kk >> 2, r = kk & 3;
        bre = b13 * x6.re[tb][r] + b11 * x4.re[tb][r] + b9 * x2.re[tb][r];
        bim = b13 * x6.im[tb][r] + b11 * x4.im[tb][r] + b9 * x2.im[tb][r];
    });
    acc_finish(w2, acc);
#pragma unroll
    for (int ti = 0; ti < NT; ++ti) {
        v.re[ti] = b6 * x6.re[ti] + b4 * x4.re[ti] + b2 * x2.re[ti];
        v.im[ti] = b6 * x6.im[ti] + b4 * x4.im[ti] + b2 * x2.im[ti];
        if (ti == w) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (4 * r + q == c) v.re[ti][r] += b0;
        }
    }
    acc_init(acc, v);
    gemm3(acc, sl, [&](int kk, double& bre, double& bim) {
        const int tb = kk >> 2, r = kk & 3;
        bre = b12 * x6.re[tb][r] + b10 * x4.re[tb][r] + b8 * x2.re[tb][r];
        bim = b12 * x6.im[tb][r] + b10 * x4.im[tb][r] + b8 * x2.im[tb][r];
    });
    acc_finish(v, acc);
    __syncthreads();  // 7: every read of a6 is done
    stage_col(sl, w, w2);

    // ---- u = a w2 + b1 a (expm.py:157), evaluated as w2 a + b1 a: w2 is a polynomial in a, the
    // two commute, and this way the A operand is the product just finished while B is the
    // wave's own column block of the generator (rebuilt rather than kept in 64 registers).
    gen(a, w);
    if (sq > 0) {
#pragma unroll
        for (int ti = 0; ti < NT; ++ti) {
            a.re[ti] *= scale;
            a.im[ti] *= scale;
        }
    }
#pragma unroll
    for (int ti = 0; ti < NT; ++ti) {
        u.re[ti] = b1 * a.re[ti];
        u.im[ti] = b1 * a.im[ti];
    }
    acc_init(acc, u);
    __syncthreads();  // 8
    gemm3(acc, sl, [&](int kk, double& bre, double& bim) {
        bre = a.re[kk >> 2][kk & 3];
        bim = a.im[kk >> 2][kk & 3];
    });
    acc_finish(u, acc);

    emit_pq<NT>(out, w, b0, u, v);
}
template <int NT, class Gen>
__device__ __attribute__((noinline)) void high_order_call(Gen gen, const Out& out, double* smem, int w,
                                                          int sq) {
    Col<NT> a;
    gen(a, w);
    if (sq > 0) {
        const double scale = ldexp(1.0, -sq);
#pragma unroll
        for (int ti = 0; ti < NT; ++ti) {
            a.re[ti] *= scale;
            a.im[ti] *= scale;
        }
    }
    high_order_impl<NT>(gen, out, smem, w, sq, a);
}

// Every wave executes the same barriers; w = the wave's column block (wave-uniform).
template <int NT, bool LOWINL, int HERM = 0, class Gen>  // HERM: 0, or the highest order of the Hermitian-tile build
__device__ __forceinline__ void body(Gen gen, const Out& out, double* smem, int w) {
    typedef Col<NT> Col;
    double* sl = smem;
    double* nrm = sl + Cfg<NT>::SLOT_F64;
    const int lane = lane_id();

    // ---- generator, 1-norm, scaling (expm.py:116, :238-241) -----------------------------
    Col a;
    gen(a, w);
    {
        double e = 0;
#pragma unroll
        for (int ti = 0; ti < NT; ++ti)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                e += sqrt(a.re[ti][r] * a.re[ti][r] + a.im[ti][r] * a.im[ti][r]);
        e += __shfl_xor(e, 16);
        e += __shfl_xor(e, 32);
        e = wave_max(e);  // the largest column sum of this column block
        if (lane == 0) nrm[w] = e;
    }
    __syncthreads();  // 1
    double norm1 = fmax(fmax(nrm[0], nrm[1]), nrm[2]);
    if constexpr (NT == 4) norm1 = fmax(norm1, nrm[3]);
    int sq = 0;
    int order = pade_order_for(norm1, out.pade_policy);  // every wave reads the same numbers
    {
        double th = QOCX_THETA13;
        while (norm1 > th && sq < 30) {
            th *= 2.0;
            ++sq;
        }
        if (!(norm1 <= th)) {  // inf / nan / absurd
            if (w == 0 && lane == 0) atomicOr(out.status, 2);
            sq = 0;
            order = 13;
        }
    }
    const double scale = ldexp(1.0, -sq);
    if (sq > 0) {
#pragma unroll
        for (int ti = 0; ti < NT; ++ti) {
            a.re[ti] *= scale;
            a.im[ti] *= scale;
        }
    }
    if (w == 0 && lane == 0) *out.s_out = step_entry(sq, order);
    stage_col(sl, w, a);
    __syncthreads();  // 2

    // The two paths are not allocated together: the sixteen-tile kernel has no registers to spare
    // (both inlined, the [13/13] path spilled 94 registers instead of 22 and ran 11 % slower). One
    // of them is a call; LOWINL - the host's choice from its bound of the norms, FactorArgs::
    // prefer_low - says which one is inlined. Same arithmetic either way.
    if constexpr (HERM != 0) {
        // Hermitian generators (FactorArgs::hermitian), the host's norm bound below theta_9 (theta_5):
        // orders 3 to 9 (5) on two thirds of the tiles
        // (an order beyond the bound here means a generator that is not finite - status bit 2 is set,
        // what is computed does not matter)
        low_order_herm<NT, HERM>(gen, out, smem, w, order > HERM ? HERM : order, a);
    } else if (order != 13) {
        if constexpr (LOWINL) low_order_impl<NT>(gen, out, smem, w, order, a);
        else low_order_call<NT>(gen, out, smem, w, order);
    } else {
        if constexpr (LOWINL) high_order_call<NT>(gen, out, smem, w, sq);
        else high_order_impl<NT>(gen, out, smem, w, sq, a);
    }
}

template <int NT, bool LOWINL, int HERM = 0>
__global__ __launch_bounds__(64 * NT, LOWINL ? 2 : 1) void pade_pq4_kernel(FactorArgs args) {
    typedef Col<NT> Col;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* smem = reinterpret_cast<double*>(smem_raw);
    const int step = args.step0 + blockIdx.x, b = blockIdx.y;
    const int lane = lane_id();
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t m = (size_t)b * args.nsteps + step;
    Out out;
    out.q_img = args.q_img + m * MAT;
    out.p_img = args.lu_img + m * MAT;
    out.s_out = args.s_arr + m;
    out.status = args.status;
    out.pade_policy = args.pade_policy;
    const StepInterp si = args.interp[step];
    const double* ctl_b = args.controls + (size_t)b * args.nc * args.K;
    const size_t tsel = (args.nt == 1) ? 0 : (size_t)step;
    const double2* h0 = args.h0_cimg + tsel * MAT;
    const double2* g = args.g_cimg + tsel * args.K * MAT;
    const double dt = args.dt;
    const int K = args.K;
    auto gen = [&](Col& a, int wcol) {
        // H = h0 + sum_k u_k g_k ; a = dt * (-i H)  (schroedingerdiscrete.py:485-486,
        // mathmethods.py:90-93); C-layout image index ((ti * 4 + tj) * 4 + r) * 64 + lane (the
        // C-images always have four tiles per side)
        d4 hre[NT], him[NT];
#pragma unroll
        for (int ti = 0; ti < NT; ++ti)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double2 e = h0[((ti * 4 + wcol) * 4 + r) * 64 + lane];
                hre[ti][r] = e.x;
                him[ti][r] = e.y;
            }
        for (int k = 0; k < K; ++k) {
            const double uk = control_at(ctl_b, si, K, k);
            const double2* gk = g + (size_t)k * MAT;
#pragma unroll
            for (int ti = 0; ti < NT; ++ti)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double2 e = gk[((ti * 4 + wcol) * 4 + r) * 64 + lane];
                    hre[ti][r] += uk * e.x;
                    him[ti][r] += uk * e.y;
                }
        }
#pragma unroll
        for (int ti = 0; ti < NT; ++ti) {
            a.re[ti] = dt * him[ti];
            a.im[ti] = -dt * hre[ti];
        }
    };
    body<NT, LOWINL, HERM>(gen, out, smem, w);
}

// Explicit-generator variant: a[count][n][n] row-major complex in HBM (opaque Hamiltonians, debug)
template <int NT, bool LOWINL, int HERM = 0>
__global__ __launch_bounds__(64 * NT, ((NT == 3 || HERM != 0) && LOWINL) ? 2 : 1) void pade_pq4_explicit_kernel(
    const double2* a_in, int n, FactorArgs args) {
    typedef Col<NT> Col;
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
    out.status = args.status;
    out.pade_policy = args.pade_policy;
    const double2* am = a_in + m * (size_t)n * n;
    auto gen = [&](Col& a, int wcol) {
#pragma unroll
        for (int ti = 0; ti < NT; ++ti)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * ti + 4 * r + q, col = 16 * wcol + c;
                double2 e = make_double2(0, 0);
                if (row < n && col < n) e = am[(size_t)row * n + col];
                a.re[ti][r] = e.x;
                a.im[ti][r] = e.y;
            }
    };
    body<NT, LOWINL, HERM>(gen, out, smem, w);
}

}  // namespace pade4

template <int NT, bool LOWINL, int HERM = 0>
static void launch_pq4_t(const FactorArgs& a, int nsteps, int batch, hipStream_t st) {
    constexpr int bytes = pade4::Cfg<NT>::LDS_BYTES;
    if (bytes > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pade4::pade_pq4_kernel<NT, LOWINL, HERM>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    hipLaunchKernelGGL((pade4::pade_pq4_kernel<NT, LOWINL, HERM>), dim3(nsteps, batch), dim3(64 * NT), bytes,
                       st, a);
}
template <int NT, bool LOWINL, int HERM = 0>
static void launch_pq4_explicit_t(const double2* a_in, int n, const FactorArgs& a, int count,
                                  hipStream_t st) {
    constexpr int bytes = pade4::Cfg<NT>::LDS_BYTES;
    if (bytes > 48 * 1024)
        (void)hipFuncSetAttribute(
            reinterpret_cast<const void*>(pade4::pade_pq4_explicit_kernel<NT, LOWINL, HERM>),
            hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    hipLaunchKernelGGL((pade4::pade_pq4_explicit_kernel<NT, LOWINL, HERM>), dim3(count), dim3(64 * NT), bytes,
                       st, a_in, n, a);
}

// a.n: the Hilbert size (33..64). Up to 48 the three-wave form computes 9 of the 16 tiles.
// a.prefer_low: the host's bound of the generator norms is below theta_9, so every step takes a
// low Pade order - the variant with THAT path inlined runs (same results either way).
void launch_pq4(const FactorArgs& a, int nsteps, int batch, hipStream_t st) {
    const bool low = a.prefer_low != 0 && a.pade_policy != 13;
    const bool herm = low && a.hermitian != 0 && a.herm_tiles != 0;
    const bool herm5 = herm && a.prefer_low == 2;  // the norm bound is below theta_5
    if (a.n > 0 && a.n <= 48) {
        if (herm) launch_pq4_t<3, true, 9>(a, nsteps, batch, st);
        else if (low) launch_pq4_t<3, true>(a, nsteps, batch, st);
        else launch_pq4_t<3, false>(a, nsteps, batch, st);
    } else {
        if (herm5) launch_pq4_t<4, true, 5>(a, nsteps, batch, st);
        else if (herm) launch_pq4_t<4, true, 9>(a, nsteps, batch, st);
        else if (low) launch_pq4_t<4, true>(a, nsteps, batch, st);
        else launch_pq4_t<4, false>(a, nsteps, batch, st);
    }
}
void launch_pq4_explicit(const double2* a_in, int n, const FactorArgs& a, int count, hipStream_t st) {
    const bool low = a.prefer_low != 0 && a.pade_policy != 13;
    // (Hermitian tiles: generators given as matrices - Magnus M4 / M6, opaque Hamiltonians - that are
    // skew-Hermitian, FactorArgs::hermitian)
    const bool herm = low && a.hermitian != 0 && a.herm_tiles != 0;
    if (a.n > 0 && a.n <= 48) {
        if (herm) launch_pq4_explicit_t<3, true, 9>(a_in, n, a, count, st);
        else if (low) launch_pq4_explicit_t<3, true>(a_in, n, a, count, st);
        else launch_pq4_explicit_t<3, false>(a_in, n, a, count, st);
    } else {
        if (herm) launch_pq4_explicit_t<4, true, 9>(a_in, n, a, count, st);
        else if (low) launch_pq4_explicit_t<4, true>(a_in, n, a, count, st);
        else launch_pq4_explicit_t<4, false>(a_in, n, a, count, st);
    }
}

}  // namespace qocx
