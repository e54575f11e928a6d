// qocx_tilewave.h - multi-wave workgroups that keep whole matrices in LDS and own them tile-wise
// (the Magnus kernels of qocx_magnus4w.hip, the four-tile Lindblad kernel of qocx_lindblad4t.hip):
// geometries, the wave's tiles of a matrix, LDS matrices of pitch NP + 1 (fragments along rows and
// along columns are both conflict free), products with both operands read from LDS - plain or
// conjugate-transposed -, the commutator and its reverse rule.
#ifndef QOCX_TILEWAVE_H
#define QOCX_TILEWAVE_H

#include "qocx_wave.h"

namespace qocx {
namespace tilewave {

// Geometry. G32 (17 <= n <= 32): four waves, wave w owns tile (w & 1, w >> 1) of every 32 x 32 matrix.
// G48 (33 <= n <= 48, round 4): three waves, wave w owns COLUMN BLOCK w - tiles (0..2, w) - of the
// active 48 x 48 part of a matrix whose HBM images are 64 x 64 (the pad block of a generator is zero);
// the one-wave kernels of qocx_magnus.hip keep sixteen-tile matrices in scratch memory there (M6 at
// n = 48: 2.8 s per 256-seed evaluation).
struct G32 {
    static constexpr int NP = 32, IMG = 32, WAVES = 4, NTW = 1;
    static __device__ __forceinline__ int ti(int i, int w) { return w & 1; }
    static __device__ __forceinline__ int tj(int w) { return w >> 1; }
};
struct G48 {
    static constexpr int NP = 48, IMG = 64, WAVES = 3, NTW = 3;
    static __device__ __forceinline__ int ti(int i, int w) { return i; }
    static __device__ __forceinline__ int tj(int w) { return w; }
};
// G64 (49 <= n <= 64, round 4): four waves, wave w the column block w (four tiles) of a 64 x 64 matrix.
// Four such matrices are 266 KiB: only TWO live in LDS at a time - the operands of the product at
// hand - and everything else waits as tiles in registers (the `_r` forms of the commutator rules
// below take their operands from registers and put them into the two slots themselves).
struct G64 {
    static constexpr int NP = 64, IMG = 64, WAVES = 4, NTW = 4;
    static __device__ __forceinline__ int ti(int i, int w) { return i; }
    static __device__ __forceinline__ int tj(int w) { return w; }
};
template <class G>
struct Dim {
    static constexpr int NP = G::NP, PM = NP + 1, MELEM = NP * PM, MBYTES = MELEM * 16, KS = NP / 4;
    static constexpr int IMG = G::IMG, IMAT = IMG * IMG, TPS = IMG / 16;  // HBM images: pitch, elements, tiles per side
    static constexpr int SLOTS = NP > 48 ? 2 : 4;             // LDS-resident matrices
    // the partial sums [node][control][wave] of the control gradients (3 x 64 x 4 doubles) take the place of
    // matrix 0 once the last product has read its operands (a barrier before the first of them is written):
    // 66 KB instead of 72 at n <= 32, and the reverse kernels fit twice on a CU like the forward ones
    static constexpr int RED_OFF = 0;
    static constexpr int LDS_BYTES = SLOTS * MBYTES;
    static constexpr int LDS_BYTES_FWD = SLOTS * MBYTES;
};
template <class G>
struct Tile {  // the wave's tiles of a matrix, C-layout
    d4 re[G::NTW], im[G::NTW];
};
template <class G>
__device__ __forceinline__ Tile<G> tile_zero() {
    Tile<G> t;
#pragma unroll
    for (int i = 0; i < G::NTW; ++i) {
        t.re[i] = d4{0, 0, 0, 0};
        t.im[i] = d4{0, 0, 0, 0};
    }
    return t;
}
template <class G>
__device__ __forceinline__ void tile_axpy(Tile<G>& y, double a, const Tile<G>& x) {
#pragma unroll
    for (int i = 0; i < G::NTW; ++i) {
        y.re[i] += a * x.re[i];
        y.im[i] += a * x.im[i];
    }
}
template <class G>
__device__ __forceinline__ void tile_scale(Tile<G>& y, double a) {
#pragma unroll
    for (int i = 0; i < G::NTW; ++i) {
        y.re[i] *= a;
        y.im[i] *= a;
    }
}

template <class G>
struct Wave {
    typedef Dim<G> D;
    typedef Tile<G> T;
    static constexpr int PM = D::PM, NTW = G::NTW, KS = D::KS;
    int q, c, lane, w, tid;
    bool skew;
    double2* lds;
    __device__ __forceinline__ int ti(int i) const { return G::ti(i, w); }
    __device__ __forceinline__ int tj() const { return G::tj(w); }
    __device__ __forceinline__ double2* mat(int which) const { return lds + (size_t)which * D::MELEM; }
    // element (row 16 ti + 4 r + q, col 16 tj + c) of an LDS matrix <-> component r of a tile
    __device__ __forceinline__ void store(const T& t, int which) const {
        double2* m = mat(which);
#pragma unroll
        for (int i = 0; i < NTW; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                m[(16 * ti(i) + 4 * r + q) * PM + 16 * tj() + c] = make_double2(t.re[i][r], t.im[i][r]);
    }
    __device__ __forceinline__ T load(int which) const {
        const double2* m = mat(which);
        T t;
#pragma unroll
        for (int i = 0; i < NTW; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double2 e = m[(16 * ti(i) + 4 * r + q) * PM + 16 * tj() + c];
                t.re[i][r] = e.x;
                t.im[i][r] = e.y;
            }
        return t;
    }
    // acc += sign * op(A) op(B), this wave's tiles; op = plain or conjugate transpose
    template <bool ADJ_A, bool ADJ_B>
    __device__ __forceinline__ void mm(T& acc, int a_which, int b_which, double sign) const {
        const double2* am = mat(a_which);
        const double2* bm = mat(b_which);
        double2 b[KS];
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            if (ADJ_B) {
                const double2 v = bm[(16 * tj() + c) * PM + 4 * kk + q];  // conj(B[c][k])
                b[kk] = make_double2(v.x, -v.y);
            } else {
                b[kk] = bm[(4 * kk + q) * PM + 16 * tj() + c];
            }
        }
#pragma unroll
        for (int i = 0; i < NTW; ++i) {
            double2 a[KS];
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
                if (ADJ_A) {
                    const double2 v = am[(4 * kk + q) * PM + 16 * ti(i) + c];  // conj(A[k][r])
                    a[kk] = make_double2(sign * v.x, -sign * v.y);
                } else {
                    const double2 v = am[(16 * ti(i) + c) * PM + 4 * kk + q];
                    a[kk] = make_double2(sign * v.x, sign * v.y);
                }
            }
            d4 t1 = {0, 0, 0, 0}, t2 = {0, 0, 0, 0}, t3 = {0, 0, 0, 0};
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
                t1 = mfma_f64(a[kk].x, b[kk].x, t1);
                t2 = mfma_f64(a[kk].y, b[kk].y, t2);
                t3 = mfma_f64(a[kk].x + a[kk].y, b[kk].x + b[kk].y, t3);
            }
            acc.re[i] += t1 - t2;
            acc.im[i] += t3 - t1 - t2;
        }
    }
    // this wave's tiles of M^H, M an LDS matrix
    __device__ __forceinline__ T load_adjoint(int which) const {
        const double2* m = mat(which);
        T t;
#pragma unroll
        for (int i = 0; i < NTW; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double2 e = m[(16 * tj() + c) * PM + 16 * ti(i) + 4 * r + q];
                t.re[i][r] = e.x;
                t.im[i][r] = -e.y;
            }
        return t;
    }
    // Z = X Y - Y X   (convenience.py:16-29). skew (every node generator skew-Hermitian, i.e. Hermitian
    // H0(t), G_k(t): then so are b1, b2, b3, every commutator and M itself): Y X = (X Y)^H, ONE product;
    // the tiles of (X Y)^H come through the free LDS matrix `scratch`. Workgroup barriers inside.
    __device__ __forceinline__ T commutator(int x, int y, int scratch) const {
        T z = tile_zero<G>();
        mm<false, false>(z, x, y, 1.0);
        if (skew) {
            store(z, scratch);
            __syncthreads();
            tile_axpy<G>(z, -1.0, load_adjoint(scratch));
            __syncthreads();  // `scratch` may be written again
            return z;
        }
        mm<false, false>(z, y, x, -1.0);
        return z;
    }
    // Cotangents of Z = X Y - Y X with Zbar in LDS matrix `zb` (complete on entry). General:
    // Xbar = Zbar Y^H - Y^H Zbar, Ybar = X^H Zbar - Zbar X^H. skew: only the skew-Hermitian part of a
    // cotangent reaches the controls (the contraction is with -i G_k) and the forward maps send skew
    // perturbations to skew ones, so with Zs = (Zbar - Zbar^H) / 2: Xbar = V - V^H, V = Y Zs;
    // Ybar = W - W^H, W = Zs X - two products instead of four. `sa`, `sb`: LDS matrices that are free
    // once the products have read their operands (they may be x's and y's own). Barriers inside.
    // ZB_SKEW: the caller knows Zbar to be skew-Hermitian already (a cotangent that came out of this rule,
    // V - V^H, or one it has projected itself): Zs = Zbar, two barriers and an exchange less.
    template <bool ZB_SKEW = false>
    __device__ __forceinline__ void commutator_vjp(T& xbar, T& ybar, int x, int y, int zb, int sa,
                                                   int sb) const {
        if (skew) {
            if (!ZB_SKEW) {
                T zs = load(zb);
                tile_axpy<G>(zs, -1.0, load_adjoint(zb));
                tile_scale<G>(zs, 0.5);
                __syncthreads();  // Zbar has been read
                store(zs, zb);
                __syncthreads();
            }
            xbar = tile_zero<G>();
            mm<false, false>(xbar, y, zb, 1.0);  // V = Y Zs
            ybar = tile_zero<G>();
            mm<false, false>(ybar, zb, x, 1.0);  // W = Zs X
            __syncthreads();  // the operands have been read
            store(xbar, sa);
            store(ybar, sb);
            __syncthreads();
            tile_axpy<G>(xbar, -1.0, load_adjoint(sa));
            tile_axpy<G>(ybar, -1.0, load_adjoint(sb));
            __syncthreads();
            return;
        }
        xbar = tile_zero<G>();
        mm<false, true>(xbar, zb, y, 1.0);   //  Zbar Y^H
        mm<true, false>(xbar, y, zb, -1.0);  // -Y^H Zbar
        ybar = tile_zero<G>();
        mm<true, false>(ybar, x, zb, 1.0);   //  X^H Zbar
        mm<false, true>(ybar, zb, x, -1.0);  // -Zbar X^H
    }
    // ---- the same two rules with the operands in REGISTERS and two LDS slots (0, 1) to work in:
    // for geometries whose matrices do not fit LDS four at a time (G64). Both slots are free on
    // entry and on return; barriers inside.
    __device__ __forceinline__ T commutator_r(const T& x, const T& y) const {
        store(x, 0);
        store(y, 1);
        __syncthreads();
        T z = tile_zero<G>();
        mm<false, false>(z, 0, 1, 1.0);
        if (!skew) mm<false, false>(z, 1, 0, -1.0);
        __syncthreads();  // the operands have been read
        if (skew) {       // Y X = (X Y)^H
            store(z, 0);
            __syncthreads();
            tile_axpy<G>(z, -1.0, load_adjoint(0));
            __syncthreads();
        }
        return z;
    }
    __device__ __forceinline__ void commutator_vjp_r(T& xbar, T& ybar, const T& x, const T& y,
                                                     const T& zbar) const {
        if (skew) {
            store(zbar, 0);
            __syncthreads();
            T zs = zbar;
            tile_axpy<G>(zs, -1.0, load_adjoint(0));
            tile_scale<G>(zs, 0.5);
            __syncthreads();  // Zbar has been read
            store(zs, 0);
            store(y, 1);
            __syncthreads();
            T v = tile_zero<G>();
            mm<false, false>(v, 1, 0, 1.0);  // V = Y Zs
            __syncthreads();
            store(x, 1);
            __syncthreads();
            T wq = tile_zero<G>();
            mm<false, false>(wq, 0, 1, 1.0);  // W = Zs X
            __syncthreads();
            store(v, 0);
            store(wq, 1);
            __syncthreads();
            xbar = v;
            tile_axpy<G>(xbar, -1.0, load_adjoint(0));
            ybar = wq;
            tile_axpy<G>(ybar, -1.0, load_adjoint(1));
            __syncthreads();
            return;
        }
        store(zbar, 0);
        store(y, 1);
        __syncthreads();
        xbar = tile_zero<G>();
        mm<false, true>(xbar, 0, 1, 1.0);   //  Zbar Y^H
        mm<true, false>(xbar, 1, 0, -1.0);  // -Y^H Zbar
        __syncthreads();
        store(x, 1);
        __syncthreads();
        ybar = tile_zero<G>();
        mm<true, false>(ybar, 1, 0, 1.0);   //  X^H Zbar
        mm<false, true>(ybar, 0, 1, -1.0);  // -Zbar X^H
        __syncthreads();
    }
    // C-image index of component r of tile i: ((ti * TPS + tj) * 4 + r) * 64 + lane
    __device__ __forceinline__ int cimg(int i, int r) const {
        return ((ti(i) * D::TPS + tj()) * 4 + r) * 64 + lane;
    }
};


template <class G>
__device__ __forceinline__ Wave<G> make_wave(char* smem, bool skew) {
    Wave<G> wv;
    wv.skew = skew;
    wv.tid = threadIdx.x;
    wv.lane = wv.tid & 63;
    wv.w = __builtin_amdgcn_readfirstlane(wv.tid >> 6);
    wv.q = wv.lane >> 4;
    wv.c = wv.lane & 15;
    wv.lds = reinterpret_cast<double2*>(smem);
    return wv;
}

}  // namespace tilewave
}  // namespace qocx

#endif
