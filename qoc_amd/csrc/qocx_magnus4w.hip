// qocx_magnus4w.hip - Magnus M4 / M6 generators and their reverse rules as multi-wave workgroups
// with every matrix resident in LDS: 17 <= n <= 32 on four waves, 33 <= n <= 48 on three.
//
// The one-wave kernels of qocx_magnus.hip hold whole 32 x 32 complex matrices in registers (32
// per matrix and lane): 400-512 registers, one wave per SIMD, the M6 forms spill, and the named
// intermediates travel through HBM scratch - 125 000 cycles per M6 forward step for 18 000 cycles
// of matrix-core work. Here a workgroup owns one propagator step; the node generators, b1, b2, b3,
// w, x, y and the cotangent being propagated are row-major LDS images of pitch NP + 1 (fragments
// along rows and along columns are both conflict free), a wave owns one tile (n <= 32) or one
// column block (n <= 48) of every result, a product is 24 (108) MFMAs per wave (3M scheme) with both
// operands read from LDS - plain or conjugate-transposed - and the cotangent accumulators b1bar,
// b2bar, b3bar, c12bar are tiles in registers. Same formulas as qocx_magnus.hip (reference:
// magnus_m4 / magnus_m6, qoc/core/mathmethods.py:96-164; reverse rule of Z = XY - YX: Xbar = Zbar
// Y^H - Y^H Zbar, Ybar = X^H Zbar - Zbar X^H), with the one-product commutators for skew-Hermitian
// node generators.
#include "qocx_wave.h"

namespace qocx {

namespace magnus4w {

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
    static constexpr int RED_OFF = SLOTS * MBYTES;
    static constexpr int LDS_BYTES = RED_OFF + 3 * 64 * 4 * 8;  // + partial sums [node][control][wave]
    static constexpr int LDS_BYTES_FWD = SLOTS * MBYTES;
};
constexpr double M4_F0 = 0.14433756729740643;  // sqrt(3)/12
constexpr double M6_F0 = 1.2909944487358056;   // sqrt(15)/3
constexpr double M6_F1 = 10.0 / 3.0;
constexpr double M6_F2 = 0.5;
constexpr double M6_F3 = 1.0 / 240.0;
constexpr double M6_F4 = 1.0 / 60.0;
// LDS matrices
// (b3 is never an operand of a product: it stays in registers. The forward kernel needs four
// matrices at a time - y takes the place of b2 - and so fits twice on a CU at n <= 32.)
enum { L_B1 = 0, L_B2, L_W, L_X, L_COUNT };

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
    __device__ __forceinline__ void commutator_vjp(T& xbar, T& ybar, int x, int y, int zb, int sa,
                                                   int sb) const {
        if (skew) {
            T zs = load(zb);
            tile_axpy<G>(zs, -1.0, load_adjoint(zb));
            tile_scale<G>(zs, 0.5);
            __syncthreads();  // Zbar has been read
            store(zs, zb);
            __syncthreads();
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

// this wave's tiles of a_q = -i (H0 + sum_k u_k(t_q) G_k) at quadrature node `node`
template <class G>
__device__ __forceinline__ Tile<G> node_generator(const Wave<G>& wv, const MagnusArgs& args, int step, int node,
                                                  const double* ctl_b) {
    typedef Dim<G> D;
    const size_t col = (size_t)step * args.nodes + node;
    const size_t tsel = (args.nt == 1) ? 0 : col;
    const double2* h0 = args.h0_cimg + tsel * D::IMAT;
    const double2* g = args.g_cimg + tsel * args.K * D::IMAT;
    const StepInterp si = args.interp[col];
    Tile<G> h;
#pragma unroll
    for (int i = 0; i < G::NTW; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double2 e = h0[wv.cimg(i, r)];
            h.re[i][r] = e.x;
            h.im[i][r] = e.y;
        }
    for (int k = 0; k < args.K; ++k) {
        const double uk = control_at(ctl_b, si, args.K, k);
        const double2* gk = g + (size_t)k * D::IMAT;
#pragma unroll
        for (int i = 0; i < G::NTW; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double2 e = gk[wv.cimg(i, r)];
                h.re[i][r] += uk * e.x;
                h.im[i][r] += uk * e.y;
            }
    }
    Tile<G> a;
#pragma unroll
    for (int i = 0; i < G::NTW; ++i) {
        a.re[i] = h.im[i];
        a.im[i] = -h.re[i];
    }
    return a;
}

// g_k = Re <abar, -i G_k> of one node: this wave's partial sums -> red[(node * 64 + k) * 4 + w]
template <class G>
__device__ __forceinline__ void contract_node(const Wave<G>& wv, const Tile<G>& abar, const MagnusArgs& args,
                                              int step, int node, double* red) {
    typedef Dim<G> D;
    const size_t col = (size_t)step * args.nodes + node;
    const size_t tsel = (args.nt == 1) ? 0 : col;
    const double2* g = args.g_cimg + tsel * args.K * D::IMAT;
    for (int k = 0; k < args.K; ++k) {
        double acc = 0;
#pragma unroll
        for (int i = 0; i < G::NTW; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double2 e = g[(size_t)k * D::IMAT + wv.cimg(i, r)];
                acc += abar.re[i][r] * e.y - abar.im[i][r] * e.x;
            }
        acc = wave_sum(acc);
        if (wv.lane == 0) red[(node * 64 + k) * 4 + wv.w] = acc;
    }
}

template <class G>
__device__ __forceinline__ Wave<G> make_wave(char* smem, const MagnusArgs& args) {
    Wave<G> wv;
    wv.skew = args.skew != 0;
    wv.tid = threadIdx.x;
    wv.lane = wv.tid & 63;
    wv.w = __builtin_amdgcn_readfirstlane(wv.tid >> 6);
    wv.q = wv.lane >> 4;
    wv.c = wv.lane & 15;
    wv.lds = reinterpret_cast<double2*>(smem);
    return wv;
}

// b1, b2 -> LDS (M6), b3 in registers; every wave writes its tiles
template <class G>
__device__ __forceinline__ void m6_nodes(const Wave<G>& wv, const MagnusArgs& args, int step,
                                         const double* ctl_b, Tile<G>& b1, Tile<G>& b2, Tile<G>& b3) {
    const double dt = args.dt;
    const Tile<G> a1 = node_generator<G>(wv, args, step, 0, ctl_b);
    const Tile<G> a2 = node_generator<G>(wv, args, step, 1, ctl_b);
    const Tile<G> a3 = node_generator<G>(wv, args, step, 2, ctl_b);
    b1 = tile_zero<G>();
    b2 = tile_zero<G>();
    b3 = tile_zero<G>();
    // b1 = dt a2 ; b2 = F0 dt (a3 - a1) ; b3 = F1 dt (a3 - 2 a2 + a1)   (mathmethods.py:153-155)
    tile_axpy<G>(b1, dt, a2);
    tile_axpy<G>(b2, -M6_F0 * dt, a1);
    tile_axpy<G>(b2, M6_F0 * dt, a3);
    tile_axpy<G>(b3, M6_F1 * dt, a1);
    tile_axpy<G>(b3, -2.0 * M6_F1 * dt, a2);
    tile_axpy<G>(b3, M6_F1 * dt, a3);
    wv.store(b1, L_B1);
    wv.store(b2, L_B2);
}

// the M6 intermediates w, x, y -> LDS; returns this wave's tiles of m. (y replaces b2)
template <class G>
__device__ __forceinline__ Tile<G> m6_forward(const Wave<G>& wv, const MagnusArgs& args, int step,
                                              const double* ctl_b) {
    constexpr int Y_SLOT = L_B2;
    constexpr int W_SLOT = L_W, X_SLOT = L_X;
    Tile<G> b1, b2, b3;
    m6_nodes<G>(wv, args, step, ctl_b, b1, b2, b3);
    __syncthreads();
    // c12 = [b1, b2] ; w = 2 b3 + c12 ; x = -20 b1 - b3 + c12
    const Tile<G> c12 = wv.commutator(L_B1, L_B2, W_SLOT);
    Tile<G> wt = c12, xt = c12;
    tile_axpy<G>(wt, 2.0, b3);
    tile_axpy<G>(xt, -1.0, b3);
    tile_axpy<G>(xt, -20.0, b1);
    wv.store(wt, W_SLOT);
    wv.store(xt, X_SLOT);
    __syncthreads();
    // y = b2 - F4 [b1, w]  (the b2 matrix is free: scratch)
    Tile<G> yt = wv.commutator(L_B1, W_SLOT, L_B2);
    tile_scale<G>(yt, -M6_F4);
    tile_axpy<G>(yt, 1.0, b2);
    wv.store(yt, Y_SLOT);  // (the b2 matrix was last read before the barrier above)
    __syncthreads();
    // m = b1 + F2 b3 + F3 [x, y]
    Tile<G> m = wv.commutator(X_SLOT, Y_SLOT, L_B1);  // (b1 is in registers: its matrix is scratch)
    tile_scale<G>(m, M6_F3);
    tile_axpy<G>(m, 1.0, b1);
    tile_axpy<G>(m, M6_F2, b3);
    return m;
}

template <class G, int NODES>
__global__ __launch_bounds__(64 * G::WAVES) void magnus4w_fwd_kernel(MagnusArgs args) {
    typedef Dim<G> D;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const Wave<G> wv = make_wave<G>(smem, args);
    const int step = args.step0 + blockIdx.x;
    const size_t b = blockIdx.y;
    const size_t m = b * args.nsteps + step;
    const double* ctl_b = args.controls + b * args.nc * args.K;
    Tile<G> mt;
    if (NODES == 2) {
        // m4 = dt/2 (a1 + a2) + F0 dt^2 [a2, a1]   (mathmethods.py:119-121)
        const Tile<G> a1 = node_generator<G>(wv, args, step, 0, ctl_b);
        const Tile<G> a2 = node_generator<G>(wv, args, step, 1, ctl_b);
        wv.store(a1, L_B1);
        wv.store(a2, L_B2);
        __syncthreads();
        mt = wv.commutator(L_B2, L_B1, L_W);
        tile_scale<G>(mt, M4_F0 * args.dt * args.dt);
        tile_axpy<G>(mt, 0.5 * args.dt, a1);
        tile_axpy<G>(mt, 0.5 * args.dt, a2);
    } else {
        mt = m6_forward<G>(wv, args, step, ctl_b);
    }
    double2* out = args.m_rm + m * D::IMAT;  // row-major, image pitch
#pragma unroll
    for (int i = 0; i < G::NTW; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            out[(size_t)(16 * wv.ti(i) + 4 * r + wv.q) * D::IMG + 16 * wv.tj() + wv.c] =
                make_double2(mt.re[i][r], mt.im[i][r]);
    if constexpr (D::NP < D::IMG) {  // the pad block of the 64 x 64 image: zero
        for (int e = wv.tid; e < D::IMAT; e += 64 * G::WAVES) {
            const int row = e / D::IMG, col = e % D::IMG;
            if (row >= D::NP || col >= D::NP) out[e] = make_double2(0.0, 0.0);
        }
    }
}

template <class G, int NODES>
__global__ __launch_bounds__(64 * G::WAVES) void magnus4w_vjp_kernel(MagnusArgs args) {
    typedef Dim<G> D;
    typedef Tile<G> T;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const Wave<G> wv = make_wave<G>(smem, args);
    double* red = reinterpret_cast<double*>(smem + D::RED_OFF);
    const int step = args.step0 + blockIdx.x;
    const size_t b = blockIdx.y;
    const size_t m = b * args.nsteps + step;
    const double* ctl_b = args.controls + b * args.nc * args.K;
    const double dt = args.dt;
    T mbar;
    {
        const double2* in = args.mbar_rm + m * D::IMAT;
#pragma unroll
        for (int i = 0; i < G::NTW; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double2 e = in[(size_t)(16 * wv.ti(i) + 4 * r + wv.q) * D::IMG + 16 * wv.tj() + wv.c];
                mbar.re[i][r] = e.x;
                mbar.im[i][r] = e.y;
            }
    }
    if (NODES == 2) {
        // a1bar = dt/2 mbar + d[a2, a1]/d a1 ; a2bar likewise (cbar = F0 dt^2 mbar)
        wv.store(node_generator<G>(wv, args, step, 0, ctl_b), L_B1);
        wv.store(node_generator<G>(wv, args, step, 1, ctl_b), L_B2);
        T cbar = mbar;
        tile_scale<G>(cbar, M4_F0 * dt * dt);
        wv.store(cbar, L_W);
        __syncthreads();
        T a2bar, a1bar;
        wv.commutator_vjp(a2bar, a1bar, L_B2, L_B1, L_W, L_X, L_B1);
        tile_axpy<G>(a1bar, 0.5 * dt, mbar);
        tile_axpy<G>(a2bar, 0.5 * dt, mbar);
        contract_node<G>(wv, a1bar, args, step, 0, red);
        contract_node<G>(wv, a2bar, args, step, 1, red);
    } else {
        // Four LDS matrices s0..s3 hold the operands of the product group at hand; everything else
        // waits as tiles in registers and is put back when it becomes an operand.
        constexpr int S0 = 0, S1 = 1, S2 = 2, S3 = 3;
        T b1, b2, b3;
        m6_nodes<G>(wv, args, step, ctl_b, b1, b2, b3);  // b1 -> s0, b2 -> s1
        __syncthreads();
        // c12 = [b1, b2] ; w = 2 b3 + c12 ; x = -20 b1 - b3 + c12 ; y = b2 - F4 [b1, w]
        const T c12 = wv.commutator(S0, S1, S2);
        T wt = c12, xt = c12;
        tile_axpy<G>(wt, 2.0, b3);
        tile_axpy<G>(xt, -1.0, b3);
        tile_axpy<G>(xt, -20.0, b1);
        wv.store(wt, S2);
        wv.store(xt, S3);
        __syncthreads();
        T yt = wv.commutator(S0, S2, S1);  // (the b2 matrix is scratch: b2 is in registers)
        tile_scale<G>(yt, -M6_F4);
        tile_axpy<G>(yt, 1.0, b2);
        // (xbar, ybar) = vjp of [x, y] with F3 mbar: x in s3, y -> s1, Zbar -> s0
        T zb = mbar;
        tile_scale<G>(zb, M6_F3);
        __syncthreads();  // b1 (s0) and b2 (s1) have been read
        wv.store(yt, S1);
        wv.store(zb, S0);
        __syncthreads();
        T xbar, ybar;
        wv.commutator_vjp(xbar, ybar, S3, S1, S0, S3, S1);
        // b1bar = mbar - 20 xbar ; b3bar = F2 mbar - xbar ; c12bar = xbar ; b2bar = ybar
        T b1bar = mbar, b3bar = tile_zero<G>(), c12bar = xbar, b2bar = ybar;
        tile_axpy<G>(b1bar, -20.0, xbar);
        tile_axpy<G>(b3bar, M6_F2, mbar);
        tile_axpy<G>(b3bar, -1.0, xbar);
        // y = b2 - F4 [b1, w]: innerbar = -F4 ybar ; (d1, wbar) = vjp of [b1, w]: b1 -> s1, w in s2
        T inner = ybar;
        tile_scale<G>(inner, -M6_F4);
        __syncthreads();
        wv.store(b1, S1);
        wv.store(inner, S0);
        __syncthreads();
        T d1, wbar;
        wv.commutator_vjp(d1, wbar, S1, S2, S0, S3, S2);
        tile_axpy<G>(b1bar, 1.0, d1);
        tile_axpy<G>(b3bar, 2.0, wbar);
        tile_axpy<G>(c12bar, 1.0, wbar);
        // c12 = [b1, b2]: b1 in s1, b2 -> s2
        __syncthreads();
        wv.store(b2, S2);
        wv.store(c12bar, S0);
        __syncthreads();
        T d2;
        wv.commutator_vjp(d1, d2, S1, S2, S0, S3, S2);
        tile_axpy<G>(b1bar, 1.0, d1);
        tile_axpy<G>(b2bar, 1.0, d2);
        // a1bar = -F0 dt b2bar + F1 dt b3bar ; a2bar = dt b1bar - 2 F1 dt b3bar ;
        // a3bar = F0 dt b2bar + F1 dt b3bar
        T abar = tile_zero<G>();
        tile_axpy<G>(abar, -M6_F0 * dt, b2bar);
        tile_axpy<G>(abar, M6_F1 * dt, b3bar);
        contract_node<G>(wv, abar, args, step, 0, red);
        abar = tile_zero<G>();
        tile_axpy<G>(abar, dt, b1bar);
        tile_axpy<G>(abar, -2.0 * M6_F1 * dt, b3bar);
        contract_node<G>(wv, abar, args, step, 1, red);
        abar = tile_zero<G>();
        tile_axpy<G>(abar, M6_F0 * dt, b2bar);
        tile_axpy<G>(abar, M6_F1 * dt, b3bar);
        contract_node<G>(wv, abar, args, step, 2, red);
    }
    __syncthreads();
    for (int e = wv.tid; e < NODES * args.K; e += 64 * G::WAVES) {
        const int node = e / args.K, k = e % args.K;
        const double* p = red + (node * 64 + k) * 4;
        double sum = 0;
#pragma unroll
        for (int w = 0; w < G::WAVES; ++w) sum += p[w];
        if (G::WAVES == 4) sum = (p[0] + p[1]) + (p[2] + p[3]);  // (the order of round 2: bit-identical results)
        args.gstep[((b * (size_t)args.nsteps + step) * NODES + node) * args.K + k] = sum;
    }
}

// ---- two LDS slots, operands from registers (G64) ---------------------------------------------------
template <class G>
__device__ __forceinline__ void m6_nodes_r(const Wave<G>& wv, const MagnusArgs& args, int step,
                                           const double* ctl_b, Tile<G>& b1, Tile<G>& b2, Tile<G>& b3) {
    const double dt = args.dt;
    b1 = tile_zero<G>();
    b2 = tile_zero<G>();
    b3 = tile_zero<G>();
    {   // b1 = dt a2 ; b2 = F0 dt (a3 - a1) ; b3 = F1 dt (a3 - 2 a2 + a1)   (mathmethods.py:153-155)
        const Tile<G> a1 = node_generator<G>(wv, args, step, 0, ctl_b);
        tile_axpy<G>(b2, -M6_F0 * dt, a1);
        tile_axpy<G>(b3, M6_F1 * dt, a1);
    }
    {
        const Tile<G> a2 = node_generator<G>(wv, args, step, 1, ctl_b);
        tile_axpy<G>(b1, dt, a2);
        tile_axpy<G>(b3, -2.0 * M6_F1 * dt, a2);
    }
    {
        const Tile<G> a3 = node_generator<G>(wv, args, step, 2, ctl_b);
        tile_axpy<G>(b2, M6_F0 * dt, a3);
        tile_axpy<G>(b3, M6_F1 * dt, a3);
    }
}

template <class G, int NODES>
__global__ __launch_bounds__(64 * G::WAVES) void magnus2s_fwd_kernel(MagnusArgs args) {
    typedef Dim<G> D;
    typedef Tile<G> T;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const Wave<G> wv = make_wave<G>(smem, args);
    const int step = args.step0 + blockIdx.x;
    const size_t b = blockIdx.y;
    const size_t m = b * args.nsteps + step;
    const double* ctl_b = args.controls + b * args.nc * args.K;
    T mt;
    if (NODES == 2) {
        // m4 = dt/2 (a1 + a2) + F0 dt^2 [a2, a1]   (mathmethods.py:119-121)
        const T a1 = node_generator<G>(wv, args, step, 0, ctl_b);
        const T a2 = node_generator<G>(wv, args, step, 1, ctl_b);
        mt = wv.commutator_r(a2, a1);
        tile_scale<G>(mt, M4_F0 * args.dt * args.dt);
        tile_axpy<G>(mt, 0.5 * args.dt, a1);
        tile_axpy<G>(mt, 0.5 * args.dt, a2);
    } else {
        T b1, b2, b3;
        m6_nodes_r<G>(wv, args, step, ctl_b, b1, b2, b3);
        // c12 = [b1, b2] ; w = 2 b3 + c12 ; x = -20 b1 - b3 + c12 ; y = b2 - F4 [b1, w] ;
        // m = b1 + F2 b3 + F3 [x, y]   (mathmethods.py:156-163)
        const T c12 = wv.commutator_r(b1, b2);
        T wt = c12, xt = c12;
        tile_axpy<G>(wt, 2.0, b3);
        tile_axpy<G>(xt, -1.0, b3);
        tile_axpy<G>(xt, -20.0, b1);
        T yt = wv.commutator_r(b1, wt);
        tile_scale<G>(yt, -M6_F4);
        tile_axpy<G>(yt, 1.0, b2);
        mt = wv.commutator_r(xt, yt);
        tile_scale<G>(mt, M6_F3);
        tile_axpy<G>(mt, 1.0, b1);
        tile_axpy<G>(mt, M6_F2, b3);
    }
    double2* out = args.m_rm + m * D::IMAT;  // row-major
#pragma unroll
    for (int i = 0; i < G::NTW; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            out[(size_t)(16 * wv.ti(i) + 4 * r + wv.q) * D::IMG + 16 * wv.tj() + wv.c] =
                make_double2(mt.re[i][r], mt.im[i][r]);
}

template <class G, int NODES>
__global__ __launch_bounds__(64 * G::WAVES) void magnus2s_vjp_kernel(MagnusArgs args) {
    typedef Dim<G> D;
    typedef Tile<G> T;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const Wave<G> wv = make_wave<G>(smem, args);
    double* red = reinterpret_cast<double*>(smem + D::RED_OFF);
    const int step = args.step0 + blockIdx.x;
    const size_t b = blockIdx.y;
    const size_t m = b * args.nsteps + step;
    const double* ctl_b = args.controls + b * args.nc * args.K;
    const double dt = args.dt;
    T mbar;
    {
        const double2* in = args.mbar_rm + m * D::IMAT;
#pragma unroll
        for (int i = 0; i < G::NTW; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double2 e = in[(size_t)(16 * wv.ti(i) + 4 * r + wv.q) * D::IMG + 16 * wv.tj() + wv.c];
                mbar.re[i][r] = e.x;
                mbar.im[i][r] = e.y;
            }
    }
    if (NODES == 2) {
        const T a1 = node_generator<G>(wv, args, step, 0, ctl_b);
        const T a2 = node_generator<G>(wv, args, step, 1, ctl_b);
        T cbar = mbar;
        tile_scale<G>(cbar, M4_F0 * dt * dt);
        T a2bar, a1bar;
        wv.commutator_vjp_r(a2bar, a1bar, a2, a1, cbar);
        tile_axpy<G>(a1bar, 0.5 * dt, mbar);
        tile_axpy<G>(a2bar, 0.5 * dt, mbar);
        contract_node<G>(wv, a1bar, args, step, 0, red);
        contract_node<G>(wv, a2bar, args, step, 1, red);
    } else {
        T b1, b2, b3;
        m6_nodes_r<G>(wv, args, step, ctl_b, b1, b2, b3);
        const T c12 = wv.commutator_r(b1, b2);
        T wt = c12, xt = c12;
        tile_axpy<G>(wt, 2.0, b3);
        tile_axpy<G>(xt, -1.0, b3);
        tile_axpy<G>(xt, -20.0, b1);
        T yt = wv.commutator_r(b1, wt);
        tile_scale<G>(yt, -M6_F4);
        tile_axpy<G>(yt, 1.0, b2);
        // (xbar, ybar) = vjp of [x, y] with F3 mbar
        T zb = mbar;
        tile_scale<G>(zb, M6_F3);
        T xbar, ybar;
        wv.commutator_vjp_r(xbar, ybar, xt, yt, zb);
        // b1bar = mbar - 20 xbar ; b3bar = F2 mbar - xbar ; c12bar = xbar ; b2bar = ybar
        T b1bar = mbar, b3bar = tile_zero<G>(), c12bar = xbar, b2bar = ybar;
        tile_axpy<G>(b1bar, -20.0, xbar);
        tile_axpy<G>(b3bar, M6_F2, mbar);
        tile_axpy<G>(b3bar, -1.0, xbar);
        // y = b2 - F4 [b1, w]: innerbar = -F4 ybar ; (d1, wbar) = vjp of [b1, w]
        T inner = ybar;
        tile_scale<G>(inner, -M6_F4);
        T d1, wbar;
        wv.commutator_vjp_r(d1, wbar, b1, wt, inner);
        tile_axpy<G>(b1bar, 1.0, d1);
        tile_axpy<G>(b3bar, 2.0, wbar);
        tile_axpy<G>(c12bar, 1.0, wbar);
        // c12 = [b1, b2]
        T d2;
        wv.commutator_vjp_r(d1, d2, b1, b2, c12bar);
        tile_axpy<G>(b1bar, 1.0, d1);
        tile_axpy<G>(b2bar, 1.0, d2);
        // a1bar = -F0 dt b2bar + F1 dt b3bar ; a2bar = dt b1bar - 2 F1 dt b3bar ;
        // a3bar = F0 dt b2bar + F1 dt b3bar
        T abar = tile_zero<G>();
        tile_axpy<G>(abar, -M6_F0 * dt, b2bar);
        tile_axpy<G>(abar, M6_F1 * dt, b3bar);
        contract_node<G>(wv, abar, args, step, 0, red);
        abar = tile_zero<G>();
        tile_axpy<G>(abar, dt, b1bar);
        tile_axpy<G>(abar, -2.0 * M6_F1 * dt, b3bar);
        contract_node<G>(wv, abar, args, step, 1, red);
        abar = tile_zero<G>();
        tile_axpy<G>(abar, M6_F0 * dt, b2bar);
        tile_axpy<G>(abar, M6_F1 * dt, b3bar);
        contract_node<G>(wv, abar, args, step, 2, red);
    }
    __syncthreads();
    for (int e = wv.tid; e < NODES * args.K; e += 64 * G::WAVES) {
        const int node = e / args.K, k = e % args.K;
        const double* p = red + (node * 64 + k) * 4;
        args.gstep[((b * (size_t)args.nsteps + step) * NODES + node) * args.K + k] =
            (p[0] + p[1]) + (p[2] + p[3]);
    }
}

}  // namespace magnus4w

// n: the Hilbert size (17..32: four waves, a tile each; 33..48: three waves, a column block each;
// 49..64: four waves, a column block each, two matrices in LDS at a time)
bool magnus4w_supports(int nb, int K, int n) { return (nb == 2 || (nb == 4 && n > 32 && n <= 64)) && K <= 64; }

template <class Kern>
static void magnus4w_attr(Kern k, int bytes) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

static void magnus4w_prepare() {
    static bool attr_set = false;
    if (attr_set) return;
    using namespace magnus4w;
    magnus4w_attr(magnus4w_fwd_kernel<G32, 2>, Dim<G32>::LDS_BYTES);
    magnus4w_attr(magnus4w_fwd_kernel<G32, 3>, Dim<G32>::LDS_BYTES);
    magnus4w_attr(magnus4w_vjp_kernel<G32, 2>, Dim<G32>::LDS_BYTES);
    magnus4w_attr(magnus4w_vjp_kernel<G32, 3>, Dim<G32>::LDS_BYTES);
    magnus4w_attr(magnus4w_fwd_kernel<G48, 2>, Dim<G48>::LDS_BYTES);
    magnus4w_attr(magnus4w_fwd_kernel<G48, 3>, Dim<G48>::LDS_BYTES);
    magnus4w_attr(magnus4w_vjp_kernel<G48, 2>, Dim<G48>::LDS_BYTES);
    magnus4w_attr(magnus4w_vjp_kernel<G48, 3>, Dim<G48>::LDS_BYTES);
    magnus4w_attr(magnus2s_fwd_kernel<G64, 2>, Dim<G64>::LDS_BYTES);
    magnus4w_attr(magnus2s_fwd_kernel<G64, 3>, Dim<G64>::LDS_BYTES);
    magnus4w_attr(magnus2s_vjp_kernel<G64, 2>, Dim<G64>::LDS_BYTES);
    magnus4w_attr(magnus2s_vjp_kernel<G64, 3>, Dim<G64>::LDS_BYTES);
    attr_set = true;
}

void launch_magnus4w_fwd(const MagnusArgs& a, int batch, hipStream_t st) {
    using namespace magnus4w;
    magnus4w_prepare();
    const dim3 grid(a.seg_len, batch);
    if (a.n > 48) {
        if (a.nodes == 2)
            hipLaunchKernelGGL((magnus2s_fwd_kernel<G64, 2>), grid, dim3(256), Dim<G64>::LDS_BYTES_FWD, st, a);
        else
            hipLaunchKernelGGL((magnus2s_fwd_kernel<G64, 3>), grid, dim3(256), Dim<G64>::LDS_BYTES_FWD, st, a);
        return;
    }
    if (a.n > 32) {
        if (a.nodes == 2)
            hipLaunchKernelGGL((magnus4w_fwd_kernel<G48, 2>), grid, dim3(192), Dim<G48>::LDS_BYTES_FWD, st, a);
        else
            hipLaunchKernelGGL((magnus4w_fwd_kernel<G48, 3>), grid, dim3(192), Dim<G48>::LDS_BYTES_FWD, st, a);
        return;
    }
    if (a.nodes == 2)
        hipLaunchKernelGGL((magnus4w_fwd_kernel<G32, 2>), grid, dim3(256), Dim<G32>::LDS_BYTES_FWD, st, a);
    else
        hipLaunchKernelGGL((magnus4w_fwd_kernel<G32, 3>), grid, dim3(256), Dim<G32>::LDS_BYTES_FWD, st, a);
}

void launch_magnus4w_vjp(const MagnusArgs& a, int batch, hipStream_t st) {
    using namespace magnus4w;
    magnus4w_prepare();
    const dim3 grid(a.seg_len, batch);
    if (a.n > 48) {
        if (a.nodes == 2)
            hipLaunchKernelGGL((magnus2s_vjp_kernel<G64, 2>), grid, dim3(256), Dim<G64>::LDS_BYTES, st, a);
        else
            hipLaunchKernelGGL((magnus2s_vjp_kernel<G64, 3>), grid, dim3(256), Dim<G64>::LDS_BYTES, st, a);
        return;
    }
    if (a.n > 32) {
        if (a.nodes == 2)
            hipLaunchKernelGGL((magnus4w_vjp_kernel<G48, 2>), grid, dim3(192), Dim<G48>::LDS_BYTES, st, a);
        else
            hipLaunchKernelGGL((magnus4w_vjp_kernel<G48, 3>), grid, dim3(192), Dim<G48>::LDS_BYTES, st, a);
        return;
    }
    if (a.nodes == 2)
        hipLaunchKernelGGL((magnus4w_vjp_kernel<G32, 2>), grid, dim3(256), Dim<G32>::LDS_BYTES, st, a);
    else
        hipLaunchKernelGGL((magnus4w_vjp_kernel<G32, 3>), grid, dim3(256), Dim<G32>::LDS_BYTES, st, a);
}

}  // namespace qocx
