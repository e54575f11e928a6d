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
#include <cstdio>
#include <cstdlib>
#include "qocx_tilewave.h"

namespace qocx {

namespace magnus4w {

using namespace tilewave;

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
    return tilewave::make_wave<G>(smem, args.skew != 0);
}

// Constant H0, G_k (one table, args.nt == 1): b1, b2, b3 are linear in (H0, G_k) with the node controls as
// coefficients - H0 drops out of the differences b2, b3 - so every image is read ONCE per step instead of
// once per node: b1 = -i dt (H0 + sum u_k(t2) G_k), b2 = -i F0 dt sum (u_k(t3) - u_k(t1)) G_k,
// b3 = -i F1 dt sum (u_k(t3) - 2 u_k(t2) + u_k(t1)) G_k.
template <class G>
__device__ __forceinline__ void m6_nodes_const(const Wave<G>& wv, const MagnusArgs& args, int step,
                                               const double* ctl_b, Tile<G>& b1, Tile<G>& b2, Tile<G>& b3) {
    typedef Dim<G> D;
    const double dt = args.dt;
    const size_t col = (size_t)step * args.nodes;
    const StepInterp s0 = args.interp[col], s1 = args.interp[col + 1], s2 = args.interp[col + 2];
    Tile<G> h1, h2 = tile_zero<G>(), h3 = tile_zero<G>();
#pragma unroll
    for (int i = 0; i < G::NTW; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double2 e = args.h0_cimg[wv.cimg(i, r)];
            h1.re[i][r] = e.x;
            h1.im[i][r] = e.y;
        }
    for (int k = 0; k < args.K; ++k) {
        const double u0 = control_at(ctl_b, s0, args.K, k), u1 = control_at(ctl_b, s1, args.K, k),
                     u2 = control_at(ctl_b, s2, args.K, k);
        const double d2 = u2 - u0, d3 = (u2 - 2.0 * u1) + u0;
        const double2* gk = args.g_cimg + (size_t)k * D::IMAT;
#pragma unroll
        for (int i = 0; i < G::NTW; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double2 e = gk[wv.cimg(i, r)];
                h1.re[i][r] += u1 * e.x;
                h1.im[i][r] += u1 * e.y;
                h2.re[i][r] += d2 * e.x;
                h2.im[i][r] += d2 * e.y;
                h3.re[i][r] += d3 * e.x;
                h3.im[i][r] += d3 * e.y;
            }
    }
    // a = -i h: (re, im) = (h.im, -h.re)
#pragma unroll
    for (int i = 0; i < G::NTW; ++i) {
        b1.re[i] = dt * h1.im[i];
        b1.im[i] = -dt * h1.re[i];
        b2.re[i] = (M6_F0 * dt) * h2.im[i];
        b2.im[i] = -(M6_F0 * dt) * h2.re[i];
        b3.re[i] = (M6_F1 * dt) * h3.im[i];
        b3.im[i] = -(M6_F1 * dt) * h3.re[i];
    }
}

// the control gradients of the three nodes from b1bar, b2bar, b3bar, constant G_k: d_j = Re <bjbar, -i G_k>
// with every image read once; a1bar = -F0 dt b2bar + F1 dt b3bar, a2bar = dt b1bar - 2 F1 dt b3bar,
// a3bar = F0 dt b2bar + F1 dt b3bar
template <class G>
__device__ __forceinline__ void contract_m6_const(const Wave<G>& wv, const Tile<G>& b1bar, const Tile<G>& b2bar,
                                                  const Tile<G>& b3bar, const MagnusArgs& args, double* red) {
    typedef Dim<G> D;
    const double dt = args.dt;
    for (int k = 0; k < args.K; ++k) {
        double d1 = 0, d2 = 0, d3 = 0;
#pragma unroll
        for (int i = 0; i < G::NTW; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double2 e = args.g_cimg[(size_t)k * D::IMAT + wv.cimg(i, r)];
                d1 += b1bar.re[i][r] * e.y - b1bar.im[i][r] * e.x;
                d2 += b2bar.re[i][r] * e.y - b2bar.im[i][r] * e.x;
                d3 += b3bar.re[i][r] * e.y - b3bar.im[i][r] * e.x;
            }
        d1 = wave_sum(d1);
        d2 = wave_sum(d2);
        d3 = wave_sum(d3);
        if (wv.lane == 0) {
            red[(0 * 64 + k) * 4 + wv.w] = -M6_F0 * dt * d2 + M6_F1 * dt * d3;
            red[(1 * 64 + k) * 4 + wv.w] = dt * d1 - 2.0 * M6_F1 * dt * d3;
            red[(2 * 64 + k) * 4 + wv.w] = M6_F0 * dt * d2 + M6_F1 * dt * d3;
        }
    }
}

// b1, b2 -> LDS (M6), b3 in registers; every wave writes its tiles
template <class G>
__device__ __forceinline__ void m6_nodes(const Wave<G>& wv, const MagnusArgs& args, int step,
                                         const double* ctl_b, Tile<G>& b1, Tile<G>& b2, Tile<G>& b3) {
    const double dt = args.dt;
    if (args.nt == 1) {
        m6_nodes_const<G>(wv, args, step, ctl_b, b1, b2, b3);
    } else {
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
    }
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

// (n <= 32: four matrices are 66 KB - two workgroups to a CU, which is what hides the barriers)
template <class G, int NODES>
__global__ __launch_bounds__(64 * G::WAVES, G::NP <= 32 ? 2 : 1) void magnus4w_vjp_kernel(MagnusArgs args) {
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
    // mskew (skew-Hermitian generators): the skew-Hermitian part of mbar, its transposed tile straight from
    // the HBM image - the cotangent the commutator rules start from needs no exchange through LDS then
    T mbar, mskew;
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
        mskew = mbar;
        if (wv.skew) {
#pragma unroll
            for (int i = 0; i < G::NTW; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double2 e = in[(size_t)(16 * wv.tj() + wv.c) * D::IMG + 16 * wv.ti(i) + 4 * r + wv.q];
                    mskew.re[i][r] = 0.5 * (mbar.re[i][r] - e.x);
                    mskew.im[i][r] = 0.5 * (mbar.im[i][r] + e.y);
                }
        }
    }
    if (NODES == 2) {
        // a1bar = dt/2 mbar + d[a2, a1]/d a1 ; a2bar likewise (cbar = F0 dt^2 mbar)
        wv.store(node_generator<G>(wv, args, step, 0, ctl_b), L_B1);
        wv.store(node_generator<G>(wv, args, step, 1, ctl_b), L_B2);
        T cbar = mskew;
        tile_scale<G>(cbar, M4_F0 * dt * dt);
        wv.store(cbar, L_W);
        __syncthreads();
        T a2bar, a1bar;
        wv.template commutator_vjp<true>(a2bar, a1bar, L_B2, L_B1, L_W, L_X, L_B1);
        tile_axpy<G>(a1bar, 0.5 * dt, mbar);
        tile_axpy<G>(a2bar, 0.5 * dt, mbar);
        __syncthreads();  // `red` is matrix 0: every product has read it
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
        T zb = mskew;
        tile_scale<G>(zb, M6_F3);
        __syncthreads();  // b1 (s0) and b2 (s1) have been read
        wv.store(yt, S1);
        wv.store(zb, S0);
        __syncthreads();
        T xbar, ybar;
        wv.template commutator_vjp<true>(xbar, ybar, S3, S1, S0, S3, S1);
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
        T d1, wbar;  // (skew: ybar = W - W^H is skew-Hermitian as it comes, and so are xbar, wbar below)
        wv.template commutator_vjp<true>(d1, wbar, S1, S2, S0, S3, S2);
        tile_axpy<G>(b1bar, 1.0, d1);
        tile_axpy<G>(b3bar, 2.0, wbar);
        tile_axpy<G>(c12bar, 1.0, wbar);
        // c12 = [b1, b2]: b1 in s1, b2 -> s2
        __syncthreads();
        wv.store(b2, S2);
        wv.store(c12bar, S0);
        __syncthreads();
        T d2;
        wv.template commutator_vjp<true>(d1, d2, S1, S2, S0, S3, S2);
        tile_axpy<G>(b1bar, 1.0, d1);
        tile_axpy<G>(b2bar, 1.0, d2);
        // a1bar = -F0 dt b2bar + F1 dt b3bar ; a2bar = dt b1bar - 2 F1 dt b3bar ;
        // a3bar = F0 dt b2bar + F1 dt b3bar
        __syncthreads();  // `red` is matrix 0: every product has read it
        if (args.nt == 1) {
            contract_m6_const<G>(wv, b1bar, b2bar, b3bar, args, red);
        } else {
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
        __syncthreads();  // `red` is matrix 0: every product has read it
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
        __syncthreads();  // `red` is matrix 0: every product has read it
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
    if (getenv("QOCX_PRINT_OCCUPANCY")) {
        int nb = 0;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, magnus4w_vjp_kernel<G32, 3>, 256, Dim<G32>::LDS_BYTES);
        fprintf(stderr, "magnus4w_vjp_kernel<G32, 3>: %d workgroups per CU (LDS %d)\n", nb, Dim<G32>::LDS_BYTES);
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, magnus4w_fwd_kernel<G32, 3>, 256, Dim<G32>::LDS_BYTES_FWD);
        fprintf(stderr, "magnus4w_fwd_kernel<G32, 3>: %d workgroups per CU (LDS %d)\n", nb, Dim<G32>::LDS_BYTES_FWD);
    }
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
