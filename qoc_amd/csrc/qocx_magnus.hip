// qocx_magnus.hip - Magnus generators of order four and six and their reverse rules.
//
// Reference: magnus_m4 / magnus_m6 (qoc/core/mathmethods.py:96-122, :125-164) applied to
// a(t) = -i H(u(t), t) inside _evolve_step_schroedinger_discrete
// (qoc/core/schroedingerdiscrete.py:483-497); autograd differentiates them op by op.
//
// M2 is linear in the controls and is fused into K1a / K3 (qocx_kernels.hip). M4 and M6 contain
// commutators of the node generators, so they get two kernels of their own:
//   magnus_fwd_kernel : controls -> M_j            (row-major padded NP x NP, feeds
//                                                    pade_pq_explicit_kernel and K3)
//   magnus_vjp_kernel : (controls, Mbar_j) -> d cost / d u_k at every quadrature node
// One wavefront per propagator step; matrices are C-layout register tiles, every product is
// the LDS-staged complex GEMM on v_mfma_f64_16x16x4_f64 of qocx_wave.h. Named intermediates
// live in per-block HBM scratch as lane-linear dumps (each lane reloads exactly what it
// stored; cross-lane exchange happens only through the LDS operand slots).
#include "qocx_wave.h"

namespace qocx {

namespace {

constexpr double M4_F0 = 0.14433756729740643;  // sqrt(3)/12
constexpr double M6_F0 = 1.2909944487358056;   // sqrt(15)/3
constexpr double M6_F1 = 10.0 / 3.0;
constexpr double M6_F2 = 0.5;
constexpr double M6_F3 = 1.0 / 240.0;
constexpr double M6_F4 = 1.0 / 60.0;

template <int NB>
struct MagnusLds {
    static constexpr int SLOT = 2 * Geo<NB>::PLANE * 8;
    static constexpr int BYTES = 2 * SLOT;  // two planar operand slots
};

template <int NB>
struct Ctx {
    typedef Geo<NB> G;
    typedef CMat<NB> Mat;
    double* are;  // slot A
    double* aim;
    double* bre;  // slot B
    double* bim;
    // Every node generator is skew-Hermitian (Hermitian H0(t), G_k(t): checked on the host). Then
    // so are b1, b2, b3, every commutator and M itself, and [X, Y] = XY - (XY)^H: ONE product per
    // commutator instead of two. In the reverse pass only the skew-Hermitian part of a cotangent
    // can reach the controls (the contraction is with -i G_k, skew-Hermitian, and Hermitian is
    // orthogonal to skew-Hermitian under Re tr(A^H B)); the forward maps send skew perturbations
    // to skew ones, so their adjoints may project: with Zs = (Zbar - Zbar^H) / 2,
    //   Xbar = [Y, Zs] = V - V^H, V = Y Zs ;  Ybar = [Zs, X] = W - W^H, W = Zs X :
    // two products instead of four.
    bool skew;

    __device__ __forceinline__ void dump_store(const Mat& m, double2* d) const {
        const int lane = lane_id();
#pragma unroll
        for (int ti = 0; ti < NB; ++ti)
#pragma unroll
            for (int tj = 0; tj < NB; ++tj)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    d[((ti * NB + tj) * 4 + r) * 64 + lane] =
                        make_double2(m.re[ti][tj][r], m.im[ti][tj][r]);
    }
    __device__ __forceinline__ void dump_load(Mat& m, const double2* d) const {
        const int lane = lane_id();
#pragma unroll
        for (int ti = 0; ti < NB; ++ti)
#pragma unroll
            for (int tj = 0; tj < NB; ++tj)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double2 e = d[((ti * NB + tj) * 4 + r) * 64 + lane];
                    m.re[ti][tj][r] = e.x;
                    m.im[ti][tj][r] = e.y;
                }
    }
    // y += alpha * x
    __device__ __forceinline__ static void axpy(Mat& y, double alpha, const Mat& x) {
#pragma unroll
        for (int ti = 0; ti < NB; ++ti)
#pragma unroll
            for (int tj = 0; tj < NB; ++tj) {
                y.re[ti][tj] += alpha * x.re[ti][tj];
                y.im[ti][tj] += alpha * x.im[ti][tj];
            }
    }
    // C-layout registers of M^H from a planar image of M
    __device__ __forceinline__ void load_adjoint(Mat& m, const double* lre, const double* lim) const {
        const int q = lane_id() >> 4, c = lane_id() & 15;
#pragma unroll
        for (int ti = 0; ti < NB; ++ti)
#pragma unroll
            for (int tj = 0; tj < NB; ++tj)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int off = (16 * tj + c) * G::PITCH + 16 * ti + 4 * r + q;
                    m.re[ti][tj][r] = lre[off];
                    m.im[ti][tj][r] = -lim[off];
                }
    }
    // acc += sign * L^H * R with L given as a planar image, R in registers
    __device__ __forceinline__ void gemm_adj_left(Mat& acc, const double* lre, const double* lim,
                                                  const Mat& right, double sign) const {
        const int q = lane_id() >> 4, c = lane_id() & 15;
#pragma unroll
        for (int kk = 0; kk < 4 * NB; ++kk) {
            double ar[NB], ai[NB], ni[NB];
#pragma unroll
            for (int ti = 0; ti < NB; ++ti) {
                const int off = (4 * kk + q) * G::PITCH + 16 * ti + c;  // conj(L[4kk+q][16ti+c])
                ar[ti] = sign * lre[off];
                ai[ti] = -sign * lim[off];
                ni[ti] = -ai[ti];
            }
#pragma unroll
            for (int tj = 0; tj < NB; ++tj) {
                const double br = right.re[kk >> 2][tj][kk & 3];
                const double bi = right.im[kk >> 2][tj][kk & 3];
#pragma unroll
                for (int ti = 0; ti < NB; ++ti) {
                    acc.re[ti][tj] = mfma_f64(ar[ti], br, acc.re[ti][tj]);
                    acc.re[ti][tj] = mfma_f64(ni[ti], bi, acc.re[ti][tj]);
                    acc.im[ti][tj] = mfma_f64(ar[ti], bi, acc.im[ti][tj]);
                    acc.im[ti][tj] = mfma_f64(ai[ti], br, acc.im[ti][tj]);
                }
            }
        }
    }
    // acc += sign * L * R, L planar, R registers
    __device__ __forceinline__ void gemm(Mat& acc, const double* lre, const double* lim,
                                         const Mat& right, double sign) const {
        zgemm_acc<NB>(acc, lre, lim, [&](int kk, int tj, double& br, double& bi) {
            br = sign * right.re[kk >> 2][tj][kk & 3];
            bi = sign * right.im[kk >> 2][tj][kk & 3];
        });
    }
    // out = X Y - Y X   (convenience.py:16-29)
    // out -= out^H (through slot B)
    __device__ __forceinline__ void minus_own_adjoint(Mat& out) const {
        wave_sync();
        cmat_to_lds<NB>(out, bre, bim);
        wave_sync();
        Mat t;
        load_adjoint(t, bre, bim);
        axpy(out, -1.0, t);
        wave_sync();
    }
    __device__ __forceinline__ void commutator(Mat& out, const Mat& x, const Mat& y) const {
        wave_sync();
        cmat_to_lds<NB>(x, are, aim);
        if (!skew) cmat_to_lds<NB>(y, bre, bim);
        wave_sync();
        cmat_zero<NB>(out);
        gemm(out, are, aim, y, 1.0);
        if (skew) {
            minus_own_adjoint(out);  // Y X = (X Y)^H
            return;
        }
        gemm(out, bre, bim, x, -1.0);
        wave_sync();
    }
    // cotangents of Z = X Y - Y X:  Xbar = Zbar Y^H - Y^H Zbar ;  Ybar = X^H Zbar - Zbar X^H.
    // X and Y are given as scratch dumps and loaded only while needed (register pressure).
    __device__ __forceinline__ void commutator_vjp(Mat& xbar, Mat& ybar, const double2* x_dump,
                                                   const double2* y_dump, const Mat& zbar) const {
        if (skew) {
            Mat zs = zbar;  // Zs = (Zbar - Zbar^H) / 2
            minus_own_adjoint(zs);
            cmat_scale<NB>(zs, 0.5);
            {
                Mat y;
                dump_load(y, y_dump);
                wave_sync();
                cmat_to_lds<NB>(y, are, aim);
                wave_sync();
            }
            cmat_zero<NB>(xbar);
            gemm(xbar, are, aim, zs, 1.0);  // V = Y Zs
            minus_own_adjoint(xbar);
            wave_sync();
            cmat_to_lds<NB>(zs, are, aim);
            wave_sync();
            {
                Mat x;
                dump_load(x, x_dump);
                cmat_zero<NB>(ybar);
                gemm(ybar, are, aim, x, 1.0);  // W = Zs X
            }
            minus_own_adjoint(ybar);
            return;
        }
        wave_sync();
        cmat_to_lds<NB>(zbar, are, aim);
        {
            Mat y;
            dump_load(y, y_dump);
            cmat_to_lds<NB>(y, bre, bim);
        }
        wave_sync();
        {
            Mat t;
            load_adjoint(t, bre, bim);  // Y^H
            cmat_zero<NB>(xbar);
            gemm(xbar, are, aim, t, 1.0);  // Zbar Y^H
        }
        gemm_adj_left(xbar, bre, bim, zbar, -1.0);  // - Y^H Zbar
        wave_sync();
        {
            Mat x;
            dump_load(x, x_dump);
            cmat_to_lds<NB>(x, bre, bim);
        }
        wave_sync();
        cmat_zero<NB>(ybar);
        gemm_adj_left(ybar, bre, bim, zbar, 1.0);  // X^H Zbar
        {
            Mat t;
            load_adjoint(t, bre, bim);  // X^H
            gemm(ybar, are, aim, t, -1.0);  // - Zbar X^H
        }
        wave_sync();
    }
};

// a_q = -i (H0 + sum_k u_k(t_q) G_k) at quadrature node `node` of step `step`
template <int NB>
__device__ __forceinline__ void node_generator(CMat<NB>& a, const MagnusArgs& args, int step,
                                               int node, const double* ctl_b) {
    typedef Geo<NB> G;
    const int lane = lane_id();
    const size_t col = (size_t)step * args.nodes + node;
    const size_t tsel = (args.nt == 1) ? 0 : col;
    const double2* h0 = args.h0_cimg + tsel * G::MAT;
    const double2* g = args.g_cimg + tsel * args.K * G::MAT;
    const StepInterp si = args.interp[col];
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
    for (int k = 0; k < args.K; ++k) {
        const double uk = control_at(ctl_b, si, args.K, k);
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
            a.re[ti][tj] = hm.im[ti][tj];
            a.im[ti][tj] = -hm.re[ti][tj];
        }
}

template <int NB>
__device__ __forceinline__ void store_row_major(const CMat<NB>& m, double2* out) {
    typedef Geo<NB> G;
    const int q = lane_id() >> 4, c = lane_id() & 15;
#pragma unroll
    for (int ti = 0; ti < NB; ++ti)
#pragma unroll
        for (int tj = 0; tj < NB; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                out[(size_t)(16 * ti + 4 * r + q) * G::NP + 16 * tj + c] =
                    make_double2(m.re[ti][tj][r], m.im[ti][tj][r]);
}

template <int NB>
__device__ __forceinline__ void load_row_major(CMat<NB>& m, const double2* in) {
    typedef Geo<NB> G;
    const int q = lane_id() >> 4, c = lane_id() & 15;
#pragma unroll
    for (int ti = 0; ti < NB; ++ti)
#pragma unroll
        for (int tj = 0; tj < NB; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double2 e = in[(size_t)(16 * ti + 4 * r + q) * G::NP + 16 * tj + c];
                m.re[ti][tj][r] = e.x;
                m.im[ti][tj][r] = e.y;
            }
}

// scratch dump slots of one block
enum { S_B1 = 0, S_B2, S_B3, S_X, S_W, S_Y, S_MBAR, S_B1BAR, S_B2BAR, S_B3BAR, S_C12BAR, S_COUNT };

// M6 intermediates -> scratch (b1, b2, b3, x, w, y); returns m. Matrices are staged through the
// scratch dumps so that at most four tiles are live in registers.
template <int NB>
__device__ __forceinline__ void m6_forward(CMat<NB>& m, const Ctx<NB>& cx, const MagnusArgs& args,
                                           int step, const double* ctl_b, double2* scr) {
    typedef Geo<NB> G;
    typedef CMat<NB> Mat;
    const double dt = args.dt;
    {
        // b2 = F0 dt (a3 - a1) ; b3 = F1 dt (a3 - 2 a2 + a1) ; b1 = dt a2   (mathmethods.py:153-155)
        Mat b2, b3;
        {
            Mat a;
            node_generator<NB>(a, args, step, 0, ctl_b);
            cmat_zero<NB>(b2);
            Ctx<NB>::axpy(b2, -M6_F0 * dt, a);
            cmat_zero<NB>(b3);
            Ctx<NB>::axpy(b3, M6_F1 * dt, a);
        }
        {
            Mat a;
            node_generator<NB>(a, args, step, 1, ctl_b);
            Ctx<NB>::axpy(b3, -2.0 * M6_F1 * dt, a);
            cmat_scale<NB>(a, dt);
            cx.dump_store(a, scr + (size_t)S_B1 * G::MAT);
        }
        {
            Mat a;
            node_generator<NB>(a, args, step, 2, ctl_b);
            Ctx<NB>::axpy(b2, M6_F0 * dt, a);
            Ctx<NB>::axpy(b3, M6_F1 * dt, a);
        }
        cx.dump_store(b2, scr + (size_t)S_B2 * G::MAT);
        cx.dump_store(b3, scr + (size_t)S_B3 * G::MAT);
    }
    {
        // c12 = [b1, b2] ; w = 2 b3 + c12 ; x = -20 b1 - b3 + c12
        Mat c12;
        {
            Mat b1, b2;
            cx.dump_load(b1, scr + (size_t)S_B1 * G::MAT);
            cx.dump_load(b2, scr + (size_t)S_B2 * G::MAT);
            cx.commutator(c12, b1, b2);
        }
        Mat b3;
        cx.dump_load(b3, scr + (size_t)S_B3 * G::MAT);
        {
            Mat w = c12;
            Ctx<NB>::axpy(w, 2.0, b3);
            cx.dump_store(w, scr + (size_t)S_W * G::MAT);
        }
        Ctx<NB>::axpy(c12, -1.0, b3);
        {
            Mat b1;
            cx.dump_load(b1, scr + (size_t)S_B1 * G::MAT);
            Ctx<NB>::axpy(c12, -20.0, b1);
        }
        cx.dump_store(c12, scr + (size_t)S_X * G::MAT);
    }
    {
        // y = b2 - F4 [b1, w]
        Mat y;
        {
            Mat b1, w;
            cx.dump_load(b1, scr + (size_t)S_B1 * G::MAT);
            cx.dump_load(w, scr + (size_t)S_W * G::MAT);
            cx.commutator(y, b1, w);
        }
        cmat_scale<NB>(y, -M6_F4);
        {
            Mat b2;
            cx.dump_load(b2, scr + (size_t)S_B2 * G::MAT);
            Ctx<NB>::axpy(y, 1.0, b2);
        }
        cx.dump_store(y, scr + (size_t)S_Y * G::MAT);
        // m = b1 + F2 b3 + F3 [x, y]
        Mat x;
        cx.dump_load(x, scr + (size_t)S_X * G::MAT);
        cx.commutator(m, x, y);
    }
    cmat_scale<NB>(m, M6_F3);
    {
        Mat t;
        cx.dump_load(t, scr + (size_t)S_B1 * G::MAT);
        Ctx<NB>::axpy(m, 1.0, t);
        cx.dump_load(t, scr + (size_t)S_B3 * G::MAT);
        Ctx<NB>::axpy(m, M6_F2, t);
    }
}

// g_k = Re <abar, -i G_k> for every control of one node
template <int NB>
__device__ __forceinline__ void contract_node(const CMat<NB>& abar, const MagnusArgs& args,
                                              size_t m, int step, int node) {
    typedef Geo<NB> G;
    const int lane = lane_id();
    const size_t col = (size_t)step * args.nodes + node;
    const size_t tsel = (args.nt == 1) ? 0 : col;
    const double2* g = args.g_cimg + tsel * args.K * G::MAT;
    for (int k = 0; k < args.K; ++k) {
        const double2* gk = g + (size_t)k * G::MAT;
        double acc = 0;
#pragma unroll
        for (int ti = 0; ti < NB; ++ti)
#pragma unroll
            for (int tj = 0; tj < NB; ++tj)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double2 e = gk[((ti * NB + tj) * 4 + r) * 64 + lane];
                    acc += abar.re[ti][tj][r] * e.y - abar.im[ti][tj][r] * e.x;
                }
        acc = wave_sum(acc);
        if (lane == 0)
            args.gstep[((m / args.nsteps) * (size_t)args.nsteps * args.nodes + col) * args.K + k] = acc;
    }
}

}  // namespace

template <int NB, int NODES>
__global__ __launch_bounds__(64) void magnus_fwd_kernel(MagnusArgs args) {
    typedef Geo<NB> G;
    typedef CMat<NB> Mat;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Ctx<NB> cx;
    cx.are = reinterpret_cast<double*>(smem);
    cx.aim = cx.are + G::PLANE;
    cx.bre = cx.aim + G::PLANE;
    cx.bim = cx.bre + G::PLANE;
    cx.skew = args.skew != 0;
    double2* scr = args.scratch + (size_t)blockIdx.x * S_COUNT * G::MAT;
    for (size_t w = blockIdx.x; w < args.total; w += gridDim.x) {
        const int step = args.step0 + (int)(w % args.seg_len);
        const size_t b = w / args.seg_len;
        const size_t m = b * args.nsteps + step;
        const double* ctl_b = args.controls + b * args.nc * args.K;
        Mat mm;
        if (NODES == 2) {
            // m4 = dt/2 (a1 + a2) + F0 dt^2 [a2, a1]   (mathmethods.py:119-121)
            Mat a1, a2;
            node_generator<NB>(a1, args, step, 0, ctl_b);
            node_generator<NB>(a2, args, step, 1, ctl_b);
            cx.commutator(mm, a2, a1);
            cmat_scale<NB>(mm, M4_F0 * args.dt * args.dt);
            Ctx<NB>::axpy(mm, 0.5 * args.dt, a1);
            Ctx<NB>::axpy(mm, 0.5 * args.dt, a2);
        } else {
            m6_forward<NB>(mm, cx, args, step, ctl_b, scr);
        }
        store_row_major<NB>(mm, args.m_rm + m * G::MAT);
    }
}

template <int NB, int NODES>
__global__ __launch_bounds__(64) void magnus_vjp_kernel(MagnusArgs args) {
    typedef Geo<NB> G;
    typedef CMat<NB> Mat;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Ctx<NB> cx;
    cx.are = reinterpret_cast<double*>(smem);
    cx.aim = cx.are + G::PLANE;
    cx.bre = cx.aim + G::PLANE;
    cx.bim = cx.bre + G::PLANE;
    cx.skew = args.skew != 0;
    double2* scr = args.scratch + (size_t)blockIdx.x * S_COUNT * G::MAT;
    const double dt = args.dt;
    for (size_t w = blockIdx.x; w < args.total; w += gridDim.x) {
        const int step = args.step0 + (int)(w % args.seg_len);
        const size_t b = w / args.seg_len;
        const size_t m = b * args.nsteps + step;
        const double* ctl_b = args.controls + b * args.nc * args.K;
        if (NODES == 2) {
            // a1bar = dt/2 mbar + d[a2,a1]/da1 ; a2bar likewise (cbar = F0 dt^2 mbar)
            {
                Mat a;
                node_generator<NB>(a, args, step, 0, ctl_b);
                cx.dump_store(a, scr + (size_t)S_B1 * G::MAT);
                node_generator<NB>(a, args, step, 1, ctl_b);
                cx.dump_store(a, scr + (size_t)S_B2 * G::MAT);
            }
            Mat mbar, a1bar, a2bar;
            load_row_major<NB>(mbar, args.mbar_rm + m * G::MAT);
            cmat_scale<NB>(mbar, M4_F0 * dt * dt);
            cx.commutator_vjp(a2bar, a1bar, scr + (size_t)S_B2 * G::MAT,
                              scr + (size_t)S_B1 * G::MAT, mbar);
            const double back = 0.5 / (M4_F0 * dt);  // (dt/2) / (F0 dt^2)
            Ctx<NB>::axpy(a1bar, back, mbar);
            Ctx<NB>::axpy(a2bar, back, mbar);
            contract_node<NB>(a1bar, args, m, step, 0);
            contract_node<NB>(a2bar, args, m, step, 1);
            continue;
        }
        // ---- M6 (the reverse of m6_forward) ---------------------------------------------
        {
            Mat unused;
            m6_forward<NB>(unused, cx, args, step, ctl_b, scr);
        }
        {
            // (xbar, ybar) = vjp of [x, y] with F3 mbar
            Mat zbar, xbar, ybar;
            load_row_major<NB>(zbar, args.mbar_rm + m * G::MAT);
            cmat_scale<NB>(zbar, M6_F3);
            cx.commutator_vjp(xbar, ybar, scr + (size_t)S_X * G::MAT, scr + (size_t)S_Y * G::MAT,
                              zbar);
            // b1bar = mbar - 20 xbar ; b3bar = F2 mbar - xbar ; c12bar = xbar ; b2bar = ybar
            cx.dump_store(ybar, scr + (size_t)S_B2BAR * G::MAT);
            cx.dump_store(xbar, scr + (size_t)S_C12BAR * G::MAT);
            load_row_major<NB>(zbar, args.mbar_rm + m * G::MAT);
            Mat t = zbar;
            Ctx<NB>::axpy(t, -20.0, xbar);
            cx.dump_store(t, scr + (size_t)S_B1BAR * G::MAT);
            cmat_scale<NB>(zbar, M6_F2);
            Ctx<NB>::axpy(zbar, -1.0, xbar);
            cx.dump_store(zbar, scr + (size_t)S_B3BAR * G::MAT);
        }
        {
            // y = b2 - F4 [b1, w]: innerbar = -F4 ybar ; (d1, wbar) = vjp of [b1, w]
            Mat inner, d1, wbar;
            cx.dump_load(inner, scr + (size_t)S_B2BAR * G::MAT);
            cmat_scale<NB>(inner, -M6_F4);
            cx.commutator_vjp(d1, wbar, scr + (size_t)S_B1 * G::MAT, scr + (size_t)S_W * G::MAT,
                              inner);
            Mat t;
            cx.dump_load(t, scr + (size_t)S_B1BAR * G::MAT);
            Ctx<NB>::axpy(t, 1.0, d1);
            cx.dump_store(t, scr + (size_t)S_B1BAR * G::MAT);
            cx.dump_load(t, scr + (size_t)S_B3BAR * G::MAT);
            Ctx<NB>::axpy(t, 2.0, wbar);
            cx.dump_store(t, scr + (size_t)S_B3BAR * G::MAT);
            cx.dump_load(t, scr + (size_t)S_C12BAR * G::MAT);
            Ctx<NB>::axpy(t, 1.0, wbar);
            cx.dump_store(t, scr + (size_t)S_C12BAR * G::MAT);
        }
        {
            // c12 = [b1, b2]
            Mat c12bar, d1, d2;
            cx.dump_load(c12bar, scr + (size_t)S_C12BAR * G::MAT);
            cx.commutator_vjp(d1, d2, scr + (size_t)S_B1 * G::MAT, scr + (size_t)S_B2 * G::MAT,
                              c12bar);
            Mat t;
            cx.dump_load(t, scr + (size_t)S_B1BAR * G::MAT);
            Ctx<NB>::axpy(t, 1.0, d1);
            cx.dump_store(t, scr + (size_t)S_B1BAR * G::MAT);
            cx.dump_load(t, scr + (size_t)S_B2BAR * G::MAT);
            Ctx<NB>::axpy(t, 1.0, d2);
            cx.dump_store(t, scr + (size_t)S_B2BAR * G::MAT);
        }
        // a1bar = -F0 dt b2bar + F1 dt b3bar ; a2bar = dt b1bar - 2 F1 dt b3bar ;
        // a3bar = F0 dt b2bar + F1 dt b3bar
        {
            Mat b2bar, b3bar, abar;
            cx.dump_load(b2bar, scr + (size_t)S_B2BAR * G::MAT);
            cx.dump_load(b3bar, scr + (size_t)S_B3BAR * G::MAT);
            cmat_zero<NB>(abar);
            Ctx<NB>::axpy(abar, -M6_F0 * dt, b2bar);
            Ctx<NB>::axpy(abar, M6_F1 * dt, b3bar);
            contract_node<NB>(abar, args, m, step, 0);
            cmat_zero<NB>(abar);
            Ctx<NB>::axpy(abar, M6_F0 * dt, b2bar);
            Ctx<NB>::axpy(abar, M6_F1 * dt, b3bar);
            contract_node<NB>(abar, args, m, step, 2);
            cx.dump_load(b2bar, scr + (size_t)S_B1BAR * G::MAT);  // b1bar
            cmat_zero<NB>(abar);
            Ctx<NB>::axpy(abar, dt, b2bar);
            Ctx<NB>::axpy(abar, -2.0 * M6_F1 * dt, b3bar);
            contract_node<NB>(abar, args, m, step, 1);
        }
    }
}

size_t magnus_scratch_elems(int nb, int blocks) {
    return (size_t)blocks * S_COUNT * (size_t)(256 * nb * nb);
}

// NB = 4 (33 <= n <= 64): the two operand slots are 135 KiB, and a C-layout matrix is 256 registers
// of the one wave, so these instantiations live on scratch (6-13 KB per lane). They exist so that M4 /
// M6 work at those sizes at all; the fast path there is M2.
template <class Kernel>
static void magnus_lds_attr(Kernel kernel, int bytes) {
    if (bytes > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

template <int NB>
static void launch_fwd_t(const MagnusArgs& a, int blocks, hipStream_t st) {
    magnus_lds_attr(magnus_fwd_kernel<NB, 2>, MagnusLds<NB>::BYTES);
    magnus_lds_attr(magnus_fwd_kernel<NB, 3>, MagnusLds<NB>::BYTES);
    if (a.nodes == 2)
        hipLaunchKernelGGL((magnus_fwd_kernel<NB, 2>), dim3(blocks), dim3(64), MagnusLds<NB>::BYTES,
                           st, a);
    else
        hipLaunchKernelGGL((magnus_fwd_kernel<NB, 3>), dim3(blocks), dim3(64), MagnusLds<NB>::BYTES,
                           st, a);
}
template <int NB>
static void launch_vjp_t(const MagnusArgs& a, int blocks, hipStream_t st) {
    magnus_lds_attr(magnus_vjp_kernel<NB, 2>, MagnusLds<NB>::BYTES);
    magnus_lds_attr(magnus_vjp_kernel<NB, 3>, MagnusLds<NB>::BYTES);
    if (a.nodes == 2)
        hipLaunchKernelGGL((magnus_vjp_kernel<NB, 2>), dim3(blocks), dim3(64), MagnusLds<NB>::BYTES,
                           st, a);
    else
        hipLaunchKernelGGL((magnus_vjp_kernel<NB, 3>), dim3(blocks), dim3(64), MagnusLds<NB>::BYTES,
                           st, a);
}

// ---- M4, time-independent H0 / G_k: effective controls and their chain rule (M4LinArgs) -------
// One thread per (seed, step). mathmethods.py:96-122 with a(t) = -i (H0 + sum u_k(t) G_k).
__global__ __launch_bounds__(256) void m4lin_controls_kernel(M4LinArgs args) {
    const size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= args.total) return;
    const int step = (int)(w % args.nsteps), K = args.K;
    const size_t b = w / args.nsteps;
    const double* ctl_b = args.controls + b * args.nc * K;
    const StepInterp s1 = args.interp[(size_t)step * 2], s2 = args.interp[(size_t)step * 2 + 1];
    double u1[QOCX_M4LIN_MAX_K], u2[QOCX_M4LIN_MAX_K];
    double* v = args.veff + w * args.Ke;
    for (int k = 0; k < K; ++k) {
        u1[k] = control_at(ctl_b, s1, K, k);
        u2[k] = control_at(ctl_b, s2, K, k);
        v[k] = 0.5 * (u1[k] + u2[k]);
        v[K + k] = args.f0dt * (u2[k] - u1[k]);
    }
    int e = 2 * K;
    for (int k = 0; k < K; ++k)
        for (int l = k + 1; l < K; ++l) v[e++] = args.f0dt * (u2[k] * u1[l] - u2[l] * u1[k]);
}

__global__ __launch_bounds__(256) void m4lin_chain_kernel(M4LinArgs args) {
    const size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= args.total) return;
    const int step = (int)(w % args.nsteps), K = args.K, Ke = args.Ke;
    const size_t b = w / args.nsteps;
    const double* ctl_b = args.controls + b * args.nc * K;
    const StepInterp s1 = args.interp[(size_t)step * 2], s2 = args.interp[(size_t)step * 2 + 1];
    double u1[QOCX_M4LIN_MAX_K], u2[QOCX_M4LIN_MAX_K], g1[QOCX_M4LIN_MAX_K], g2[QOCX_M4LIN_MAX_K];
    // cotangent of effective control e
    auto ge = [&](int e) -> double {
        if (args.lam_scale != nullptr) {  // unit adjoint: Re(conj(c) gamma), as scatter_kernel
            const double2 c = args.lam_scale[b * args.S];
            const double* g = args.gstep + (w * Ke + e) * 2;
            return fma(c.y, g[1], c.x * g[0]);
        }
        return args.gstep[w * Ke + e];
    };
    for (int k = 0; k < K; ++k) {
        u1[k] = control_at(ctl_b, s1, K, k);
        u2[k] = control_at(ctl_b, s2, K, k);
        const double gv = ge(k), gw = ge(K + k);
        g1[k] = 0.5 * gv - args.f0dt * gw;
        g2[k] = 0.5 * gv + args.f0dt * gw;
    }
    int e = 2 * K;
    for (int k = 0; k < K; ++k)
        for (int l = k + 1; l < K; ++l) {
            const double gz = args.f0dt * ge(e++);  // z = F0 dt (u2_k u1_l - u2_l u1_k)
            g1[l] += gz * u2[k];
            g1[k] -= gz * u2[l];
            g2[k] += gz * u1[l];
            g2[l] -= gz * u1[k];
        }
    double* out = args.gnode + (b * (size_t)args.nsteps * 2 + (size_t)step * 2) * K;
    for (int k = 0; k < K; ++k) {
        out[k] = g1[k];
        out[K + k] = g2[k];
    }
}

void launch_m4lin_controls(const M4LinArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(m4lin_controls_kernel, dim3((unsigned)((a.total + 255) / 256)), dim3(256), 0, st, a);
}
void launch_m4lin_chain(const M4LinArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(m4lin_chain_kernel, dim3((unsigned)((a.total + 255) / 256)), dim3(256), 0, st, a);
}

void launch_magnus_fwd(int nb, const MagnusArgs& a, int blocks, hipStream_t st) {
    if (nb == 1) launch_fwd_t<1>(a, blocks, st);
    else if (nb == 2) launch_fwd_t<2>(a, blocks, st);
    else launch_fwd_t<4>(a, blocks, st);
}

void launch_magnus_vjp(int nb, const MagnusArgs& a, int blocks, hipStream_t st) {
    if (nb == 1) launch_vjp_t<1>(a, blocks, st);
    else if (nb == 2) launch_vjp_t<2>(a, blocks, st);
    else launch_vjp_t<4>(a, blocks, st);
}

}  // namespace qocx
