// qocx_lindblad4t.hip - the Lindblad engine at 17 <= n <= 32 with the density tile-wise on four waves.
//
// Replaces (reference): _evaluate_lindblad_discrete (qoc/core/lindbladdiscrete.py:357-441) with its
// right-hand side _get_rhs_lindbladian (:444-495) / get_lindbladian (qoc/core/mathmethods.py:169-206) and
// integrate_rkdp5 (mathmethods.py:352-480 - here the fixed-step DOP853 of qocx_lindblad.hip on
// sub-intervals the host sizes); the gradient autograd takes through them is the discrete adjoint of the
// scheme. Density costs: targetdensityinfidelity.py:41-69, forbiddensities.py:53-85.
//
// One workgroup = one seed = four waves; wave w owns tile (w & 1, w >> 1) of every 32 x 32 matrix - of
// the density, of the stage value, of every stage derivative k_j, of the cotangents. A stage
//     k = A_L(c) Y + Y A_R(c) + sum_i gamma_i L_i Y L_i^H        (adjoint: A^H, L_i^H . L_i)
// is three rounds of LDS-operand products with one workgroup barrier before each: the generator
// products, t_i = gamma_i L_i Y (stored over the generators), t_i L_i^H (the operators two at a time: the
// last two rounds again for a third and fourth). No partial sums are exchanged -
// every wave computes ITS tile of every product - and the Runge-Kutta combinations are tile
// arithmetic: a wave writes its tile of k_j to the seed's HBM scratch and reads only that tile back (L2
// hits, no synchronisation; twelve 16 KB matrices fit neither LDS next to the operands nor - as the
// compiler allocates them - the accumulation registers). The stage loops stay ROLLED: unrolled, the
// kernel is 258 KB of straight-line code against 64 KB of instruction cache and runs at the pace of
// the instruction fetch (11 us per stage). The one-wave form of qocx_lindblad.hip (LB<2, true, false>)
// carries four tiles per matrix in one wave; it stays behind the knob lindblad_4t = 0 as the form the
// tests hold this one against.
#include "qocx_device.h"
#include "qocx_tilewave.h"
#include "dop853_tableau.h"

namespace qocx {

namespace lindblad4t {

using namespace tilewave;

typedef G32 G;
typedef Tile<G> T;
typedef Wave<G> W;
typedef Dim<G> D;

constexpr int STAGES = QOCX_RK_STAGES;
constexpr int MAT = D::IMAT;  // complex elements of a C-dump
constexpr int MAX_OPS = 4;

// LDS matrices (pitch 33): the argument of the right-hand side, the two generators at the stage's
// time (then t_0, t_1), the operators, the stage value of the control-cotangent products; behind them
// the partial sums of a reduction
// (Hermitian problems: M_GR carries X = A_L Y to the wave that needs its mirror tile, t_1 has M_T1H)
enum { M_ARG = 0, M_GL, M_GR, M_YS, M_T1H, M_OP0, M_COUNT = M_OP0 + MAX_OPS, M_T0 = M_GL, M_T1 = M_GR, M_XS = M_GR };
constexpr int RED_OFF = M_COUNT * D::MBYTES;
constexpr int LDS_BYTES = RED_OFF + 2 * 4 * 16;  // [parity][wave] of a complex scalar
static_assert(LDS_BYTES <= 160 * 1024, "one seed per CU");

// the Butcher tableau in constant memory (runtime-indexed by the rolled stage loops)
struct TableauInit {
    double a[STAGES * STAGES], b[STAGES], c[STAGES];
    constexpr TableauInit() : a(), b(), c() {
        for (int i = 0; i < STAGES; ++i) {
            b[i] = QOCX_RK_B[i];
            c[i] = QOCX_RK_C[i];
            for (int j = 0; j < STAGES; ++j) a[i * STAGES + j] = QOCX_RK_A[i][j];
        }
    }
};
__device__ __constant__ const TableauInit RK = TableauInit{};

struct Ctx {
    const LindbladArgs& a;
    W wv;
    double2* red;
    int parity;

    __device__ __forceinline__ T load_dump(const double2* d) const {
        T t;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double2 e = d[wv.cimg(0, r)];
            t.re[0][r] = e.x;
            t.im[0][r] = e.y;
        }
        return t;
    }
    __device__ __forceinline__ void store_dump(const T& t, double2* d) const {
#pragma unroll
        for (int r = 0; r < 4; ++r) d[wv.cimg(0, r)] = make_double2(t.re[0][r], t.im[0][r]);
    }
    // the sum over the workgroup of a complex scalar, in every lane (one barrier)
    __device__ __forceinline__ void block_sum(double& re, double& im) {
        re = wave_sum(re);
        im = wave_sum(im);
        double2* slot = red + parity * 4;
        parity ^= 1;
        if (wv.lane == 0) slot[wv.w] = make_double2(re, im);
        __syncthreads();
        const double2 s0 = slot[0], s1 = slot[1], s2 = slot[2], s3 = slot[3];
        re = (s0.x + s1.x) + (s2.x + s3.x);
        im = (s0.y + s1.y) + (s2.y + s3.y);
    }
    // tr(X^H Y): this wave's share
    static __device__ __forceinline__ void frob_part(const T& x, const T& y, double& re, double& im) {
        re = 0;
        im = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            re += x.re[0][r] * y.re[0][r] + x.im[0][r] * y.im[0][r];
            im += x.re[0][r] * y.im[0][r] - x.im[0][r] * y.re[0][r];
        }
    }
    static __device__ __forceinline__ void add_scaled(T& l, double zr, double zi, const T& t) {
        l.re[0] += zr * t.re[0] - zi * t.im[0];
        l.im[0] += zr * t.im[0] + zi * t.re[0];
    }

    // Density costs on the S densities in `dens` (HBM dumps); optionally adds the cotangents into
    // `lam` (qoc/standard/costs/targetdensityinfidelity.py:41-69, forbiddensities.py:53-85): as
    // density_costs of qocx_lindblad.hip, the traces summed over the four tiles.
    __device__ __forceinline__ double costs(bool step_pass, bool final_pass, const double2* dens, double2* lam) {
        const int S = a.S, n = a.n;
        double total = 0;
        for (int ci = 0; ci < a.cost_count; ++ci) {
            const DevCost c = a.costs[ci];
            const bool on = c.step_cost ? step_pass : final_pass;
            if (!on) continue;
            const double2* pool = a.cost_matrices + (size_t)c.vec_offset * MAT;
            if (c.kind == QOCX_DEV_COST_TARGET_DENSITY) {
                double fid = 0;
                for (int s = 0; s < S; ++s) {
                    const T t = load_dump(pool + (size_t)s * MAT);
                    const T rho = load_dump(dens + (size_t)s * MAT);
                    double zr, zi;
                    frob_part(t, rho, zr, zi);
                    block_sum(zr, zi);
                    const double mag = sqrt(zr * zr + zi * zi);
                    fid += mag;
                    if (lam != nullptr && mag > 0) {
                        const double f = -c.scale / ((double)S * n * mag);
                        T l = load_dump(lam + (size_t)s * MAT);
                        add_scaled(l, f * zr, f * zi, t);
                        store_dump(l, lam + (size_t)s * MAT);
                    }
                }
                total += c.scale * (1.0 - fid / ((double)S * n));
            } else {  // QOCX_DEV_COST_FORBID_DENSITY
                int base = 0;
                double acc = 0;
                for (int s = 0; s < S; ++s) {
                    const int fs = a.cost_counts[c.cnt_offset + s];
                    const T rho = load_dump(dens + (size_t)s * MAT);
                    T l = tile_zero<G>();
                    if (lam != nullptr) l = load_dump(lam + (size_t)s * MAT);
                    for (int f = 0; f < fs; ++f) {
                        const T t = load_dump(pool + (size_t)(base + f) * MAT);
                        double zr, zi;
                        frob_part(t, rho, zr, zi);
                        block_sum(zr, zi);
                        zr /= n;
                        zi /= n;
                        acc += (zr * zr + zi * zi) / fs;
                        if (lam != nullptr) {
                            const double g = 2.0 * c.scale / ((double)fs * n);
                            add_scaled(l, g * zr, g * zi, t);
                        }
                    }
                    if (lam != nullptr) store_dump(l, lam + (size_t)s * MAT);
                    base += fs;
                }
                total += c.scale * acc;
            }
        }
        return total;
    }

    // the generators of a sub-interval, linear in the stage's position c (the controls are linear in
    // time between knots): left(c) = la + c ld, right(c) = ra - c ld
    struct Gen {
        T la, ld, ra;
    };
    __device__ __forceinline__ Gen generators(const SubStep& ss, const double* ctl, bool adjoint) const {
        Gen g;
        g.la = load_dump(adjoint ? a.a0ld_cimg : a.a0l_cimg);
        g.ra = load_dump(adjoint ? a.a0rd_cimg : a.a0r_cimg);
        g.ld = tile_zero<G>();
        const int K = a.K;
        for (int k = 0; k < K; ++k) {
            const double ua = ss.wa1 * ctl[(size_t)ss.ia1 * K + k] + ss.wa2 * ctl[(size_t)ss.ia2 * K + k];
            const double ub = ss.wb1 * ctl[(size_t)ss.ib1 * K + k] + ss.wb2 * ctl[(size_t)ss.ib2 * K + k];
            const T gk = load_dump((adjoint ? a.gpd_cimg : a.gp_cimg) + (size_t)k * MAT);
            tile_axpy<G>(g.la, ua, gk);
            tile_axpy<G>(g.ra, -ua, gk);
            tile_axpy<G>(g.ld, ub - ua, gk);
        }
        return g;
    }

    // out = left(c) y + y right(c) + sum_i gamma_i L_i y L_i^H (ADJ: L_i^H y L_i; the generators are
    // then the conjugate transposes already). The operators go in pairs: t_i = gamma_i L_i y into the
    // slots of the generators, a barrier, t_i L_i^H - three workgroup barriers for up to two operators,
    // five for up to four (one more in the forward stage); y stays in M_ARG afterwards.
    // HERM (host-checked: y Hermitian, right = left^H): y right = (left y)^H - one product less, the
    // mirror tile of X = left y comes from its owner through LDS.
    // `stage` = sub-interval index * STAGES + stage index: selects the time samples of a time-dependent
    // Hamiltonian (a0_tab: A0L, A0R, A0L^H, A0R^H per stage; gp_tab: Gp, Gp^H, Gp^T per stage and
    // control) and of time-dependent lindblad_data (op_tab, gamma_tab) when the host supplied them
    // (qocx_lindblad.hip: build_generator, rhs_split); such problems take the general stages.
    template <bool ADJ, bool HERM>
    __device__ __forceinline__ T rhs(const T& y, const Gen& g, double c, const SubStep& ss, const double* ctl,
                                     size_t stage) const {
        const int nops = a.nops;
        const double* gam = a.gammas;
        T gl, gr;
        if (!HERM && a.a0_tab != nullptr) {
            const double2* t = a.a0_tab + stage * 4 * MAT;
            gl = load_dump(t + (ADJ ? 2 : 0) * (size_t)MAT);
            gr = load_dump(t + (ADJ ? 3 : 1) * (size_t)MAT);
            const int K = a.K;
            for (int k = 0; k < K; ++k) {
                const double ua = ss.wa1 * ctl[(size_t)ss.ia1 * K + k] + ss.wa2 * ctl[(size_t)ss.ia2 * K + k];
                const double ub = ss.wb1 * ctl[(size_t)ss.ib1 * K + k] + ss.wb2 * ctl[(size_t)ss.ib2 * K + k];
                const double u = (1.0 - c) * ua + c * ub;  // u(t) is linear inside a sub-interval
                const T gk = a.gp_tab != nullptr ? load_dump(a.gp_tab + ((stage * K + k) * 3 + (ADJ ? 1 : 0)) * MAT)
                                                 : load_dump((ADJ ? a.gpd_cimg : a.gp_cimg) + (size_t)k * MAT);
                tile_axpy<G>(gl, u, gk);
                tile_axpy<G>(gr, -u, gk);
            }
        } else {
            gl = g.la;
            tile_axpy<G>(gl, c, g.ld);
            if (!HERM) {
                gr = g.ra;
                tile_axpy<G>(gr, -c, g.ld);
            }
        }
        // (the last products of the previous stage read t_i from the generators' slots - and the
        // operators; the adjoint stage ends with a barrier of its own)
        if (!ADJ) __syncthreads();
        if (!HERM && a.op_tab != nullptr) {  // this stage's L_i(t), gamma_i(t)
            for (int i = 0; i < nops; ++i) wv.store(load_dump(a.op_tab + (stage * nops + i) * MAT), M_OP0 + i);
            gam = a.gamma_tab + stage * nops;
        }
        wv.store(y, M_ARG);
        wv.store(gl, M_GL);
        if (!HERM) wv.store(gr, M_GR);
        __syncthreads();
        T acc = tile_zero<G>();
        wv.template mm<false, false>(acc, M_GL, M_ARG, 1.0);
        int first = 0;  // the first operator of the pair loop below
        if (HERM) {
            wv.store(acc, M_XS);
            T t0 = tile_zero<G>();
            if (nops > 0) wv.template mm<ADJ, false>(t0, M_OP0, M_ARG, gam[0]);
            if (nops > 1) {
                T t1 = tile_zero<G>();
                wv.template mm<ADJ, false>(t1, M_OP0 + 1, M_ARG, gam[1]);
                wv.store(t1, M_T1H);
            }
            __syncthreads();  // X complete; the generator has been read: t_0 takes its place
            tile_axpy<G>(acc, 1.0, wv.load_adjoint(M_XS));
            if (nops == 0) return acc;
            wv.store(t0, M_T0);
            __syncthreads();
            wv.template mm<false, !ADJ>(acc, M_T0, M_OP0, 1.0);
            if (nops > 1) wv.template mm<false, !ADJ>(acc, M_T1H, M_OP0 + 1, 1.0);
            first = 2;
        } else {
            wv.template mm<false, false>(acc, M_ARG, M_GR, 1.0);
        }
        const int second_slot = HERM ? M_T1H : M_T1;
#pragma unroll 1
        for (int i = first; i < nops; i += 2) {
            __syncthreads();  // the generators (the previous pair's t_i) have been read
            {
                T t = tile_zero<G>();
                wv.template mm<ADJ, false>(t, M_OP0 + i, M_ARG, gam[i]);
                wv.store(t, M_T0);
            }
            if (i + 1 < nops) {
                T t = tile_zero<G>();
                wv.template mm<ADJ, false>(t, M_OP0 + i + 1, M_ARG, gam[i + 1]);
                wv.store(t, second_slot);
            }
            __syncthreads();
            wv.template mm<false, !ADJ>(acc, M_T0, M_OP0 + i, 1.0);
            if (i + 1 < nops) wv.template mm<false, !ADJ>(acc, second_slot, M_OP0 + i + 1, 1.0);
        }
        return acc;
    }
};

template <bool HERM>
__global__ __launch_bounds__(256) void lindblad4t_kernel(LindbladArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Ctx cx{a, make_wave<G>(smem, false), reinterpret_cast<double2*>(smem + RED_OFF), 0};
    const W& wv = cx.wv;
    const int S = a.S, K = a.K, nsub = a.nsub, nops = a.nops;
    const int b = blockIdx.x;
    // Two-sided evaluation (LindbladArgs::phase; one final TargetDensityInfidelity): phase 1 is the
    // forward pass alone and leaves the scalars of the final cotangents, phase 2 the adjoint of the
    // TARGETS - beside it, on other CUs - which stores its stage cotangents kbar_i instead of
    // contracting them; lindblad4t_combine_kernel does that afterwards over the whole chip.
    const int phase = a.phase;
    double2* dens = a.scratch + (size_t)b * (2 * S + 3 * STAGES) * MAT;
    double2* lam = dens + (size_t)S * MAT;
    // k_j / Ybar_j: [STAGES] dumps, a set of its own for the pass that runs beside the forward
    double2* kdump = lam + (size_t)S * MAT + (phase == 2 ? (size_t)STAGES * MAT : 0);
    // stage values recomputed from a checkpoint when the batch's did not fit into HBM (ystages == nullptr)
    double2* ysdump = lam + (size_t)S * MAT + 2 * (size_t)STAGES * MAT;
    double2* ckpt_b = a.checkpoints + (size_t)b * nsub * S * MAT;
    const double* ctl = a.controls + (size_t)b * a.nc * K;

    for (int i = 0; i < nops; ++i) wv.store(cx.load_dump(a.op_cimg + (size_t)i * MAT), M_OP0 + i);
    if (phase != 2)
        for (int s = 0; s < S; ++s) cx.store_dump(cx.load_dump(a.rho0_cimg + (size_t)s * MAT), dens + (size_t)s * MAT);
    __syncthreads();

    // ---- forward ----------------------------------------------------------------------------------
    // (a wave reads back from `dens`, `lam`, `ckpt` only the tile it wrote itself: no fences needed)
    // the twelve stages of sub-interval q on y0 (in: the density at its start, out: at its end); the
    // stage values go to `ys` if that is not null
    auto forward_stages = [&](const SubStep& ss, const Ctx::Gen& g, int q, T& y0, double2* ys) {
        const double h = ss.h;
        T klast = tile_zero<G>();
#pragma unroll 1
        for (int i = 0; i < STAGES; ++i) {
            T y = y0;
#pragma unroll 1
            for (int j = 0; j + 1 < i; ++j) {
                const double aij = RK.a[i * STAGES + j];
                if (aij != 0.0) tile_axpy<G>(y, h * aij, cx.load_dump(kdump + (size_t)j * MAT));
            }
            if (i > 0) tile_axpy<G>(y, h * RK.a[i * STAGES + i - 1], klast);
            if (ys != nullptr) cx.store_dump(y, ys + (size_t)i * MAT);
            klast = cx.template rhs<false, HERM>(y, g, RK.c[i], ss, ctl, (size_t)q * STAGES + i);
            cx.store_dump(klast, kdump + (size_t)i * MAT);
        }
#pragma unroll 1
        for (int i = 0; i < STAGES; ++i) {
            const double bi = RK.b[i];
            if (bi != 0.0) tile_axpy<G>(y0, h * bi, cx.load_dump(kdump + (size_t)i * MAT));
        }
    };
    // lambda += host-supplied cotangent of the densities at system step `step`, if there is one
    auto inject = [&](int step) {
        if (a.inj_index == nullptr) return;
        const int row = a.inj_index[step];
        if (row < 0) return;
        for (int s = 0; s < S; ++s) {
            T l = cx.load_dump(lam + (size_t)s * MAT);
            tile_axpy<G>(l, 1.0, cx.load_dump(a.inj_bars + (((size_t)b * a.inj_count + row) * S + s) * MAT));
            cx.store_dump(l, lam + (size_t)s * MAT);
        }
    };
    double cost = 0;
    for (int q = 0; q < (phase == 2 ? 0 : nsub); ++q) {
        const SubStep ss = a.substeps[q];
        if (ss.first_of_step) {
            if (a.has_step_costs && ss.step != 0 && (ss.step % a.cost_eval_step) == 0)
                cost += cx.costs(true, false, dens, nullptr);
            if (a.step_densities != nullptr)
                for (int s = 0; s < S; ++s)
                    cx.store_dump(cx.load_dump(dens + (size_t)s * MAT),
                                  a.step_densities + (((size_t)b * (a.nsteps + 1) + ss.step) * S + s) * MAT);
        }
        const Ctx::Gen g = cx.generators(ss, ctl, false);
        for (int s = 0; s < S; ++s) {
            T y0 = cx.load_dump(dens + (size_t)s * MAT);
            cx.store_dump(y0, ckpt_b + ((size_t)q * S + s) * MAT);
            double2* ys = a.ystages != nullptr ? a.ystages + ((((size_t)b * nsub + q) * S + s) * STAGES) * MAT : nullptr;
            forward_stages(ss, g, q, y0, ys);
            cx.store_dump(y0, dens + (size_t)s * MAT);
        }
    }
    if (phase != 2) {
        if (a.has_step_costs && (a.nsteps % a.cost_eval_step) == 0) cost += cx.costs(true, false, dens, nullptr);
        cost += cx.costs(false, true, dens, nullptr);
        if (wv.tid == 0) a.cost_out[b] = cost;
        for (int s = 0; s < S; ++s) {
            const T rho = cx.load_dump(dens + (size_t)s * MAT);
            cx.store_dump(rho, a.final_out + ((size_t)b * S + s) * MAT);
            if (a.step_densities != nullptr)
                cx.store_dump(rho, a.step_densities + (((size_t)b * (a.nsteps + 1) + a.nsteps) * S + s) * MAT);
            if (phase == 1) {
                // cotangent of final density s = (f zr + i f zi) T_s, f = -scale / (S n |z|), z = tr(T_s^H rho_s)
                // (costs()): the scalars the combine kernel applies
                const DevCost c = a.costs[0];
                double zr, zi;
                Ctx::frob_part(cx.load_dump(a.cost_matrices + ((size_t)c.vec_offset + s) * MAT), rho, zr, zi);
                cx.block_sum(zr, zi);
                const double mag = sqrt(zr * zr + zi * zi);
                const double f = mag > 0 ? -c.scale / ((double)S * a.n * mag) : 0.0;
                if (wv.tid == 0) a.lam_scale[(size_t)b * S + s] = make_double2(f * zr, f * zi);
            }
        }
    }
    if (!a.want_grad || phase == 1) return;

    // ---- discrete adjoint -------------------------------------------------------------------------
    __syncthreads();  // (the forward's last products are done with the slots)
    if (phase == 2) {  // lambda_s = the target of density s
        for (int s = 0; s < S; ++s)
            cx.store_dump(cx.load_dump(a.cost_matrices + ((size_t)a.costs[0].vec_offset + s) * MAT), lam + (size_t)s * MAT);
    } else {
        for (int s = 0; s < S; ++s) cx.store_dump(tile_zero<G>(), lam + (size_t)s * MAT);
        (void)cx.costs((a.nsteps % a.cost_eval_step) == 0, true, dens, lam);
        inject(a.nsteps);
    }
    for (int q = nsub - 1; q >= 0; --q) {
        const SubStep ss = a.substeps[q];
        const Ctx::Gen g = cx.generators(ss, ctl, true);
        // control cotangents at the two ends of the sub-interval: per-lane partial sums
        double ga[QOCX_LINDBLAD_MAX_K], gb[QOCX_LINDBLAD_MAX_K];
#pragma unroll
        for (int k = 0; k < QOCX_LINDBLAD_MAX_K; ++k) {
            ga[k] = 0;
            gb[k] = 0;
        }
        for (int s = 0; s < S; ++s) {
            const size_t stage0 = ((((size_t)b * nsub + q) * S + s) * STAGES) * MAT;
            const double2* ys = a.ystages + stage0;
            if (a.ystages == nullptr) {  // the stage values again, from the checkpoint
                const Ctx::Gen gf = cx.generators(ss, ctl, false);
                T y0 = cx.load_dump(ckpt_b + ((size_t)q * S + s) * MAT);
                forward_stages(ss, gf, q, y0, ysdump);
                ys = ysdump;
                __syncthreads();  // (the forward's last products are done with the slots)
            }
            const T lambda = cx.load_dump(lam + (size_t)s * MAT);
            T lambda_new = lambda;
            const double h = ss.h;
            T yblast = tile_zero<G>();
#pragma unroll 1
            for (int i = STAGES - 1; i >= 0; --i) {
                // kbar_i = h (b_i lambda + sum_{j > i} a_ji Ybar_j)
                const double ci = RK.c[i];
                T kb = lambda;
                tile_scale<G>(kb, h * RK.b[i]);
#pragma unroll 1
                for (int j = i + 2; j < STAGES; ++j) {
                    const double aji = RK.a[j * STAGES + i];
                    if (aji != 0.0) tile_axpy<G>(kb, h * aji, cx.load_dump(kdump + (size_t)j * MAT));
                }
                if (i + 1 < STAGES) tile_axpy<G>(kb, h * RK.a[(i + 1) * STAGES + i], yblast);
                if (phase == 2) cx.store_dump(kb, a.kbstages + stage0 + (size_t)i * MAT);
                else wv.store(cx.load_dump(ys + (size_t)i * MAT), M_YS);
                yblast = cx.template rhs<true, HERM>(kb, g, ci, ss, ctl, (size_t)q * STAGES + i);
                cx.store_dump(yblast, kdump + (size_t)i * MAT);
                tile_axpy<G>(lambda_new, 1.0, yblast);
                if (phase == 2) {
                    __syncthreads();  // M_ARG free for the next stage
                    continue;
                }
                // control cotangent of this stage: Re <kbar, Gp_k Y - Y Gp_k> = Re tr(Z Gp_k),
                // Z = Y kbar^H - kbar^H Y (kbar is still in M_ARG, Y_i in M_YS since before rhs's barriers).
                // HERM: Z = W - W^H with W = Y kbar, and Re tr(W^H Gp_k) = -Re tr(W Gp_k) as Gp_k^H = -Gp_k
                T z = tile_zero<G>();
                if (HERM) {
                    wv.template mm<false, false>(z, M_YS, M_ARG, 2.0);
                } else {
                    wv.template mm<false, true>(z, M_YS, M_ARG, 1.0);
                    wv.template mm<true, false>(z, M_ARG, M_YS, -1.0);
                }
#pragma unroll
                for (int k = 0; k < QOCX_LINDBLAD_MAX_K; ++k)
                    if (k < K) {
                        const T gt = (!HERM && a.gp_tab != nullptr)
                                         ? cx.load_dump(a.gp_tab + ((((size_t)q * STAGES + i) * K + k) * 3 + 2) * MAT)
                                         : cx.load_dump(a.gpt_cimg + (size_t)k * MAT);
                        double pr = 0;
#pragma unroll
                        for (int r = 0; r < 4; ++r) pr += z.re[0][r] * gt.re[0][r] - z.im[0][r] * gt.im[0][r];
                        ga[k] += (1.0 - ci) * pr;
                        gb[k] += ci * pr;
                    }
                __syncthreads();  // M_ARG, M_YS free for the next stage
            }
            cx.store_dump(lambda_new, lam + (size_t)s * MAT);
        }
        if (phase == 2) continue;
#pragma unroll
        for (int k = 0; k < QOCX_LINDBLAD_MAX_K; ++k)
            if (k < K) {
                double x = ga[k], y = gb[k];
                cx.block_sum(x, y);
                if (wv.tid == 0) {
                    a.gsub[(((size_t)b * nsub + q) * 2 + 0) * K + k] = x;
                    a.gsub[(((size_t)b * nsub + q) * 2 + 1) * K + k] = y;
                }
            }
        // step costs are evaluated on the densities at the START of their system step
        if (ss.first_of_step && ss.step != 0) {
            if ((ss.step % a.cost_eval_step) == 0 && a.has_step_costs) {
                for (int s = 0; s < S; ++s)
                    cx.store_dump(cx.load_dump(ckpt_b + ((size_t)q * S + s) * MAT), dens + (size_t)s * MAT);
                (void)cx.costs(true, false, dens, lam);
            }
            inject(ss.step);
        }
    }
}

// Two-sided evaluation, third kernel: the control cotangents of sub-interval q of seed b from the stage
// values Y_i (phase 1) and the stage cotangents kbar_i of the unit adjoint (phase 2),
//     g_k += Re( conj(c_s) tr(Z_i Gp_k) ),  Z_i = Y_i kbar_i^H - kbar_i^H Y_i,  c_s = lam_scale[b][s],
// with the weights (1 - c_i, c_i) of the sub-interval's end points: gsub[B][nsub][2][K] as the classic
// launch writes it. HERM: Z = W - W^H, W = Y kbar, and tr(Z Gp_k) = 2 Re tr(W Gp_k). Throughput work on
// the whole chip: nsub x B workgroups of four waves, two LDS matrices each.
template <bool HERM>
__global__ __launch_bounds__(256) void lindblad4t_combine_kernel(LindbladArgs a) {
    __shared__ __attribute__((aligned(16))) char smem[2 * D::MBYTES + 2 * 4 * 16];
    Ctx cx{a, make_wave<G>(smem, false), reinterpret_cast<double2*>(smem + 2 * D::MBYTES), 0};
    const W& wv = cx.wv;
    const int q = blockIdx.x, b = blockIdx.y, S = a.S, K = a.K, nsub = a.nsub;
    double ga[QOCX_LINDBLAD_MAX_K], gb[QOCX_LINDBLAD_MAX_K];
#pragma unroll
    for (int k = 0; k < QOCX_LINDBLAD_MAX_K; ++k) {
        ga[k] = 0;
        gb[k] = 0;
    }
    for (int s = 0; s < S; ++s) {
        const double2 cs = a.lam_scale[(size_t)b * S + s];
        const size_t base = ((((size_t)b * nsub + q) * S + s) * STAGES) * MAT;
#pragma unroll 1
        for (int i = 0; i < STAGES; ++i) {
            wv.store(cx.load_dump(a.ystages + base + (size_t)i * MAT), 0);
            wv.store(cx.load_dump(a.kbstages + base + (size_t)i * MAT), 1);
            __syncthreads();
            T z = tile_zero<G>();
            if (HERM) {
                wv.template mm<false, false>(z, 0, 1, 2.0);
            } else {
                wv.template mm<false, true>(z, 0, 1, 1.0);
                wv.template mm<true, false>(z, 1, 0, -1.0);
            }
            const double ci = RK.c[i];
#pragma unroll
            for (int k = 0; k < QOCX_LINDBLAD_MAX_K; ++k)
                if (k < K) {
                    const T gt = cx.load_dump(a.gpt_cimg + (size_t)k * MAT);
                    double pr = 0, pi = 0;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        pr += z.re[0][r] * gt.re[0][r] - z.im[0][r] * gt.im[0][r];
                        pi += z.re[0][r] * gt.im[0][r] + z.im[0][r] * gt.re[0][r];
                    }
                    // Re(conj(c) gamma); HERM: gamma = 2 Re tr(W Gp_k) is real
                    const double gk = HERM ? cs.x * pr : fma(cs.y, pi, cs.x * pr);
                    ga[k] += (1.0 - ci) * gk;
                    gb[k] += ci * gk;
                }
            __syncthreads();  // the two matrices are free again
        }
    }
#pragma unroll
    for (int k = 0; k < QOCX_LINDBLAD_MAX_K; ++k)
        if (k < K) {
            double x = ga[k], y = gb[k];
            cx.block_sum(x, y);
            if (wv.tid == 0) {
                a.gsub[(((size_t)b * nsub + q) * 2 + 0) * K + k] = x;
                a.gsub[(((size_t)b * nsub + q) * 2 + 1) * K + k] = y;
            }
        }
}

}  // namespace lindblad4t

bool lindblad4t_supports(const LindbladArgs& a) {
    return a.tile4 && a.n > 16 && a.n <= 32 && (a.phase == 0 || a.ystages != nullptr) &&
           (!a.hermitian || (a.a0_tab == nullptr && a.inj_index == nullptr)) && a.nops <= lindblad4t::MAX_OPS &&
           a.scratch != nullptr && a.K <= QOCX_LINDBLAD_MAX_K;
}

void launch_lindblad4t_combine(const LindbladArgs& a, int batch, hipStream_t st) {
    if (a.hermitian)
        hipLaunchKernelGGL(lindblad4t::lindblad4t_combine_kernel<true>, dim3(a.nsub, batch), dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL(lindblad4t::lindblad4t_combine_kernel<false>, dim3(a.nsub, batch), dim3(256), 0, st, a);
}

void launch_lindblad4t(const LindbladArgs& a, int batch, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(lindblad4t::lindblad4t_kernel<false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lindblad4t::LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(lindblad4t::lindblad4t_kernel<true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lindblad4t::LDS_BYTES);
        attr_set = true;
    }
    if (a.hermitian)
        hipLaunchKernelGGL(lindblad4t::lindblad4t_kernel<true>, dim3(batch), dim3(256), lindblad4t::LDS_BYTES, st, a);
    else
        hipLaunchKernelGGL(lindblad4t::lindblad4t_kernel<false>, dim3(batch), dim3(256), lindblad4t::LDS_BYTES, st, a);
}

}  // namespace qocx
