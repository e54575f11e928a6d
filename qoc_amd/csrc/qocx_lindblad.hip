// qocx_lindblad.hip - hand-written CDNA4 (gfx950) kernel of the Lindblad GRAPE path.
//
// Reference: _evaluate_lindblad_discrete (qoc/core/lindbladdiscrete.py:357-441) integrates
//     d rho/dt = -i[H, rho] + sum_i gamma_i (L_i rho L_i^H - 1/2 {L_i^H L_i, rho})
// (get_lindbladian, qoc/core/mathmethods.py:169-206) with an adaptive RK5(4) restarted at every
// system step, and autograd differentiates through that integrator. The device integrates the
// SAME matrix-form equation, rewritten as
//     d rho/dt = A_L rho + rho A_R + sum_i gamma_i L_i rho L_i^H,
//     A_L = -i H(u(t)) - 1/2 sum_i gamma_i L_i^H L_i,   A_R = +i H(u(t)) - 1/2 sum_i gamma_i L_i^H L_i,
// with the FIXED-step 12-stage Dormand-Prince 8(5,3) scheme on host-chosen sub-intervals that
// never straddle a control knot (so u(t) is linear inside each), and applies the exact discrete
// adjoint of that scheme from per-sub-interval checkpoints and (when they fit HBM) the stored
// stage values (tests/lindblad_model.py is the NumPy model of exactly this). All products are
// n x n: one MFMA tile for n <= 16, four for n <= 32; every right-hand side is 2 + 2L complex
// GEMMs on v_mfma_f64_16x16x4_f64.
//
// One wavefront per seed; the S densities of a seed advance together, sub-interval by
// sub-interval (the recursion in time is serial).
#include <cstddef>
#include "dop853_tableau.h"
#include "qocx_wave.h"

namespace qocx {

namespace {

constexpr int STAGES = QOCX_RK_STAGES;

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
constexpr TableauInit TABLEAU{};
__device__ __constant__ const TableauInit TABLEAU_DEV = TABLEAU;
#define RK_A_DEV TABLEAU_DEV.a
#define RK_B_DEV TABLEAU_DEV.b
#define RK_C_DEV TABLEAU_DEV.c

struct Slot {
    double* re;
    double* im;
};

// Everything is a member of this template: LNB = 1 (n <= 16, one MFMA tile per matrix) or
// LNB = 2 (n <= 32, four tiles). GS: the stage derivatives, the densities and the cotangents
// live in per-seed HBM scratch instead of LDS (always for LNB = 2: 12 x 16 KB do not fit).
// MW: several wavefronts work on one seed - wave 0 the generator terms A_L y + y A_R, wave 1 + i
// the term of Lindblad operator i, the last wave the control cotangents of the adjoint - and
// exchange their partial right-hand sides through LDS (two workgroup barriers per stage).
template <int LNB, bool GS, bool MW, bool RG = false, bool STAMP = false, bool QP = false, bool Q2 = false,
          bool CH = false>
struct LB {
typedef Geo<LNB> LG;
typedef CMat<LNB> Mat;                               // C-layout register tiles
static constexpr int MAT = 256 * LNB * LNB;          // complex elements of one matrix dump
static constexpr int LPLANE = LG::PLANE;             // doubles per LDS plane
static constexpr int SLOT_BYTES = 2 * LPLANE * 8;    // planar left-operand slot
static constexpr int DUMP_BYTES = MAT * 16;          // C-layout dump of one matrix (lane-linear)
// register-resident stage loops: one tile per matrix, densities in LDS
static constexpr bool REG = RG && (LNB == 1) && !GS;
// four waves with a quarter of every stage vector each (nops = 2)
static constexpr bool QUARTER = QP && MW && (LNB == 1) && !GS;

static __device__ __forceinline__ Slot slot_at(char* base) {
    Slot s;
    s.re = reinterpret_cast<double*>(base);
    s.im = s.re + LPLANE;
    return s;
}

// C-layout dump: reg r of tile (ti, tj) of lane l <-> complex index ((ti LNB + tj) 4 + r) 64 + l
static __device__ __forceinline__ void dump_store(const Mat& m, double2* d) {
    const int lane = lane_id();
#pragma unroll
    for (int ti = 0; ti < LNB; ++ti)
#pragma unroll
        for (int tj = 0; tj < LNB; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                d[((ti * LNB + tj) * 4 + r) * 64 + lane] =
                    make_double2(m.re[ti][tj][r], m.im[ti][tj][r]);
}
static __device__ __forceinline__ void dump_load(Mat& m, const double2* d) {
    const int lane = lane_id();
#pragma unroll
    for (int ti = 0; ti < LNB; ++ti)
#pragma unroll
        for (int tj = 0; tj < LNB; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double2 e = d[((ti * LNB + tj) * 4 + r) * 64 + lane];
                m.re[ti][tj][r] = e.x;
                m.im[ti][tj][r] = e.y;
            }
}

// C-layout registers of M^H from the planar image of M
static __device__ __forceinline__ void load_adjoint(Mat& m, const Slot& s) {
    const int q = lane_id() >> 4, c = lane_id() & 15;
#pragma unroll
    for (int ti = 0; ti < LNB; ++ti)
#pragma unroll
        for (int tj = 0; tj < LNB; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                // element (R, C) of M^H = conj M[C][R], R = 16 ti + 4 r + q, C = 16 tj + c
                const int off = (16 * tj + c) * LG::PITCH + 16 * ti + 4 * r + q;
                m.re[ti][tj][r] = s.re[off];
                m.im[ti][tj][r] = -s.im[off];
            }
}

// C-layout registers of M from the planar image of M
static __device__ __forceinline__ void load_plain(Mat& m, const Slot& s) {
    const int q = lane_id() >> 4, c = lane_id() & 15;
#pragma unroll
    for (int ti = 0; ti < LNB; ++ti)
#pragma unroll
        for (int tj = 0; tj < LNB; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int off = (16 * ti + 4 * r + q) * LG::PITCH + 16 * tj + c;
                m.re[ti][tj][r] = s.re[off];
                m.im[ti][tj][r] = s.im[off];
            }
}

static __device__ __forceinline__ void mat_zero(Mat& m) { cmat_zero<LNB>(m); }
static __device__ __forceinline__ void mat_axpy(Mat& y, double a, const Mat& x) {
#pragma unroll
    for (int ti = 0; ti < LNB; ++ti)
#pragma unroll
        for (int tj = 0; tj < LNB; ++tj) {
            y.re[ti][tj] += a * x.re[ti][tj];
            y.im[ti][tj] += a * x.im[ti][tj];
        }
}

// acc += Left * right, Left = the planar slot (or its conjugate transpose), right in registers.
// One tile (n <= 16): the 3M scheme, T1 = Ar Br, T2 = Ai Bi, T3 = (Ar + Ai)(Br + Bi),
// Re = T1 - T2, Im = T3 - T1 - T2: three independent MFMA chains of four instead of two chains
// of eight, a quarter fewer MFMAs (the kernel is bound by the matrix pipe of its one wave).
template <bool LEFT_ADJ>
static __device__ __forceinline__ void gemm(Mat& acc, const Slot& left, const Mat& right) {
    const int q = lane_id() >> 4, c = lane_id() & 15;
    if constexpr (LNB == 1) {
        d4 t1 = {0, 0, 0, 0}, t2 = {0, 0, 0, 0}, t3 = {0, 0, 0, 0};
        // all eight operand reads first, then twelve MFMAs back to back (three independent chains)
        double are[4], aim[4], asum[4], bsum[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int off = LEFT_ADJ ? ((4 * kk + q) * LG::PITCH + c) : (c * LG::PITCH + 4 * kk + q);
            are[kk] = left.re[off];
            aim[kk] = LEFT_ADJ ? -left.im[off] : left.im[off];
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            asum[kk] = are[kk] + aim[kk];
            bsum[kk] = right.re[0][0][kk] + right.im[0][0][kk];
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            t1 = mfma_f64(are[kk], right.re[0][0][kk], t1);
            t2 = mfma_f64(aim[kk], right.im[0][0][kk], t2);
            t3 = mfma_f64(asum[kk], bsum[kk], t3);
        }
        acc.re[0][0] += t1 - t2;
        acc.im[0][0] += t3 - t1 - t2;
    } else {
        // four tiles, one wave (17 <= n <= 32): the kernel is bound by the matrix pipe of its one wave
        // (768 MFMAs per right-hand side with four real products per complex one), so the 3M scheme
        // here too - tile by tile, three accumulators at a time: a quarter fewer MFMAs for twice the
        // (cheap) reads of the left operand
#pragma unroll
        for (int tj = 0; tj < LNB; ++tj)
#pragma unroll
            for (int ti = 0; ti < LNB; ++ti) {
                d4 t1 = {0, 0, 0, 0}, t2 = {0, 0, 0, 0}, t3 = {0, 0, 0, 0};
#pragma unroll
                for (int kk = 0; kk < 4 * LNB; ++kk) {
                    const int off = LEFT_ADJ ? ((4 * kk + q) * LG::PITCH + 16 * ti + c)
                                             : ((16 * ti + c) * LG::PITCH + 4 * kk + q);
                    const double are = left.re[off];
                    const double aim = LEFT_ADJ ? -left.im[off] : left.im[off];
                    const double bre = right.re[kk >> 2][tj][kk & 3], bim = right.im[kk >> 2][tj][kk & 3];
                    t1 = mfma_f64(are, bre, t1);
                    t2 = mfma_f64(aim, bim, t2);
                    t3 = mfma_f64(are + aim, bre + bim, t3);
                }
                acc.re[ti][tj] += t1 - t2;
                acc.im[ti][tj] += t3 - t1 - t2;
            }
    }
}

// l += (zr + i zi) * t
static __device__ __forceinline__ void add_scaled(Mat& l, double zr, double zi, const Mat& t) {
#pragma unroll
    for (int ti = 0; ti < LNB; ++ti)
#pragma unroll
        for (int tj = 0; tj < LNB; ++tj) {
            l.re[ti][tj] += zr * t.re[ti][tj] - zi * t.im[ti][tj];
            l.im[ti][tj] += zr * t.im[ti][tj] + zi * t.re[ti][tj];
        }
}

// <X, Y> = sum conj(X) Y over the whole matrix, wave-uniform
static __device__ __forceinline__ void frob_inner(const Mat& x, const Mat& y, double& re, double& im) {
    double pr = 0, pi = 0;
#pragma unroll
    for (int ti = 0; ti < LNB; ++ti)
#pragma unroll
        for (int tj = 0; tj < LNB; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pr += x.re[ti][tj][r] * y.re[ti][tj][r] + x.im[ti][tj][r] * y.im[ti][tj][r];
                pi += x.re[ti][tj][r] * y.im[ti][tj][r] - x.im[ti][tj][r] * y.re[ti][tj][r];
            }
    re = wave_sum(pr);
    im = wave_sum(pi);
}

// Density costs on the S densities in `dens` (LDS dumps); optionally adds the cotangents into
// `lam`. qoc/standard/costs/targetdensityinfidelity.py:41-69, forbiddensities.py:53-85.
static __device__ __forceinline__ double density_costs(const LindbladArgs& a, bool step_pass,
                                                bool final_pass, const double2* dens,
                                                double2* lam) {
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
                Mat t, rho;
                dump_load(t, pool + (size_t)s * MAT);
                dump_load(rho, dens + (size_t)s * MAT);
                double zr, zi;
                frob_inner(t, rho, zr, zi);  // tr(T^H rho)
                const double mag = sqrt(zr * zr + zi * zi);
                fid += mag;
                if (lam != nullptr && mag > 0) {
                    const double f = -c.scale / ((double)S * n * mag);
                    Mat l;
                    dump_load(l, lam + (size_t)s * MAT);
                    add_scaled(l, f * zr, f * zi, t);
                    dump_store(l, lam + (size_t)s * MAT);
                }
            }
            total += c.scale * (1.0 - fid / ((double)S * n));
        } else {  // QOCX_DEV_COST_FORBID_DENSITY
            int base = 0;
            double acc = 0;
            for (int s = 0; s < S; ++s) {
                const int fs = a.cost_counts[c.cnt_offset + s];
                Mat rho, l;
                dump_load(rho, dens + (size_t)s * MAT);
                if (lam != nullptr) dump_load(l, lam + (size_t)s * MAT);
                for (int f = 0; f < fs; ++f) {
                    Mat t;
                    dump_load(t, pool + (size_t)(base + f) * MAT);
                    double zr, zi;
                    frob_inner(t, rho, zr, zi);
                    zr /= n;
                    zi /= n;
                    acc += (zr * zr + zi * zi) / fs;
                    if (lam != nullptr) {
                        const double g = 2.0 * c.scale / ((double)fs * n);
                        add_scaled(l, g * zr, g * zi, t);
                    }
                }
                if (lam != nullptr) dump_store(l, lam + (size_t)s * MAT);
                base += fs;
            }
            total += c.scale * acc;
        }
    }
    wave_sync();
    return total;
}

// LDS carve (bytes): 3 planar work slots | per operator: planar L | and unless GS: S density
// dumps | S lambda dumps | STAGES stage-derivative dumps. 80 KB at LNB = 1, S = 1, L = 2: two
// seeds per CU.
static __host__ __device__ int waves(int nops) { return MW ? nops + 2 : 1; }
// the chain form of the two-sided stage loop (substep_chain): waves per seed
static __host__ __device__ int chain_waves(int) { return 4; }
static __host__ __device__ int lds_bytes(int S, int nops, int cached_controls = -1) {
    // work slots: single wave gen | y | tmp; MW: gen | y | tmp per operator | zk | zy ; then the
    // operator images, MW: one partial-result dump per wave, the per-seed dumps, and (MW,
    // cached_controls = K >= 0) copies of the constant generator dumps A0L A0R A0L^H A0R^H and
    // Gp Gp^H Gp^T per control
    const int slots = MW ? (2 + nops + 2) : 3;
    return slots * SLOT_BYTES + nops * SLOT_BYTES + (MW ? 2 * waves(nops) * DUMP_BYTES : 0) +
           (GS ? 0 : (2 * S + STAGES) * DUMP_BYTES) +
           (cached_controls >= 0 ? (4 + 3 * cached_controls) * DUMP_BYTES : 0);
}
static __device__ __forceinline__ void block_sync() {
    if (MW) __syncthreads();
    else wave_sync();
}

// Everything one wave needs. The stage loops are plain runtime loops over the Butcher tableau
// (constant memory); the stage derivatives live in `kdump`, not in registers.
struct Wave {
    const LindbladArgs& a;
    int wv, nwaves;                   // this wave's index in the block, waves per seed
    Slot slot_gen, slot_y;            // wave 0: A_L (or A_L^H) and the argument as left operands
    Slot slot_tmp;                    // this wave's gamma_i L_i Y
    Slot slot_zk, slot_zy;            // control-cotangent wave: kbar and Y as left operands
    char* op_planar;                  // [L] planar images of L_i
    double2* parts;                   // MW: [nwaves] partial right-hand sides
    double2* kdump;  // STAGES dumps: stage derivatives k_j, then (adjoint) Ybar_j
    const double* ctl_b;
    // constant generator dumps: the HBM images, or their copies in LDS (MW with room to spare)
    const double2 *c_a0l, *c_a0r, *c_a0ld, *c_a0rd, *c_gp, *c_gpd, *c_gpt;
    StampClock<STAMP>* clk;  // diagnostic build: cycles per phase of a stage (this wave)

    __device__ __forceinline__ bool first() const { return !MW || wv == 0; }
    __device__ __forceinline__ bool z_wave() const { return !MW || wv == nwaves - 1; }

    // Generators of one stage. The reference applies -i[H, rho] with the SAME H on both sides
    // (mathmethods.py:188), so the left and right factors are kept separate (they are adjoints
    // of each other only for Hermitian H):
    //   forward : left = A_L = A0L + sum u_k Gp_k ,  right = A_R = A0R - sum u_k Gp_k , Gp = -i G
    //   adjoint : left = A_L^H,  right = A_R^H  (images of the conjugate transposes)
    // `stage` = sub-interval index * STAGES + stage index: selects the time samples of a
    // time-dependent Hamiltonian (a0_tab: A0L, A0R, A0L^H, A0R^H per stage; gp_tab: Gp, Gp^H, Gp^T
    // per stage and control) when the host supplied them.
    __device__ __forceinline__ void build_generator(const SubStep& ss, size_t stage, double c,
                                                    bool adjoint, Mat& left, Mat& right) const {
        if (a.a0_tab != nullptr) {
            const double2* t = a.a0_tab + stage * 4 * MAT;
            dump_load(left, t + (adjoint ? 2 : 0) * (size_t)MAT);
            dump_load(right, t + (adjoint ? 3 : 1) * (size_t)MAT);
        } else {
            dump_load(left, adjoint ? c_a0ld : c_a0l);
            dump_load(right, adjoint ? c_a0rd : c_a0r);
        }
        const int K = a.K;
        for (int k = 0; k < K; ++k) {
            const double ua = ss.wa1 * ctl_b[(size_t)ss.ia1 * K + k] + ss.wa2 * ctl_b[(size_t)ss.ia2 * K + k];
            const double ub = ss.wb1 * ctl_b[(size_t)ss.ib1 * K + k] + ss.wb2 * ctl_b[(size_t)ss.ib2 * K + k];
            const double u = (1.0 - c) * ua + c * ub;  // u(t) is linear inside a sub-interval
            Mat g;
            if (a.gp_tab != nullptr)
                dump_load(g, a.gp_tab + ((stage * K + k) * 3 + (adjoint ? 1 : 0)) * MAT);
            else
                dump_load(g, (adjoint ? c_gpd : c_gp) + (size_t)k * MAT);
            mat_axpy(left, u, g);
            mat_axpy(right, -u, g);
        }
    }

    // out = Gen y + y GenRight + sum_i gamma_i Op_i y Op_i^H   (ADJ: Gen^H, GenRight^H,
    // Op_i^H y Op_i). Single wave: everything; MW: each wave its share, summed through `parts`
    // (one workgroup barrier inside; every wave returns the full result).
    template <bool ADJ>
    __device__ __forceinline__ void rhs_split(Mat& out, const Mat& y, const SubStep& ss,
                                              size_t stage, double c) const {
        Mat acc;
        mat_zero(acc);
        if (first()) {
            Mat gl, gr;
            build_generator(ss, stage, c, ADJ, gl, gr);
            wave_sync();
            cmat_to_lds<LNB>(gl, slot_gen.re, slot_gen.im);
            cmat_to_lds<LNB>(y, slot_y.re, slot_y.im);
            wave_sync();
            gemm<false>(acc, slot_gen, y);
            gemm<false>(acc, slot_y, gr);
        }
        for (int i = 0; i < a.nops; ++i) {
            if (MW && wv != 1 + i) continue;
            Mat t, opr;
            mat_zero(t);
            const Slot op = slot_at(op_planar + (size_t)i * SLOT_BYTES);
            double gamma = a.gammas[i];
            if (a.op_tab != nullptr) {
                // time-dependent lindblad_data: this stage's L_i(t) replaces the image in LDS
                // (the slot of operator i is read by the wave that owns the operator only)
                Mat opm;
                dump_load(opm, a.op_tab + (stage * a.nops + i) * MAT);
                gamma = a.gamma_tab[stage * a.nops + i];
                wave_sync();
                cmat_to_lds<LNB>(opm, op.re, op.im);
                wave_sync();
            }
            gemm<ADJ>(t, op, y);
            cmat_scale<LNB>(t, gamma);
            wave_sync();
            cmat_to_lds<LNB>(t, slot_tmp.re, slot_tmp.im);
            wave_sync();
            if (ADJ) load_plain(opr, op);
            else load_adjoint(opr, op);
            gemm<false>(acc, slot_tmp, opr);
        }
        wave_sync();
        if (MW) {
            // (the last wave - control cotangents - has no share of the right-hand side)
            if (wv < nwaves - 1) dump_store(acc, parts + (size_t)wv * MAT);
            __syncthreads();
            mat_zero(out);
            for (int w = 0; w < nwaves - 1; ++w) {
                Mat p;
                dump_load(p, parts + (size_t)w * MAT);
                mat_axpy(out, 1.0, p);
            }
        } else {
            out = acc;
        }
    }

    // ---- register-resident stages (n <= 16, stage values kept in HBM, constant H0 / G_k) ------
    // Inside one sub-interval the controls are linear in time, so the generators of its 12
    // stages are A(c) = A_a + c dA: built ONCE per sub-interval (3 matrices = 48 registers) instead
    // of 2 + K dump loads and axpys per stage.
    // (kept as three LDS dumps in kdump[0..2], which the register-resident loops leave unused:
    // 48 registers less per wave; only the first wave reads them)
    struct GenLin {
        double2* d;  // la | ld | ra : left(c) = la + c ld, right(c) = ra - c ld
    };
    // u_k at the two ends of a sub-interval (linear interpolation between control knots)
    __device__ __forceinline__ void end_controls(const SubStep& ss, const double* ctl, int k, double& ua,
                                                 double& ub) const {
        const int K = a.K;
        ua = ss.wa1 * ctl[(size_t)ss.ia1 * K + k] + ss.wa2 * ctl[(size_t)ss.ia2 * K + k];
        ub = ss.wb1 * ctl[(size_t)ss.ib1 * K + k] + ss.wb2 * ctl[(size_t)ss.ib2 * K + k];
    }
    // `ua`, `ub` (substep_q2): the end-point controls, fetched one sub-interval ahead by run()
    __device__ __forceinline__ void build_linear(const SubStep& ss, bool adjoint, GenLin& out,
                                                 const double* ua_pre = nullptr,
                                                 const double* ub_pre = nullptr) const {
        struct { Mat la, ld, ra; } g;
        out.d = kdump;
        if (ua_pre == nullptr) {
            dump_load(g.la, adjoint ? c_a0ld : c_a0l);
            dump_load(g.ra, adjoint ? c_a0rd : c_a0r);
        }
        mat_zero(g.ld);
        const int K = a.K;
        if (ua_pre != nullptr) {
            // (Q2: the pass's constant dumps sit in kdump[6 ..], see run() - named here so that the
            // reads are LDS instructions, not flat ones that would wait for the loads in flight)
            const double2* cache = kdump + 6 * (size_t)MAT;
            dump_load(g.la, cache);
            dump_load(g.ra, cache + MAT);
#pragma unroll
            for (int k = 0; k < QOCX_LINDBLAD_MAX_K; ++k)  // (unrolled: the arrays stay in registers)
                if (k < K) {
                    Mat gk;
                    dump_load(gk, cache + (size_t)(2 + k) * MAT);
                    mat_axpy(g.la, ua_pre[k], gk);
                    mat_axpy(g.ra, -ua_pre[k], gk);
                    mat_axpy(g.ld, ub_pre[k] - ua_pre[k], gk);
                }
        } else {
            for (int k = 0; k < K; ++k) {
                double ua, ub;
                end_controls(ss, ctl_b, k, ua, ub);
                Mat gk;
                dump_load(gk, (adjoint ? c_gpd : c_gp) + (size_t)k * MAT);
                mat_axpy(g.la, ua, gk);
                mat_axpy(g.ra, -ua, gk);
                mat_axpy(g.ld, ub - ua, gk);
            }
        }
        wave_sync();
        dump_store(g.la, out.d);
        dump_store(g.ld, out.d + MAT);
        dump_store(g.ra, out.d + 2 * (size_t)MAT);
        wave_sync();
    }

    // rhs_split with the generator from `gen` and the partial sums in the `parity` set of `parts`
    // (two sets: ONE workgroup barrier per stage instead of two)
    // `opr` (multi-wave form): the right-operand image of THIS wave's operator (L^H forward, L
    // adjoint), loaded once per sub-interval instead of once per stage
    template <bool ADJ>
    __device__ __forceinline__ void rhs_lin(Mat& out, const Mat& y, const GenLin& gen, double c,
                                            int parity, const Mat& opr_mw) const {
        Mat acc;
        mat_zero(acc);
        clk->lap(0);  // stage value (axpys), kbar
        if (first()) {
            Mat gl, gr, gd;
            dump_load(gl, gen.d);
            dump_load(gd, gen.d + MAT);
            dump_load(gr, gen.d + 2 * (size_t)MAT);
            mat_axpy(gl, c, gd);
            mat_axpy(gr, -c, gd);
            wave_sync();
            cmat_to_lds<LNB>(gl, slot_gen.re, slot_gen.im);
            cmat_to_lds<LNB>(y, slot_y.re, slot_y.im);
            wave_sync();
            clk->lap(1);  // generator of the stage, operands to LDS
            gemm<false>(acc, slot_gen, y);
            gemm<false>(acc, slot_y, gr);
            clk->lap(2);  // products
        }
        for (int i = 0; i < a.nops; ++i) {
            if (MW && wv != 1 + i) continue;
            Mat t, opr;
            mat_zero(t);
            const Slot op = slot_at(op_planar + (size_t)i * SLOT_BYTES);
            gemm<ADJ>(t, op, y);
            cmat_scale<LNB>(t, a.gammas[i]);
            wave_sync();
            cmat_to_lds<LNB>(t, slot_tmp.re, slot_tmp.im);
            wave_sync();
            clk->lap(1);
            if (MW) {
                gemm<false>(acc, slot_tmp, opr_mw);
                clk->lap(2);
            } else {
                if (ADJ) load_plain(opr, op);
                else load_adjoint(opr, op);
                gemm<false>(acc, slot_tmp, opr);
            }
        }
        wave_sync();
        if (MW) {
            double2* set = parts + (size_t)parity * nwaves * MAT;
            if (wv < nwaves - 1) dump_store(acc, set + (size_t)wv * MAT);
            clk->lap(3);  // partial result to LDS
            __syncthreads();
            clk->lap(4);  // barrier
            mat_zero(out);
            for (int w = 0; w < nwaves - 1; ++w) {
                Mat p;
                dump_load(p, set + (size_t)w * MAT);
                mat_axpy(out, 1.0, p);
            }
            clk->lap(5);  // sum of the partial results
        } else {
            out = acc;
        }
    }

    // ---- four waves, each a QUARTER of every stage vector -----------------------------------
    // (nops = 2: waves = generator | operator 1 | operator 2 | control cotangents.) The stamps of
    // the register-resident loops (tools/lindblad_stamps.py, profiles/r02_lindblad_stamps_*.json)
    // show a third of a stage going into the Runge-Kutta combinations y_i = y0 + h sum a_ij k_j:
    // every wave formed all of them, from 192 registers of k_j that only fit as AGPRs (16 copies
    // per use). Here wave w keeps component w of the C-layout registers of every k_j (one complex
    // per lane: 48 registers for the 12 of them), forms its quarter of y_i, and the four quarters
    // meet in one LDS dump that every wave reads back whole: two workgroup barriers per stage.
    __device__ __forceinline__ static void quarter_of(const Mat& m, int w, double& re, double& im) {
        re = w == 0 ? m.re[0][0][0] : w == 1 ? m.re[0][0][1] : w == 2 ? m.re[0][0][2] : m.re[0][0][3];
        im = w == 0 ? m.im[0][0][0] : w == 1 ? m.im[0][0][1] : w == 2 ? m.im[0][0][2] : m.im[0][0][3];
    }
    // the products of one right-hand side that belong to this wave -> acc
    template <bool ADJ>
    __device__ __forceinline__ void rhs_products(Mat& acc, const Mat& y, const GenLin& gen, double c,
                                                 const Mat& opr_mw) const {
        mat_zero(acc);
        if (wv == 0) {
            Mat gl, gr, gd;
            dump_load(gl, gen.d);
            dump_load(gd, gen.d + MAT);
            dump_load(gr, gen.d + 2 * (size_t)MAT);
            mat_axpy(gl, c, gd);
            mat_axpy(gr, -c, gd);
            wave_sync();
            cmat_to_lds<LNB>(gl, slot_gen.re, slot_gen.im);
            cmat_to_lds<LNB>(y, slot_y.re, slot_y.im);
            wave_sync();
            clk->lap(1);
            gemm<false>(acc, slot_gen, y);
            gemm<false>(acc, slot_y, gr);
            clk->lap(2);
        } else if (wv <= a.nops) {
            Mat t;
            mat_zero(t);
            const Slot op = slot_at(op_planar + (size_t)(wv - 1) * SLOT_BYTES);
            gemm<ADJ>(t, op, y);
            cmat_scale<LNB>(t, a.gammas[wv - 1]);
            wave_sync();
            cmat_to_lds<LNB>(t, slot_tmp.re, slot_tmp.im);
            wave_sync();
            clk->lap(1);
            gemm<false>(acc, slot_tmp, opr_mw);
            clk->lap(2);
        }
    }
    // this wave's component of sum_w parts[w] (the three waves that own products)
    __device__ __forceinline__ void sum_quarter(double& re, double& im) const {
        const int lane = lane_id();
        re = 0;
        im = 0;
#pragma unroll
        for (int w = 0; w < 3; ++w) {
            const double2 e = parts[(size_t)w * MAT + wv * 64 + lane];
            re += e.x;
            im += e.y;
        }
    }

    __device__ __forceinline__ void substep_q(const SubStep& ss, Mat& y0, double2* ystore) const {
        const int lane = lane_id();
        double2* ybuf = kdump + 3 * (size_t)MAT;  // kdump[0..2]: the linear generator
        double kr[STAGES], ki[STAGES];
        GenLin gen{kdump};
        if (wv == 0) build_linear(ss, false, gen);
        Mat opr;
        mat_zero(opr);
        if (wv >= 1 && wv <= a.nops)
            load_adjoint(opr, slot_at(op_planar + (size_t)(wv - 1) * SLOT_BYTES));
        double y0r, y0i;
        quarter_of(y0, wv, y0r, y0i);
#pragma unroll
        for (int i = 0; i < STAGES; ++i) {
            double yr = y0r, yi = y0i;
#pragma unroll
            for (int j = 0; j < i; ++j)  // kr / ki hold h k_j: the tableau entries stay literals
                if (QOCX_RK_A[i][j] != 0.0) {
                    yr = fma(QOCX_RK_A[i][j], kr[j], yr);
                    yi = fma(QOCX_RK_A[i][j], ki[j], yi);
                }
            ybuf[wv * 64 + lane] = make_double2(yr, yi);
            clk->lap(0);
            __syncthreads();
            clk->lap(4);
            Mat y, acc;
            dump_load(y, ybuf);
            if (ystore != nullptr && wv == 3) dump_store(y, ystore + (size_t)i * MAT);
            rhs_products<false>(acc, y, gen, QOCX_RK_C[i], opr);
            if (wv < 3) dump_store(acc, parts + (size_t)wv * MAT);
            clk->lap(3);
            __syncthreads();
            clk->lap(4);
            sum_quarter(kr[i], ki[i]);
            kr[i] *= ss.h;
            ki[i] *= ss.h;
            clk->lap(5);
        }
#pragma unroll
        for (int i = 0; i < STAGES; ++i)
            if (QOCX_RK_B[i] != 0.0) {
                y0r = fma(QOCX_RK_B[i], kr[i], y0r);
                y0i = fma(QOCX_RK_B[i], ki[i], y0i);
            }
        ybuf[wv * 64 + lane] = make_double2(y0r, y0i);
        __syncthreads();
        dump_load(y0, ybuf);
        __syncthreads();  // ybuf is free again
    }

    __device__ __forceinline__ void adjoint_substep_q(const SubStep& ss, const Mat& lambda,
                                                      Mat& lambda_new,
                                                      double (&ga)[QOCX_LINDBLAD_MAX_K],
                                                      double (&gb)[QOCX_LINDBLAD_MAX_K],
                                                      const double2* ystore,
                                                      double2* kbstore = nullptr) const {
        const int lane = lane_id();
        double2* ybuf = kdump + 3 * (size_t)MAT;
        double br[STAGES], bi[STAGES];  // this wave's component of every Ybar_j
        GenLin gen{kdump};
        if (wv == 0) build_linear(ss, true, gen);
        Mat opr;
        mat_zero(opr);
        if (wv >= 1 && wv <= a.nops)
            load_plain(opr, slot_at(op_planar + (size_t)(wv - 1) * SLOT_BYTES));
        double lr, li;
        quarter_of(lambda, wv, lr, li);
        double nr = lr, ni = li;  // lambda_new = lambda + sum_i Ybar_i
        const double hlr = ss.h * lr, hli = ss.h * li;  // br / bi hold h Ybar_j
#pragma unroll
        for (int ii = 0; ii < STAGES; ++ii) {
            constexpr int LAST = STAGES - 1;
            const int i = LAST - ii;
            // kbar_i = h (b_i lambda + sum_{j>i} a_ji Ybar_j)
            double kbr = 0, kbi = 0;
            if (QOCX_RK_B[LAST - ii] != 0.0) {
                kbr = QOCX_RK_B[LAST - ii] * hlr;
                kbi = QOCX_RK_B[LAST - ii] * hli;
            }
#pragma unroll
            for (int jj = 0; jj < ii; ++jj)
                if (QOCX_RK_A[LAST - jj][LAST - ii] != 0.0) {
                    kbr = fma(QOCX_RK_A[LAST - jj][LAST - ii], br[LAST - jj], kbr);
                    kbi = fma(QOCX_RK_A[LAST - jj][LAST - ii], bi[LAST - jj], kbi);
                }
            ybuf[wv * 64 + lane] = make_double2(kbr, kbi);
            clk->lap(0);
            __syncthreads();
            clk->lap(4);
            Mat kb, acc;
            dump_load(kb, ybuf);
            const double ci = QOCX_RK_C[LAST - ii];
            if (wv < 3) {
                rhs_products<true>(acc, kb, gen, ci, opr);
                dump_store(acc, parts + (size_t)wv * MAT);
                clk->lap(3);
            } else if (kbstore != nullptr) {
                // two-sided evaluation: the stage cotangent goes to HBM, lindblad_combine forms
                // the control cotangents from it and the forward pass's stage value
                dump_store(kb, kbstore + (size_t)i * MAT);
            } else {
                // control cotangent of this stage, beside the other waves' products:
                // Re tr(Z Gp_k), Z = Y kbar^H - kbar^H Y
                Mat y, kbd, z, z2;
                dump_load(y, ystore + (size_t)i * MAT);
                wave_sync();
                cmat_to_lds<LNB>(kb, slot_zk.re, slot_zk.im);
                wave_sync();
                load_adjoint(kbd, slot_zk);
                mat_zero(z2);
                gemm<true>(z2, slot_zk, y);  // kbar^H Y
                wave_sync();
                cmat_to_lds<LNB>(y, slot_zy.re, slot_zy.im);
                wave_sync();
                mat_zero(z);
                gemm<false>(z, slot_zy, kbd);  // Y kbar^H
                mat_axpy(z, -1.0, z2);
                const int K = a.K;
                for (int k = 0; k < K; ++k) {
                    Mat gt;
                    dump_load(gt, c_gpt + (size_t)k * MAT);
                    double pr = 0;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        pr += z.re[0][0][r] * gt.re[0][0][r] - z.im[0][0][r] * gt.im[0][0][r];
                    const double g = wave_sum(pr);
#pragma unroll
                    for (int kk = 0; kk < QOCX_LINDBLAD_MAX_K; ++kk)
                        if (kk == k) {
                            ga[kk] += (1.0 - ci) * g;
                            gb[kk] += ci * g;
                        }
                }
                clk->lap(6);
            }
            __syncthreads();
            clk->lap(4);
            sum_quarter(br[LAST - ii], bi[LAST - ii]);
            nr += br[LAST - ii];
            ni += bi[LAST - ii];
            br[LAST - ii] *= ss.h;
            bi[LAST - ii] *= ss.h;
            clk->lap(5);
        }
        ybuf[wv * 64 + lane] = make_double2(nr, ni);
        __syncthreads();
        dump_load(lambda_new, ybuf);
        __syncthreads();
    }

    // ---- four waves, the two-sided evaluation's stage loop (round 4) ---------------------------
    // In the two launches of the two-sided evaluation the fourth wave has no control cotangents to
    // form, and the stamps (tools/lindblad_stamps.py) show the generator wave as the critical one:
    // 700 cycles of generator build-up, then 24 MFMAs, while the operator waves wait 600 cycles and
    // the fourth 3 200. Here every wave owns 18 of the 72 MFMAs of a right-hand side:
    //   round 1   wave 0: A_L(c) y | wave 1: t_1 = gamma_1 L_1 y | wave 2: t_2 | wave 3: y A_R(c)
    //   barrier M (t_1, t_2 are in their slots)
    //   round 2   t_i L_i^H split along K: k-steps 0, 1 on the wave that formed t_i, 2, 3 on wave 0
    //             (operator 1) and wave 3 (operator 2)
    //   barrier E (the four partial sums are in `parts`)
    // and the Runge-Kutta combination needs NO barrier of its own: before E every wave writes its
    // quarter of the part of the next argument that is known already (y0 + h sum_{j<i} a_{i+1,j} k_j),
    // behind E every wave reads that dump and the four partial sums whole and finishes y_{i+1} = pre
    // + h a_{i+1,i} k_i itself. The generator image of the next stage (wave 0) and its right factor
    // (wave 3) are prepared while the products of this one are in the pipe.
    // ADJ: the same loop on kbar_i = h (b_i lambda + sum_{j>i} a_ji Ybar_j), stages 11 .. 0.
    template <int HALF>
    static __device__ __forceinline__ void gemm_half(Mat& acc, const Slot& left, const Mat& right) {
        const int q = lane_id() >> 4, c = lane_id() & 15;
        d4 t1 = {0, 0, 0, 0}, t2 = {0, 0, 0, 0}, t3 = {0, 0, 0, 0};
        double are[2], aim[2];
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            const int off = c * LG::PITCH + 4 * (2 * HALF + k2) + q;
            are[k2] = left.re[off];
            aim[k2] = left.im[off];
        }
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            const int kk = 2 * HALF + k2;
            t1 = mfma_f64(are[k2], right.re[0][0][kk], t1);
            t2 = mfma_f64(aim[k2], right.im[0][0][kk], t2);
            t3 = mfma_f64(are[k2] + aim[k2], right.re[0][0][kk] + right.im[0][0][kk], t3);
        }
        acc.re[0][0] += t1 - t2;
        acc.im[0][0] += t3 - t1 - t2;
    }
    // coefficient of the stage processed t-th in the argument of the stage processed s-th (t < s)
    static constexpr double q2_coef(bool adj, int s, int t) {
        return adj ? QOCX_RK_A[STAGES - 1 - t][STAGES - 1 - s] : QOCX_RK_A[s][t];
    }
    // `ua`, `ub`: the controls at the two ends of the sub-interval (run() takes them from the mail box);
    // `post`: what wave 0 does before the last barrier of the sub-interval (the next one's mail).
    // Returns with `result` whole in every wave and NO barrier behind it: the new density (cotangent)
    // is finished like a thirteenth argument, from the dump of its known part and the last k_i.
    template <bool ADJ, class Post>
    __device__ __forceinline__ void substep_q2(const SubStep& ss, const double (&ua)[QOCX_LINDBLAD_MAX_K],
                                               const double (&ub)[QOCX_LINDBLAD_MAX_K], const Mat& base_in,
                                               Mat& result, double2* store, Post post) const {
        constexpr int LAST = STAGES - 1;
        const int lane = lane_id();
        const Mat base = base_in;
        double2* pre = kdump + 4 * (size_t)MAT;   // two dumps, by the parity of the stage
        const Slot gen_slot[2] = {slot_gen, slot_zk};
        const int myop = (wv == 0 || wv == 1) ? 0 : 1;
        const Slot tmp = slot_at(reinterpret_cast<char*>(slot_gen.re) + (size_t)(2 + myop) * SLOT_BYTES);
        const Slot op = slot_at(op_planar + (size_t)myop * SLOT_BYTES);
        const double gamma = a.gammas[myop];
        double hr[STAGES], hi[STAGES];  // this wave's component of h k_j (ADJ: h Ybar_j), by order of processing
        // the generators of the sub-interval, left(c) = la + c ld, right(c) = ra - c ld: kdump[0..2],
        // one dump per wave, from the pass's constant dumps in kdump[6 ..] (run())
        GenLin gen{kdump};
        {
            const double2* cache = kdump + 6 * (size_t)MAT;
            const int K = a.K;
            Mat m;
            mat_zero(m);
            if (wv == 0) dump_load(m, cache);
            if (wv == 3) dump_load(m, cache + MAT);
            if (wv != 2) {
#pragma unroll
                for (int k = 0; k < QOCX_LINDBLAD_MAX_K; ++k)
                    if (k < K) {
                        Mat gk;
                        dump_load(gk, cache + (size_t)(2 + k) * MAT);
                        mat_axpy(m, wv == 0 ? ua[k] : wv == 1 ? ub[k] - ua[k] : -ua[k], gk);
                    }
                dump_store(m, gen.d + (size_t)(wv == 0 ? 0 : wv == 1 ? 1 : 2) * MAT);
            }
        }
        Mat opr;
        if (ADJ) load_plain(opr, op);
        else load_adjoint(opr, op);
        double br, bi;
        quarter_of(base, wv, br, bi);
        double nr = br, ni = bi;  // ADJ: lambda_new = lambda + sum_i Ybar_i
        Mat arg, gr;
        mat_zero(gr);
        if (ADJ) {
            mat_zero(arg);
            if (QOCX_RK_B[LAST] != 0.0) mat_axpy(arg, ss.h * QOCX_RK_B[LAST], base);
        } else {
            arg = base;
        }
        __syncthreads();  // the three generator dumps are there
        // generator factors of a stage: left -> the slot (wave 0), right in registers (wave 3)
        auto left_to_slot = [&](double c, const Slot& dst) {
            Mat gl, gd;
            dump_load(gl, gen.d);
            dump_load(gd, gen.d + MAT);
            mat_axpy(gl, c, gd);
            cmat_to_lds<LNB>(gl, dst.re, dst.im);
        };
        auto right_of = [&](double c, Mat& gr_) {
            Mat gd;
            dump_load(gr_, gen.d + 2 * (size_t)MAT);
            dump_load(gd, gen.d + MAT);
            mat_axpy(gr_, -c, gd);
        };
        if (wv == 0) left_to_slot(QOCX_RK_C[ADJ ? LAST : 0], gen_slot[0]);
        if (wv == 3) right_of(QOCX_RK_C[ADJ ? LAST : 0], gr);
        wave_sync();
        clk->lap(6);  // (stamped build: between the last stage of a sub-interval and the first of the next)
#pragma unroll
        for (int s = 0; s < STAGES; ++s) {
            const int i = ADJ ? LAST - s : s;  // the stage of the tableau
            const int sn = s < LAST ? s + 1 : s;  // (keeps the tableau subscripts below in range)
            if (wv == 3 && store != nullptr) dump_store(arg, store + (size_t)i * MAT);
            Mat acc;
            mat_zero(acc);
            // ---- round 1
            if (wv == 0) {
                clk->lap(0);
                gemm<false>(acc, gen_slot[s & 1], arg);
            } else if (wv == 3) {
                cmat_to_lds<LNB>(arg, slot_y.re, slot_y.im);
                wave_sync();
                clk->lap(1);
                gemm<false>(acc, slot_y, gr);
            } else {
                clk->lap(0);
                Mat t;
                mat_zero(t);
                gemm<ADJ>(t, op, arg);
                cmat_scale<LNB>(t, gamma);
                cmat_to_lds<LNB>(t, tmp.re, tmp.im);
            }
            // this wave's quarter of what is known of the next argument (after the last stage: of the
            // new density / cotangent)
            {
                double pr, pi;
                if (s == LAST) {
                    pr = nr;
                    pi = ni;
                    if (!ADJ) {
#pragma unroll
                        for (int t = 0; t < LAST; ++t)
                            if (QOCX_RK_B[t] != 0.0) {
                                pr = fma(QOCX_RK_B[t], hr[t], pr);
                                pi = fma(QOCX_RK_B[t], hi[t], pi);
                            }
                    }
                } else {
                    if (ADJ) {
                        pr = 0;
                        pi = 0;
                        if (QOCX_RK_B[LAST - sn] != 0.0) {
                            pr = (ss.h * QOCX_RK_B[LAST - sn]) * br;
                            pi = (ss.h * QOCX_RK_B[LAST - sn]) * bi;
                        }
                    } else {
                        pr = br;
                        pi = bi;
                    }
#pragma unroll
                    for (int t = 0; t < s; ++t)
                        if (q2_coef(ADJ, sn, t) != 0.0) {
                            pr = fma(q2_coef(ADJ, sn, t), hr[t], pr);
                            pi = fma(q2_coef(ADJ, sn, t), hi[t], pi);
                        }
                }
                pre[(size_t)((s + 1) & 1) * MAT + wv * 64 + lane] = make_double2(pr, pi);
            }
            // the next stage's generator factors, beside the products
            if (s < LAST) {
                const double cn = QOCX_RK_C[ADJ ? LAST - sn : sn];
                if (wv == 0) left_to_slot(cn, gen_slot[(s + 1) & 1]);
                if (wv == 3) right_of(cn, gr);
            }
            clk->lap(2);
            __syncthreads();  // M
            clk->lap(4);
            // ---- round 2: t_i Op_i^H along K
            if (wv == 1 || wv == 2) gemm_half<0>(acc, tmp, opr);
            else gemm_half<1>(acc, tmp, opr);
            dump_store(acc, parts + (size_t)wv * MAT);
            if (s == LAST && wv == 0) post();
            clk->lap(3);
            __syncthreads();  // E
            clk->lap(4);
            // ---- k_i whole, this wave's quarter of it, the next argument
            Mat k;
            dump_load(k, parts);
#pragma unroll
            for (int w = 1; w < 4; ++w) {
                Mat pw;
                dump_load(pw, parts + (size_t)w * MAT);
                mat_axpy(k, 1.0, pw);
            }
            double kr_, ki_;
            quarter_of(k, wv, kr_, ki_);
            if (ADJ) {
                nr += kr_;
                ni += ki_;
            }
            hr[s] = ss.h * kr_;
            hi[s] = ss.h * ki_;
            if (s < LAST) {
                dump_load(arg, pre + (size_t)((s + 1) & 1) * MAT);
                if (q2_coef(ADJ, sn, s) != 0.0) mat_axpy(arg, ss.h * q2_coef(ADJ, sn, s), k);
            } else {
                dump_load(result, pre + (size_t)((s + 1) & 1) * MAT);
                if (ADJ) mat_axpy(result, 1.0, k);
                else if (QOCX_RK_B[LAST] != 0.0) mat_axpy(result, ss.h * QOCX_RK_B[LAST], k);
            }
            clk->lap(5);
        }
    }

    // ---- the two-sided stage loop with ONE workgroup barrier per stage (chain form, CH) -------------
    // gamma L y L^H = gamma L (y L^H): the wave that owns an operator forms u = y L^H and u comes out
    // of the matrix pipe in the C layout, which IS the right-operand layout, so L u follows without a
    // trip through LDS and without the barrier substep_q2 has between its two rounds (24 MFMAs in a
    // row on the operator waves instead of 12 + 6 on every wave: a stage is bound by its barriers and
    // LDS round trips, not by the matrix pipe). Every LEFT operand is held in registers (lane (q, c):
    // element (c, 4 kk + q), i.e. the C layout of the transpose): L for the whole sub-interval, the
    // generator as la + c ld (eight fma per stage), and the argument y itself - for a Hermitian problem
    // (LindbladArgs::hermitian) its left-operand image is conj of its C-layout registers, so a stage
    // touches LDS for the exchange of the partial sums only; otherwise through the wave's own planar
    // slot. Real Lindblad operators (LindbladArgs::ops_real): 16 MFMAs per chain instead of 24. Jobs by wave:
    //   nops = 2: A_L y | chain 1 | chain 2 | y A_R            (12 | 24 | 24 | 12 MFMAs)
    //   nops = 3: A_L y + y A_R | chain 1 | chain 2 | chain 3  (24 each; Hermitian: y A_R = (A_L y)^H, a
    //             transposition through the wave's slot instead of the second product)
    //   nops = 4: A_L y + chain 1 | y A_R + chain 2 | chain 3 | chain 4  (36 | 36 | 24 | 24)
    // The Runge-Kutta bookkeeping is substep_q2's: every wave keeps a quarter of every h k_j.
    // ADJ: Gen^H, GenRight^H, L^H (y L). Returns with `result` whole in every wave, no barrier behind it.
    template <bool LEFT_ADJ>
    static __device__ __forceinline__ void left_regs(const Slot& left, double (&are)[4], double (&aim)[4]) {
        const int q = lane_id() >> 4, c = lane_id() & 15;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int off = LEFT_ADJ ? ((4 * kk + q) * LG::PITCH + c) : (c * LG::PITCH + 4 * kk + q);
            are[kk] = left.re[off];
            aim[kk] = LEFT_ADJ ? -left.im[off] : left.im[off];
        }
    }
    // acc += Left right, Left in registers (the 3M scheme of gemm())
    static __device__ __forceinline__ void gemm_r(Mat& acc, const double (&are)[4], const double (&aim)[4],
                                                  const Mat& right) {
        d4 t1 = {0, 0, 0, 0}, t2 = {0, 0, 0, 0}, t3 = {0, 0, 0, 0};
        double asum[4], bsum[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            asum[kk] = are[kk] + aim[kk];
            bsum[kk] = right.re[0][0][kk] + right.im[0][0][kk];
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            t1 = mfma_f64(are[kk], right.re[0][0][kk], t1);
            t2 = mfma_f64(aim[kk], right.im[0][0][kk], t2);
            t3 = mfma_f64(asum[kk], bsum[kk], t3);
        }
        acc.re[0][0] += t1 - t2;
        acc.im[0][0] += t3 - t1 - t2;
    }
    // One straight-line copy of the stage loop per set of jobs (JOBS: 1 = A_L y, 2 = y A_R, 4 = an
    // operator chain): a wave carries the registers of its own jobs only.
    template <bool ADJ, class Post>
    __device__ __forceinline__ void substep_chain(const SubStep& ss, const double (&ua)[QOCX_LINDBLAD_MAX_K],
                                                  const double (&ub)[QOCX_LINDBLAD_MAX_K], const Mat& base_in,
                                                  Mat& result, double2* store, Post& post) const {
        const int nops = a.nops;
        if (nops == 2) {
            if (wv == 0) chain_jobs<ADJ, 1>(ss, ua, ub, base_in, result, store, post, 0);
            else if (wv == 3) chain_jobs<ADJ, 2>(ss, ua, ub, base_in, result, store, post, 0);
            else chain_jobs<ADJ, 4>(ss, ua, ub, base_in, result, store, post, wv - 1);
        } else if (nops == 3) {
            if (wv == 0) chain_jobs<ADJ, 3>(ss, ua, ub, base_in, result, store, post, 0);
            else chain_jobs<ADJ, 4>(ss, ua, ub, base_in, result, store, post, wv - 1);
        } else {
            if (wv == 0) chain_jobs<ADJ, 5>(ss, ua, ub, base_in, result, store, post, 0);
            else if (wv == 1) chain_jobs<ADJ, 6>(ss, ua, ub, base_in, result, store, post, 1);
            else chain_jobs<ADJ, 4>(ss, ua, ub, base_in, result, store, post, wv);
        }
    }
    template <bool ADJ, int JOBS, class Post>
    __device__ __forceinline__ void chain_jobs(const SubStep& ss, const double (&ua)[QOCX_LINDBLAD_MAX_K],
                                               const double (&ub)[QOCX_LINDBLAD_MAX_K], const Mat& base_in,
                                               Mat& result, double2* store, Post& post, int chain_index) const {
        static_assert(LNB == 1, "one tile per matrix");
        constexpr int LAST = STAGES - 1;
        constexpr bool has_gl = (JOBS & 1) != 0, has_gr = (JOBS & 2) != 0;
        const int lane = lane_id();
        const int nops = a.nops;
        const bool herm = a.hermitian != 0, real_ops = a.ops_real != 0;
        const int chain = (JOBS & 4) ? chain_index : -1;
        const Mat base = base_in;
        double2* pre = kdump + 4 * (size_t)MAT;   // two dumps, by the parity of the stage
        // this wave's planar slot (slot_tmp: one per operator, that of waves 1 and 2)
        const Slot mine = wv == 0 ? slot_y : wv == 3 ? slot_zy : slot_tmp;
        const Slot op = slot_at(op_planar + (size_t)(chain >= 0 ? chain : 0) * SLOT_BYTES);
        const double gamma = a.gammas[chain >= 0 ? chain : 0];
        double hr[STAGES], hi[STAGES];  // this wave's quarter of h k_j (ADJ: h Ybar_j), by order of processing
        // the generators of the sub-interval, left(c) = la + c ld (their left-operand registers),
        // right(c) = ra - c ld (C layout), from the pass's constant dumps in kdump[6 ..] (run())
        double lar[4], lai[4], ldr[4], ldi[4];
        Mat ra, rd;
        mat_zero(ra);
        mat_zero(rd);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) lar[kk] = lai[kk] = ldr[kk] = ldi[kk] = 0;
        if constexpr (has_gr) {
            const double2* cache = kdump + 6 * (size_t)MAT;
            const int K = a.K;
            dump_load(ra, cache + MAT);
#pragma unroll
            for (int k = 0; k < QOCX_LINDBLAD_MAX_K; ++k)
                if (k < K) {
                    Mat gk;
                    dump_load(gk, cache + (size_t)(2 + k) * MAT);
                    mat_axpy(ra, -ua[k], gk);
                    mat_axpy(rd, ub[k] - ua[k], gk);
                }
        }
        if constexpr (has_gl) {
            // The left-operand registers of M are the C-layout registers of M^T, and the transposes are
            // (conjugates of) dumps that exist: forward A0L^T = conj(A0L^H), Gp_k^T; adjoint (A0L^H)^T =
            // conj(A0L), (Gp_k^H)^T = conj(Gp_k) - run() keeps them in kdump[0 .. 2] (the first two
            // controls; further ones come from their HBM images). Linear in the controls like la, ld.
            const int K = a.K;
            {
                Mat t0;
                dump_load(t0, kdump);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    lar[kk] = t0.re[0][0][kk];
                    lai[kk] = -t0.im[0][0][kk];
                }
            }
#pragma unroll
            for (int k = 0; k < QOCX_LINDBLAD_MAX_K; ++k)
                if (k < K) {
                    Mat tk;
                    dump_load(tk, k < 2 ? kdump + (size_t)(1 + k) * MAT
                                        : (ADJ ? a.gp_cimg : a.gpt_cimg) + (size_t)k * MAT);
                    const double sg = ADJ ? -1.0 : 1.0, du = ub[k] - ua[k];
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        const double tr = tk.re[0][0][kk], ti = sg * tk.im[0][0][kk];
                        lar[kk] = fma(ua[k], tr, lar[kk]);
                        lai[kk] = fma(ua[k], ti, lai[kk]);
                        ldr[kk] = fma(du, tr, ldr[kk]);
                        ldi[kk] = fma(du, ti, ldi[kk]);
                    }
                }
        }
        Mat opr;
        mat_zero(opr);
        double lr[4], li[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) lr[kk] = li[kk] = 0;
        if constexpr ((JOBS & 4) != 0) {
            if (ADJ) load_plain(opr, op);
            else load_adjoint(opr, op);
            left_regs<ADJ>(op, lr, li);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {  // gamma L (u) = (gamma L) u
                lr[kk] *= gamma;
                li[kk] *= gamma;
            }
        }
        double br, bi;
        quarter_of(base, wv, br, bi);
        double nr = br, ni = bi;  // ADJ: lambda_new = lambda + sum_i Ybar_i
        Mat arg;
        if (ADJ) {
            mat_zero(arg);
            if (QOCX_RK_B[LAST] != 0.0) mat_axpy(arg, ss.h * QOCX_RK_B[LAST], base);
        } else {
            arg = base;
        }
        clk->lap(6);  // (stamped build: between the last stage of a sub-interval and the first of the next)
#pragma unroll
        for (int s = 0; s < STAGES; ++s) {
            const int i = ADJ ? LAST - s : s;  // the stage of the tableau
            const int sn = s < LAST ? s + 1 : s;  // (keeps the tableau subscripts below in range)
            const double ci = QOCX_RK_C[i];
            if (wv == (nops == 2 ? 3 : 2) && store != nullptr) dump_store(arg, store + (size_t)i * MAT);
            Mat acc;
            mat_zero(acc);
            // the argument as a left operand
            double yr[4], yi[4];
            if constexpr ((JOBS & 6) != 0) {
                if (herm) {
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        yr[kk] = arg.re[0][0][kk];
                        yi[kk] = -arg.im[0][0][kk];
                    }
                } else {
                    cmat_to_lds<LNB>(arg, mine.re, mine.im);
                    wave_sync();
                    left_regs<false>(mine, yr, yi);
                }
            }
            clk->lap(1);
            if constexpr (has_gl) {
                double gr_[4], gi_[4];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    gr_[kk] = fma(ci, ldr[kk], lar[kk]);
                    gi_[kk] = fma(ci, ldi[kk], lai[kk]);
                }
                gemm_r(acc, gr_, gi_, arg);
            }
            if constexpr (JOBS == 3) {
                if (herm) {
                    // three operators, Hermitian problem: y A_R = (A_L y)^H - the mirror of the product this
                    // wave has just formed, through its own planar slot, instead of twelve more MFMAs on the
                    // wave that is the critical one here
                    Mat ah;
                    cmat_to_lds<LNB>(acc, mine.re, mine.im);
                    wave_sync();
                    load_adjoint(ah, mine);
                    mat_axpy(acc, 1.0, ah);
                    wave_sync();
                } else {
                    Mat gr = ra;
                    mat_axpy(gr, -ci, rd);
                    gemm_r(acc, yr, yi, gr);
                }
            } else if constexpr (has_gr) {
                Mat gr = ra;
                mat_axpy(gr, -ci, rd);
                gemm_r(acc, yr, yi, gr);
            }
            if constexpr ((JOBS & 4) != 0) {
                if (real_ops) {
                    // real L (a, a^dagger a, sigma_-, ...; host-checked): two real products per complex
                    // one instead of three, and the second one accumulates straight into acc
                    d4 ur = {0, 0, 0, 0}, ui = {0, 0, 0, 0};
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        ur = mfma_f64(yr[kk], opr.re[0][0][kk], ur);
                        ui = mfma_f64(yi[kk], opr.re[0][0][kk], ui);
                    }
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        acc.re[0][0] = mfma_f64(lr[kk], ur[kk], acc.re[0][0]);
                        acc.im[0][0] = mfma_f64(lr[kk], ui[kk], acc.im[0][0]);
                    }
                } else {
                    Mat u;
                    mat_zero(u);
                    gemm_r(u, yr, yi, opr);
                    gemm_r(acc, lr, li, u);
                }
            }
            // this wave's quarter of what is known of the next argument (after the last stage: of the
            // new density / cotangent)
            {
                double pr, pi;
                if (s == LAST) {
                    pr = nr;
                    pi = ni;
                    if (!ADJ) {
#pragma unroll
                        for (int t = 0; t < LAST; ++t)
                            if (QOCX_RK_B[t] != 0.0) {
                                pr = fma(QOCX_RK_B[t], hr[t], pr);
                                pi = fma(QOCX_RK_B[t], hi[t], pi);
                            }
                    }
                } else {
                    if (ADJ) {
                        pr = 0;
                        pi = 0;
                        if (QOCX_RK_B[LAST - sn] != 0.0) {
                            pr = (ss.h * QOCX_RK_B[LAST - sn]) * br;
                            pi = (ss.h * QOCX_RK_B[LAST - sn]) * bi;
                        }
                    } else {
                        pr = br;
                        pi = bi;
                    }
#pragma unroll
                    for (int t = 0; t < s; ++t)
                        if (q2_coef(ADJ, sn, t) != 0.0) {
                            pr = fma(q2_coef(ADJ, sn, t), hr[t], pr);
                            pi = fma(q2_coef(ADJ, sn, t), hi[t], pi);
                        }
                }
                pre[(size_t)((s + 1) & 1) * MAT + wv * 64 + lane] = make_double2(pr, pi);
            }
            clk->lap(2);
            const double2* set = parts + (size_t)(s & 1) * 4 * MAT;  // (read while the other set fills)
            dump_store(acc, parts + ((size_t)(s & 1) * 4 + wv) * MAT);
            if (s == LAST && has_gl) post();
            clk->lap(3);
            __syncthreads();
            clk->lap(4);
            // ---- k_i whole, this wave's quarter of it, the next argument
            Mat k;
            dump_load(k, set);
#pragma unroll
            for (int w = 1; w < 4; ++w) {
                Mat pw;
                dump_load(pw, set + (size_t)w * MAT);
                mat_axpy(k, 1.0, pw);
            }
            double kr_, ki_;
            quarter_of(k, wv, kr_, ki_);
            if (ADJ) {
                nr += kr_;
                ni += ki_;
            }
            hr[s] = ss.h * kr_;
            hi[s] = ss.h * ki_;
            if (s < LAST) {
                dump_load(arg, pre + (size_t)((s + 1) & 1) * MAT);
                if (q2_coef(ADJ, sn, s) != 0.0) mat_axpy(arg, ss.h * q2_coef(ADJ, sn, s), k);
            } else {
                dump_load(result, pre + (size_t)((s + 1) & 1) * MAT);
                if (ADJ) mat_axpy(result, 1.0, k);
                else if (QOCX_RK_B[LAST] != 0.0) mat_axpy(result, ss.h * QOCX_RK_B[LAST], k);
            }
            clk->lap(5);
        }
    }

    // forward sub-interval, the 12 stage derivatives in registers (192 of them at n <= 16); the
    // stage loop is unrolled and the zeros of the tableau vanish at compile time
    __device__ __forceinline__ void substep_reg(const SubStep& ss, Mat& y0, double2* ystore) const {
        Mat k[STAGES];
        GenLin gen{kdump};
        if (first()) build_linear(ss, false, gen);
        Mat opr;
        mat_zero(opr);
        if (MW && wv >= 1 && wv <= a.nops)
            load_adjoint(opr, slot_at(op_planar + (size_t)(wv - 1) * SLOT_BYTES));
#pragma unroll
        for (int i = 0; i < STAGES; ++i) {
            Mat y = y0;
#pragma unroll
            for (int j = 0; j < i; ++j)
                if (QOCX_RK_A[i][j] != 0.0) mat_axpy(y, ss.h * QOCX_RK_A[i][j], k[j]);
            if (ystore != nullptr && first()) dump_store(y, ystore + (size_t)i * MAT);
            rhs_lin<false>(k[i], y, gen, QOCX_RK_C[i], i & 1, opr);
            __builtin_amdgcn_sched_barrier(0);  // one stage at a time (register pressure)
        }
#pragma unroll
        for (int i = 0; i < STAGES; ++i)
            if (QOCX_RK_B[i] != 0.0) mat_axpy(y0, ss.h * QOCX_RK_B[i], k[i]);
    }

    // its discrete adjoint, the 12 Ybar_j in registers, the stage values Y_i from `ystore`
    __device__ __forceinline__ void adjoint_substep_reg(const SubStep& ss, const Mat& lambda,
                                                        Mat& lambda_new,
                                                        double (&ga)[QOCX_LINDBLAD_MAX_K],
                                                        double (&gb)[QOCX_LINDBLAD_MAX_K],
                                                        const double2* ystore) const {
        Mat yb[STAGES];
        GenLin gen{kdump};
        if (first()) build_linear(ss, true, gen);
        Mat opr;
        mat_zero(opr);
        if (MW && wv >= 1 && wv <= a.nops)
            load_plain(opr, slot_at(op_planar + (size_t)(wv - 1) * SLOT_BYTES));
#pragma unroll
        for (int ii = 0; ii < STAGES; ++ii) {
            constexpr int LAST = STAGES - 1;
            const int i = LAST - ii;
            // kbar_i = h (b_i lambda + sum_{j>i} a_ji Ybar_j)
            Mat kb;
            mat_zero(kb);
            if (QOCX_RK_B[LAST - ii] != 0.0) mat_axpy(kb, ss.h * QOCX_RK_B[LAST - ii], lambda);
#pragma unroll
            for (int jj = 0; jj < ii; ++jj)  // j = LAST - jj > i
                if (QOCX_RK_A[LAST - jj][LAST - ii] != 0.0)
                    mat_axpy(kb, ss.h * QOCX_RK_A[LAST - jj][LAST - ii], yb[LAST - jj]);
            const double ci = QOCX_RK_C[LAST - ii];
            rhs_lin<true>(yb[LAST - ii], kb, gen, ci, ii & 1, opr);
            mat_axpy(lambda_new, 1.0, yb[LAST - ii]);
            if (z_wave()) {
                // control cotangent of this stage: Re tr(Z Gp_k), Z = Y kbar^H - kbar^H Y
                Mat y, kbd, z, z2;
                dump_load(y, ystore + (size_t)i * MAT);
                wave_sync();
                if (MW) cmat_to_lds<LNB>(kb, slot_zk.re, slot_zk.im);  // (single wave: slot_y
                wave_sync();                                          //  still holds kbar)
                load_adjoint(kbd, slot_zk);
                mat_zero(z2);
                gemm<true>(z2, slot_zk, y);  // kbar^H Y
                wave_sync();
                cmat_to_lds<LNB>(y, slot_zy.re, slot_zy.im);
                wave_sync();
                mat_zero(z);
                gemm<false>(z, slot_zy, kbd);  // Y kbar^H
                mat_axpy(z, -1.0, z2);
                const int K = a.K;
                for (int k = 0; k < K; ++k) {
                    Mat gt;
                    dump_load(gt, c_gpt + (size_t)k * MAT);
                    double pr = 0;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        pr += z.re[0][0][r] * gt.re[0][0][r] - z.im[0][0][r] * gt.im[0][0][r];
                    const double g = wave_sum(pr);
#pragma unroll
                    for (int kk = 0; kk < QOCX_LINDBLAD_MAX_K; ++kk)
                        if (kk == k) {
                            ga[kk] += (1.0 - ci) * g;
                            gb[kk] += ci * g;
                        }
                }
                wave_sync();
                clk->lap(6);  // control cotangents of the stage
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---- forward stages -------------------------------------------------------------------
    // The stage derivatives k_j live in slot j of `kdump`. Plain runtime loops over the
    // tableau (in constant memory): the body is GEMM bound, and rolled loops keep the register
    // allocation small.
    __device__ __forceinline__ void stage_value(Mat& y, const Mat& y0, int i, double h) const {
        y = y0;
        for (int j = 0; j < i; ++j) {
            const double aij = RK_A_DEV[i * STAGES + j];
            if (aij != 0.0) {
                Mat k;
                dump_load(k, kdump + (size_t)j * MAT);
                mat_axpy(y, h * aij, k);
            }
        }
    }
    // all stage derivatives of the sub-interval starting at y0 -> kdump; optionally advance y0
    // (every wave ends with the same y0)
    __device__ __forceinline__ void substep(const SubStep& ss, int q, Mat& y0, bool advance,
                                            double2* ystore = nullptr) const {
        for (int i = 0; i < STAGES; ++i) {
            Mat y, k;
            stage_value(y, y0, i, ss.h);
            if (ystore != nullptr && first()) dump_store(y, ystore + (size_t)i * MAT);
            rhs_split<false>(k, y, ss, (size_t)q * STAGES + i, RK_C_DEV[i]);
            if (first()) dump_store(k, kdump + (size_t)i * MAT);
            block_sync();
        }
        if (advance)
            for (int i = 0; i < STAGES; ++i) {
                const double bi = RK_B_DEV[i];
                if (bi != 0.0) {
                    Mat k;
                    dump_load(k, kdump + (size_t)i * MAT);
                    mat_axpy(y0, ss.h * bi, k);
                }
            }
    }

    // ---- adjoint stages -------------------------------------------------------------------
    // Going down in i, slot i of kdump is turned from k_i into Ybar_i once Y_i has been rebuilt
    // (Y_i needs only k_j, j < i; kbar_i needs only Ybar_j, j > i). ga / gb accumulate in the
    // control-cotangent wave only.
    __device__ __forceinline__ void adjoint_substep(const SubStep& ss, int q, const Mat& y0,
                                                    const Mat& lambda, Mat& lambda_new,
                                                    double (&ga)[QOCX_LINDBLAD_MAX_K],
                                                    double (&gb)[QOCX_LINDBLAD_MAX_K],
                                                    const double2* ystore = nullptr,
                                                    double2* kbstore = nullptr) const {
        for (int i = STAGES - 1; i >= 0; --i) {
            // kbar_i = h (b_i lambda + sum_{j>i} a_ji Ybar_j)
            Mat kb;
            mat_zero(kb);
            const double bi = RK_B_DEV[i], ci = RK_C_DEV[i];
            if (bi != 0.0) mat_axpy(kb, ss.h * bi, lambda);
            for (int j = i + 1; j < STAGES; ++j) {
                const double aji = RK_A_DEV[j * STAGES + i];
                if (aji != 0.0) {
                    Mat yb;
                    dump_load(yb, kdump + (size_t)j * MAT);
                    mat_axpy(kb, ss.h * aji, yb);
                }
            }
            Mat ybar;
            rhs_split<true>(ybar, kb, ss, (size_t)q * STAGES + i, ci);
            mat_axpy(lambda_new, 1.0, ybar);
            if (z_wave() && kbstore != nullptr) {
                // two-sided evaluation: lindblad_combine forms the control cotangents
                dump_store(kb, kbstore + (size_t)i * MAT);
            } else if (z_wave()) {
                // control cotangent of this stage: Re <kbar, Gp_k Y - Y Gp_k> = Re tr(Z Gp_k),
                // Z = Y kbar^H - kbar^H Y
                Mat y, kbd, z, z2;
                if (ystore != nullptr) dump_load(y, ystore + (size_t)i * MAT);
                else stage_value(y, y0, i, ss.h);  // the stage value Y_i again
                wave_sync();
                if (MW) cmat_to_lds<LNB>(kb, slot_zk.re, slot_zk.im);  // (single wave: slot_y
                wave_sync();                                          //  still holds kbar)
                load_adjoint(kbd, slot_zk);
                mat_zero(z2);
                gemm<true>(z2, slot_zk, y);  // kbar^H Y
                wave_sync();
                cmat_to_lds<LNB>(y, slot_zy.re, slot_zy.im);
                wave_sync();
                mat_zero(z);
                gemm<false>(z, slot_zy, kbd);  // Y kbar^H
                mat_axpy(z, -1.0, z2);
                const int K = a.K;
                for (int k = 0; k < K; ++k) {
                    Mat gt;
                    if (a.gp_tab != nullptr)  // C-image of Gp_k^T
                        dump_load(gt, a.gp_tab + ((((size_t)q * STAGES + i) * K + k) * 3 + 2) * MAT);
                    else
                        dump_load(gt, c_gpt + (size_t)k * MAT);
                    double pr = 0;
#pragma unroll
                    for (int ti = 0; ti < LNB; ++ti)
#pragma unroll
                        for (int tj = 0; tj < LNB; ++tj)
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                pr += z.re[ti][tj][r] * gt.re[ti][tj][r] -
                                      z.im[ti][tj][r] * gt.im[ti][tj][r];
                    const double g = wave_sum(pr);
#pragma unroll
                    for (int kk = 0; kk < QOCX_LINDBLAD_MAX_K; ++kk)
                        if (kk == k) {
                            ga[kk] += (1.0 - ci) * g;
                            gb[kk] += ci * g;
                        }
                }
                wave_sync();
            }
            // k_i is no longer needed (Y_i is rebuilt from k_j, j < i, or read from ystore)
            if (first()) dump_store(ybar, kdump + (size_t)i * MAT);
            block_sync();  // Ybar_i visible to every wave; `parts` free again
        }
    }
};

// the kernel body: one workgroup (1 or nops + 2 wavefronts) = one seed
static __device__ __forceinline__ void run(const LindbladArgs& a, char* smem) {
    const int lane = lane_id();
    const int S = a.S, K = a.K, nops = a.nops, nsub = a.nsub;
    const int b = blockIdx.x;
    // (Q2: the wave index as a scalar the compiler knows to be uniform)
    const int wv = Q2 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : MW ? (int)(threadIdx.x >> 6) : 0,
              nwaves = CH ? chain_waves(nops) : waves(nops);
    const bool lead = !MW || wv == 0;  // the wave that owns the shared bookkeeping
    char* p = smem;
    const Slot slot_gen = slot_at(p); p += SLOT_BYTES;
    const Slot slot_y = slot_at(p); p += SLOT_BYTES;
    Slot slot_tmp, slot_zk, slot_zy;
    if (MW) {
        // one private slot per operator wave, two for the control-cotangent wave
        slot_tmp = slot_at(p + (size_t)(wv >= 1 && wv <= nops ? wv - 1 : 0) * SLOT_BYTES);
        p += (size_t)nops * SLOT_BYTES;
        slot_zk = slot_at(p); p += SLOT_BYTES;
        slot_zy = slot_at(p); p += SLOT_BYTES;
    } else {
        slot_tmp = slot_at(p); p += SLOT_BYTES;
        slot_zk = slot_y;   // rhs_split leaves kbar there
        slot_zy = slot_tmp;
    }
    char* op_planar = p; p += (size_t)nops * SLOT_BYTES;     // L_i
    double2* parts = reinterpret_cast<double2*>(p);  // MW: two sets (stage parity)
    if (MW) p += 2 * (size_t)nwaves * DUMP_BYTES;
    double2 *dens, *lam, *kdump;
    if (GS) {  // per-seed HBM scratch: S densities | S cotangents | STAGES stage derivatives
        dens = a.scratch + (size_t)b * (2 * S + STAGES) * MAT;
        lam = dens + (size_t)S * MAT;
        kdump = lam + (size_t)S * MAT;
    } else {
        dens = reinterpret_cast<double2*>(p); p += (size_t)S * DUMP_BYTES;
        lam = reinterpret_cast<double2*>(p); p += (size_t)S * DUMP_BYTES;
        kdump = reinterpret_cast<double2*>(p);
    }

    // static operators into LDS (planar: left operand directly, right operand by a
    // transposing read)
    if (lead) {
        for (int i = 0; i < nops; ++i) {
            Mat op;
            dump_load(op, a.op_cimg + (size_t)i * MAT);
            cmat_to_lds<LNB>(op, slot_at(op_planar + (size_t)i * SLOT_BYTES).re,
                             slot_at(op_planar + (size_t)i * SLOT_BYTES).im);
        }
        for (int s = 0; s < S; ++s) {
            Mat rho;
            dump_load(rho, a.rho0_cimg + (size_t)s * MAT);
            dump_store(rho, dens + (size_t)s * MAT);
        }
    }
    block_sync();

    // constant generator dumps into LDS when the host found room (a.cache_gen): every stage
    // then builds its generator without a trip to L2
    const double2 *c_a0l = a.a0l_cimg, *c_a0r = a.a0r_cimg, *c_a0ld = a.a0ld_cimg,
                  *c_a0rd = a.a0rd_cimg, *c_gp = a.gp_cimg, *c_gpd = a.gpd_cimg,
                  *c_gpt = a.gpt_cimg;
    if (MW && a.cache_gen) {
        double2* cache = kdump + (size_t)STAGES * MAT;
        const double2* src[4] = {a.a0l_cimg, a.a0r_cimg, a.a0ld_cimg, a.a0rd_cimg};
        if (lead) {
            for (int m = 0; m < 4; ++m) {
                Mat t;
                dump_load(t, src[m]);
                dump_store(t, cache + (size_t)m * MAT);
            }
            for (int k = 0; k < K; ++k) {
                Mat t;
                dump_load(t, a.gp_cimg + (size_t)k * MAT);
                dump_store(t, cache + (size_t)(4 + k) * MAT);
                dump_load(t, a.gpd_cimg + (size_t)k * MAT);
                dump_store(t, cache + (size_t)(4 + K + k) * MAT);
                dump_load(t, a.gpt_cimg + (size_t)k * MAT);
                dump_store(t, cache + (size_t)(4 + 2 * K + k) * MAT);
            }
        }
        c_a0l = cache; c_a0r = cache + MAT; c_a0ld = cache + 2 * (size_t)MAT;
        c_a0rd = cache + 3 * (size_t)MAT;
        c_gp = cache + 4 * (size_t)MAT; c_gpd = c_gp + (size_t)K * MAT; c_gpt = c_gpd + (size_t)K * MAT;
        block_sync();
    }
    if constexpr (Q2) {
        // substep_q2 leaves kdump[6 ..] unused: the constant dumps THIS pass builds its generators
        // from (forward: A0L A0R Gp_k, unit adjoint: their conjugate transposes; K <= 4,
        // launch_lindblad) go there - a trip to L2 per sub-interval otherwise
        const bool adj = a.phase == 2;
        double2* cache = kdump + 6 * (size_t)MAT;
        if (lead) {
            Mat t;
            dump_load(t, adj ? a.a0ld_cimg : a.a0l_cimg);
            dump_store(t, cache);
            dump_load(t, adj ? a.a0rd_cimg : a.a0r_cimg);
            dump_store(t, cache + MAT);
            for (int k = 0; k < K; ++k) {
                dump_load(t, (adj ? a.gpd_cimg : a.gp_cimg) + (size_t)k * MAT);
                dump_store(t, cache + (size_t)(2 + k) * MAT);
            }
            if constexpr (CH) {  // the sources of the generator's left-operand registers (chain_jobs)
                dump_load(t, adj ? a.a0l_cimg : a.a0ld_cimg);
                dump_store(t, kdump);
                for (int k = 0; k < K && k < 2; ++k) {
                    dump_load(t, (adj ? a.gp_cimg : a.gpt_cimg) + (size_t)k * MAT);
                    dump_store(t, kdump + (size_t)(1 + k) * MAT);
                }
            }
        }
        block_sync();
    }
    StampClock<STAMP> clock;
    clock.start();
    const Wave w{a, wv, nwaves, slot_gen, slot_y, slot_tmp, slot_zk, slot_zy, op_planar, parts,
                 kdump, a.controls + (size_t)b * a.nc * K,
                 c_a0l, c_a0r, c_a0ld, c_a0rd, c_gp, c_gpd, c_gpt, &clock};
    double2* ckpt_b = a.checkpoints + (size_t)b * nsub * S * MAT;

    // ---- forward ------------------------------------------------------------------------
    double cost = 0;
    // 1 / 2: the two launches of the two-sided evaluation (several waves per seed, n <= 16)
    const int phase = (MW && LNB == 1 && !GS) ? a.phase : 0;
    // (Q2) the descriptor of a sub-interval and the control knots at its ends arrive one sub-interval
    // ahead: two dependent trips to L2 per sub-interval leave the critical path. WAVE 0 fetches
    // them - lane l holds dword l of the descriptor and control value l of the 4 K in flight (three
    // registers across the stage loop), by VECTOR loads (a scalar load shares its counter with the
    // LDS traffic of the stages) - and hands the descriptor to the others through sixteen words of
    // LDS behind the barriers the sub-interval ends with anyway. Only wave 0: the wait for a load
    // issued before the loop's back edge is a wait for everything the wave has in flight, and the
    // fourth wave has 48 stores to HBM (stage values / cotangents) in flight at that point.
    static_assert(sizeof(SubStep) == 64, "sixteen dwords");
    const double* ctl_v = a.controls + (size_t)b * a.nc * K;
    auto issue_ss = [&](int q) {  // this lane's dword of substeps[q]
        const int* ptr = reinterpret_cast<const int*>(a.substeps + min(max(q, 0), nsub - 1)) + (lane & 15);
        return *ptr;
    };
    auto take_ss = [&](int dword) {  // (field by field: a copy through an array would go through scratch)
        auto i32 = [&](int l) { return __builtin_amdgcn_readlane(dword, l); };
        auto f64 = [&](int l) { return __hiloint2double(i32(l + 1), i32(l)); };
        static_assert(offsetof(SubStep, ia1) == 8 && offsetof(SubStep, ib2) == 20 && offsetof(SubStep, wa1) == 24 &&
                          offsetof(SubStep, step) == 56, "dword indices below and in issue_u");
        SubStep sd;
        sd.h = f64(0);
        sd.ia1 = i32(2); sd.ia2 = i32(3); sd.ib1 = i32(4); sd.ib2 = i32(5);
        sd.wa1 = f64(6); sd.wa2 = f64(8); sd.wb1 = f64(10); sd.wb2 = f64(12);
        sd.step = i32(14); sd.first_of_step = i32(15);
        return sd;
    };
    auto issue_u = [&](int dword) {  // lane l < 4 K: knot (l / K) of a1 a2 b1 b2, control l % K
        const int which = min(lane / K, 3), k = lane - which * K;
        const int idx = __shfl(dword, 2 + which);  // ia1 ia2 ib1 ib2 are dwords 2 .. 5 of the descriptor
        return lane < 4 * K ? ctl_v[(size_t)idx * K + k] : 0.0;
    };
    auto take_u = [&](const SubStep& sd, double cv, double (&ua)[QOCX_LINDBLAD_MAX_K],
                      double (&ub)[QOCX_LINDBLAD_MAX_K]) {
        const int lo = __double2loint(cv), hi = __double2hiint(cv);
        auto at = [&](int l) {
            return __hiloint2double(__builtin_amdgcn_readlane(hi, l), __builtin_amdgcn_readlane(lo, l));
        };
#pragma unroll
        for (int k = 0; k < QOCX_LINDBLAD_MAX_K; ++k) {
            ua[k] = 0;
            ub[k] = 0;
            if (k < K) {
                ua[k] = sd.wa1 * at(k) + sd.wa2 * at(K + k);
                ub[k] = sd.wb1 * at(2 * K + k) + sd.wb2 * at(3 * K + k);
            }
        }
    };
    // the mail box (the second set of `parts`: unused by substep_q2): sixteen dwords of descriptor,
    // then u_k at the two ends of the sub-interval
    // (chain form: both sets of `parts` are in use; kdump[3] is free)
    int* mail = reinterpret_cast<int*>(CH ? kdump + 3 * (size_t)MAT : parts + 4 * (size_t)MAT);
    double* mail_u = reinterpret_cast<double*>(mail + 16);
    double cv_nxt = 0;  // wave 0: control knots of the NEXT sub-interval (in flight)
    int sw_nxt = 0;     // wave 0: its descriptor, and `sw_after` the one after it (in flight)
    int sw_after = 0;
    // wave 0 posts the next sub-interval: descriptor and end-point controls
    auto post_mail = [&](int dword, double cv) {
        if (lane < 16) mail[lane] = dword;
        const SubStep sd = take_ss(dword);
        double ua1[QOCX_LINDBLAD_MAX_K], ub1[QOCX_LINDBLAD_MAX_K];
        take_u(sd, cv, ua1, ub1);
#pragma unroll
        for (int k = 0; k < QOCX_LINDBLAD_MAX_K; ++k)
            if (k < K && lane == 0) {
                mail_u[k] = ua1[k];
                mail_u[QOCX_LINDBLAD_MAX_K + k] = ub1[k];
            }
    };
    // the top of a sub-interval: every wave reads the mail; wave 0 sends out the next fetches
    auto read_mail = [&](double (&ua)[QOCX_LINDBLAD_MAX_K], double (&ub)[QOCX_LINDBLAD_MAX_K]) {
#pragma unroll
        for (int k = 0; k < QOCX_LINDBLAD_MAX_K; ++k) {
            ua[k] = 0;
            ub[k] = 0;
            if (k < K) {
                ua[k] = mail_u[k];
                ub[k] = mail_u[QOCX_LINDBLAD_MAX_K + k];
            }
        }
    };
    const int q2_dir = phase == 2 ? -1 : 1;
    if constexpr (Q2) {
        const int q0 = phase == 2 ? nsub - 1 : 0;
        if (wv == 0) {
            const int sw0 = issue_ss(q0);
            post_mail(sw0, issue_u(sw0));
            sw_nxt = issue_ss(q0 + q2_dir);
        }
        block_sync();
    }
    // (Q2) one density, no step costs, no densities or cotangents per step asked for: the density
    // (cotangent) stays in every wave's registers from sub-interval to sub-interval - no trip
    // through `dens` / `lam`, none of the three barriers that went with it
    const bool q2_fast = Q2 && S == 1 && !a.has_step_costs && a.step_densities == nullptr &&
                         a.inj_index == nullptr;
    Mat carried;
    mat_zero(carried);
    if (q2_fast && phase != 2) dump_load(carried, dens);
    for (int q = 0; q < (phase == 2 ? 0 : nsub); ++q) {
        SubStep ss;
        double ua[QOCX_LINDBLAD_MAX_K], ub[QOCX_LINDBLAD_MAX_K];
        if constexpr (Q2) {
            ss = take_ss(mail[lane & 15]);
            read_mail(ua, ub);
            if (wv == 0) {  // the controls of the next sub-interval and the descriptor after it set out
                cv_nxt = issue_u(sw_nxt);
                sw_after = issue_ss(q + 2);
            }
        } else {
            ss = a.substeps[q];
        }
        auto post = [&]() {
            post_mail(sw_nxt, cv_nxt);
            sw_nxt = sw_after;
        };
        (void)post;
        if (ss.first_of_step && lead) {
            // (without step costs the walk over the cost list is two dependent trips to L2 for nothing)
            if (a.has_step_costs && ss.step != 0 && (ss.step % a.cost_eval_step) == 0)
                cost += density_costs(a, true, false, dens, nullptr);
            if (a.step_densities != nullptr)
                for (int s = 0; s < S; ++s) {
                    Mat rho;
                    dump_load(rho, dens + (size_t)s * MAT);
                    dump_store(rho, a.step_densities +
                                        (((size_t)b * (a.nsteps + 1) + ss.step) * S + s) * MAT);
                }
        }
        if (q2_fast) {
            if constexpr (Q2) {
                if (lead) dump_store(carried, ckpt_b + (size_t)q * MAT);
                if constexpr (CH)
                    w.template substep_chain<false>(ss, ua, ub, carried, carried,
                                                    a.ystages + (((size_t)b * nsub + q) * STAGES) * MAT, post);
                else
                    w.template substep_q2<false>(ss, ua, ub, carried, carried,
                                                 a.ystages + (((size_t)b * nsub + q) * STAGES) * MAT, post);
            }
            continue;
        }
        for (int s = 0; s < S; ++s) {
            Mat y0;
            dump_load(y0, dens + (size_t)s * MAT);
            if (lead) dump_store(y0, ckpt_b + ((size_t)q * S + s) * MAT);
            double2* ys = a.ystages != nullptr
                              ? a.ystages + ((((size_t)b * nsub + q) * S + s) * STAGES) * MAT
                              : nullptr;
            if constexpr (Q2 && CH)
            {
                auto post_s = [&]() { if (s == S - 1) post(); };
                w.template substep_chain<false>(ss, ua, ub, y0, y0, ys, post_s);
            }
            else if constexpr (Q2)  // (launch_lindblad: phases 1 / 2, constant H0 / G_k, stage values kept)
                w.template substep_q2<false>(ss, ua, ub, y0, y0, ys, [&]() { if (s == S - 1) post(); });
            else if (QUARTER && a.a0_tab == nullptr && a.gp_tab == nullptr)
                w.substep_q(ss, y0, ys);
            else if (REG && a.a0_tab == nullptr && a.gp_tab == nullptr)
                w.substep_reg(ss, y0, ys);  // (ys == nullptr: forward only, or the adjoint recomputes)
            else
                w.substep(ss, q, y0, true, ys);
            wave_sync();
            if (lead) dump_store(y0, dens + (size_t)s * MAT);
            block_sync();
        }
    }
    if (q2_fast && phase != 2) {
        if (lead) dump_store(carried, dens);
        block_sync();
    }
    if (lead && phase != 2) {
        if (a.has_step_costs && (a.nsteps % a.cost_eval_step) == 0)
            cost += density_costs(a, true, false, dens, nullptr);
        cost += density_costs(a, false, true, dens, nullptr);
        if (lane == 0) a.cost_out[b] = cost;
        if (phase == 1) {
            // cotangent of final density s = (f zr + i f zi) T_s, f = -scale / (S n |z|),
            // z = tr(T_s^H rho_s) (density_costs): the scalars lindblad_combine applies
            const DevCost c = a.costs[0];
            for (int s = 0; s < S; ++s) {
                Mat t, rho;
                dump_load(t, a.cost_matrices + ((size_t)c.vec_offset + s) * MAT);
                dump_load(rho, dens + (size_t)s * MAT);
                double zr, zi;
                frob_inner(t, rho, zr, zi);
                const double mag = sqrt(zr * zr + zi * zi);
                const double f = mag > 0 ? -c.scale / ((double)S * a.n * mag) : 0.0;
                if (lane == 0) a.lam_scale[(size_t)b * S + s] = make_double2(f * zr, f * zi);
            }
        }
        for (int s = 0; s < S; ++s) {
            Mat rho;
            dump_load(rho, dens + (size_t)s * MAT);
            dump_store(rho, a.final_out + ((size_t)b * S + s) * MAT);
            if (a.step_densities != nullptr)
                dump_store(rho, a.step_densities +
                                    (((size_t)b * (a.nsteps + 1) + a.nsteps) * S + s) * MAT);
        }
    }
    if (!a.want_grad || phase == 1) {
        clock.finish(a.stamps, 6, wv);
        return;
    }

    // ---- discrete adjoint ----------------------------------------------------------------
    // lambda += host-supplied cotangent of the densities at system step `step`, if there is one
    auto inject = [&](int step) {
        if (a.inj_index == nullptr) return;
        const int row = a.inj_index[step];
        if (row < 0) return;
        for (int s = 0; s < S; ++s) {
            Mat l, e;
            dump_load(l, lam + (size_t)s * MAT);
            dump_load(e, a.inj_bars + (((size_t)b * a.inj_count + row) * S + s) * MAT);
            mat_axpy(l, 1.0, e);
            dump_store(l, lam + (size_t)s * MAT);
        }
        wave_sync();
    };
    if (lead && phase == 2) {  // lambda_s = the target of density s
        for (int s = 0; s < S; ++s) {
            Mat t;
            dump_load(t, a.cost_matrices + ((size_t)a.costs[0].vec_offset + s) * MAT);
            dump_store(t, lam + (size_t)s * MAT);
        }
        wave_sync();
    } else if (lead) {
        Mat zero;
        mat_zero(zero);
        for (int s = 0; s < S; ++s) dump_store(zero, lam + (size_t)s * MAT);
        wave_sync();
        (void)density_costs(a, (a.nsteps % a.cost_eval_step) == 0, true, dens, lam);
        inject(a.nsteps);
    }
    block_sync();

    if (q2_fast && phase == 2) dump_load(carried, lam);
    for (int q = nsub - 1; q >= 0; --q) {
        SubStep ss;
        double ua[QOCX_LINDBLAD_MAX_K], ub[QOCX_LINDBLAD_MAX_K];
        if constexpr (Q2) {
            ss = take_ss(mail[lane & 15]);
            read_mail(ua, ub);
            if (wv == 0) {  // the controls of the next sub-interval and the descriptor after it set out
                cv_nxt = issue_u(sw_nxt);
                sw_after = issue_ss(q - 2);
            }
        } else {
            ss = a.substeps[q];
        }
        auto post = [&]() {
            post_mail(sw_nxt, cv_nxt);
            sw_nxt = sw_after;
        };
        (void)post;
        if (q2_fast) {
            if constexpr (Q2) {
                const Mat lambda = carried;
                if constexpr (CH)
                    w.template substep_chain<true>(ss, ua, ub, lambda, carried,
                                                   a.kbstages + (((size_t)b * nsub + q) * STAGES) * MAT, post);
                else
                    w.template substep_q2<true>(ss, ua, ub, lambda, carried,
                                                a.kbstages + (((size_t)b * nsub + q) * STAGES) * MAT, post);
            }
            continue;
        }
        double ga[QOCX_LINDBLAD_MAX_K], gb[QOCX_LINDBLAD_MAX_K];
#pragma unroll
        for (int k = 0; k < QOCX_LINDBLAD_MAX_K; ++k) {
            ga[k] = 0;
            gb[k] = 0;
        }
        for (int s = 0; s < S; ++s) {
            const double2* ys = a.ystages != nullptr
                                    ? a.ystages + ((((size_t)b * nsub + q) * S + s) * STAGES) * MAT
                                    : nullptr;
            Mat y0;
            mat_zero(y0);
            if (ys == nullptr) {  // recompute the stage derivatives from the checkpoint
                dump_load(y0, ckpt_b + ((size_t)q * S + s) * MAT);
                w.substep(ss, q, y0, false);
            }
            Mat lambda, lambda_new;
            dump_load(lambda, lam + (size_t)s * MAT);
            lambda_new = lambda;
            if constexpr (Q2 && CH)
            {
                auto post_s = [&]() { if (s == S - 1) post(); };
                w.template substep_chain<true>(ss, ua, ub, lambda, lambda_new,
                                               a.kbstages + ((((size_t)b * nsub + q) * S + s) * STAGES) * MAT, post_s);
            }
            else if constexpr (Q2)
                w.template substep_q2<true>(ss, ua, ub, lambda, lambda_new,
                                            a.kbstages + ((((size_t)b * nsub + q) * S + s) * STAGES) * MAT,
                                            [&]() { if (s == S - 1) post(); });
            else if (QUARTER && ys != nullptr && a.a0_tab == nullptr && a.gp_tab == nullptr)
                w.adjoint_substep_q(ss, lambda, lambda_new, ga, gb, ys,
                                    phase == 2 ? a.kbstages + ((((size_t)b * nsub + q) * S + s) * STAGES) * MAT
                                               : nullptr);
            else if (REG && ys != nullptr && a.a0_tab == nullptr && a.gp_tab == nullptr)
                w.adjoint_substep_reg(ss, lambda, lambda_new, ga, gb, ys);
            else
                w.adjoint_substep(ss, q, y0, lambda, lambda_new, ga, gb, ys,
                                  phase == 2 ? a.kbstages + ((((size_t)b * nsub + q) * S + s) * STAGES) * MAT
                                             : nullptr);
            block_sync();  // every wave has read lambda
            if (lead) dump_store(lambda_new, lam + (size_t)s * MAT);
            block_sync();
        }
        if (w.z_wave() && lane == 0 && phase != 2)
            for (int k = 0; k < K; ++k) {
                a.gsub[(((size_t)b * nsub + q) * 2 + 0) * K + k] = ga[k];
                a.gsub[(((size_t)b * nsub + q) * 2 + 1) * K + k] = gb[k];
            }
        if (ss.first_of_step && ss.step != 0) {
            if (lead) {
                // step costs are evaluated on the densities at the START of their system step
                if ((ss.step % a.cost_eval_step) == 0 && a.has_step_costs) {
                    for (int s = 0; s < S; ++s) {
                        Mat rho;
                        dump_load(rho, ckpt_b + ((size_t)q * S + s) * MAT);
                        dump_store(rho, dens + (size_t)s * MAT);
                    }
                    wave_sync();
                    (void)density_costs(a, true, false, dens, lam);
                }
                inject(ss.step);
            }
            block_sync();
        }
    }
    clock.finish(a.stamps, 6, wv);
}
};  // struct LB

// MW4: the multi-wave form with exactly four wavefronts (nops = 2), one per SIMD: the
// quarter-split stage loops. Q2: the launches of the two-sided evaluation (substep_q2).
// CH: the chain form of that stage loop (substep_chain), four waves for nops = 2, 3, 4.
template <int LNB, bool GS, bool MW, bool MW4, bool STAMP = false, bool Q2 = false, bool CH = false>
__global__ __launch_bounds__(MW ? (MW4 ? 256 : 384) : 64) void lindblad_kernel(LindbladArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    LB<LNB, GS, MW, (LNB == 1 && !GS && !MW), STAMP, MW4, Q2, CH>::run(a, smem);
}

template <int LNB, bool GS, bool MW, bool MW4 = false, bool STAMP = false, bool Q2 = false, bool CH = false>
void launch_t(const LindbladArgs& a, int batch, hipStream_t st) {
    typedef LB<LNB, GS, MW> I;
    const int bytes = I::lds_bytes(a.S, a.nops, (MW && a.cache_gen) ? a.K : -1);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(lindblad_kernel<LNB, GS, MW, MW4, STAMP, Q2, CH>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    hipLaunchKernelGGL((lindblad_kernel<LNB, GS, MW, MW4, STAMP, Q2, CH>), dim3(batch),
                       dim3(64 * (CH ? I::chain_waves(a.nops) : I::waves(a.nops))), bytes, st, a);
}

// Two-sided evaluation, third kernel: one wave per (sub-interval, seed) contracts the forward
// stage values Y_i with the stage cotangents kbar_i of the unit adjoint into
//   gamma_k = tr(Z_i Gp_k),  Z_i = Y_i kbar_i^H - kbar_i^H Y_i   (complex: kbar_i belongs to the
// back-propagated TARGET; the true stage cotangent is c_s kbar_i with the scalar c_s phase 1 left
// in lam_scale), adds Re(conj(c_s) gamma_k) over stages and densities and splits it between the
// sub-interval's end points with weights (1 - c_i, c_i): gsub[B][nsub][2][K], as the classic
// launch writes it. Throughput work on the whole chip (nsub x B waves).
__global__ __launch_bounds__(64) void lindblad_combine_kernel(LindbladArgs a) {
    typedef LB<1, false, false> I;
    typedef I::Mat Mat;
    __shared__ __attribute__((aligned(16))) char smem[2 * I::SLOT_BYTES];
    const Slot slot_zk = I::slot_at(smem), slot_zy = I::slot_at(smem + I::SLOT_BYTES);
    const int q = blockIdx.x, b = blockIdx.y, S = a.S, K = a.K, nsub = a.nsub;
    double ga[QOCX_LINDBLAD_MAX_K], gb[QOCX_LINDBLAD_MAX_K];
#pragma unroll
    for (int k = 0; k < QOCX_LINDBLAD_MAX_K; ++k) ga[k] = gb[k] = 0;
    for (int s = 0; s < S; ++s) {
        const double2 cs = a.lam_scale[(size_t)b * S + s];
        const size_t base = ((((size_t)b * nsub + q) * S + s) * STAGES) * I::MAT;
        for (int i = 0; i < STAGES; ++i) {
            Mat y, kb, kbd, z, z2;
            I::dump_load(y, a.ystages + base + (size_t)i * I::MAT);
            I::dump_load(kb, a.kbstages + base + (size_t)i * I::MAT);
            if (a.hermitian) {
                // Hermitian Y_i, kbar_i, anti-Hermitian Gp_k (host-checked): Z = P - P^H with P = Y kbar, and
                // tr(Z Gp_k) = 2 Re tr(P Gp_k) - ONE product, its left operand the conjugate of Y's own
                // registers (the C layout of Y^T), nothing through LDS
                double yr[4], yi[4];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    yr[kk] = y.re[0][0][kk];
                    yi[kk] = -y.im[0][0][kk];
                }
                I::mat_zero(z);
                I::Wave::gemm_r(z, yr, yi, kb);
                const double ci = RK_C_DEV[i];
                for (int k = 0; k < K; ++k) {
                    Mat gt;
                    I::dump_load(gt, a.gpt_cimg + (size_t)k * I::MAT);
                    double pr = 0;
#pragma unroll
                    for (int r = 0; r < 4; ++r) pr += z.re[0][0][r] * gt.re[0][0][r] - z.im[0][0][r] * gt.im[0][0][r];
                    const double g = cs.x * (2.0 * wave_sum(pr));
#pragma unroll
                    for (int kk = 0; kk < QOCX_LINDBLAD_MAX_K; ++kk)
                        if (kk == k) {
                            ga[kk] += (1.0 - ci) * g;
                            gb[kk] += ci * g;
                        }
                }
                continue;
            }
            wave_sync();
            cmat_to_lds<1>(kb, slot_zk.re, slot_zk.im);
            wave_sync();
            I::load_adjoint(kbd, slot_zk);
            I::mat_zero(z2);
            I::template gemm<true>(z2, slot_zk, y);  // kbar^H Y
            wave_sync();
            cmat_to_lds<1>(y, slot_zy.re, slot_zy.im);
            wave_sync();
            I::mat_zero(z);
            I::template gemm<false>(z, slot_zy, kbd);  // Y kbar^H
            I::mat_axpy(z, -1.0, z2);
            const double ci = RK_C_DEV[i];
            for (int k = 0; k < K; ++k) {
                Mat gt;
                I::dump_load(gt, a.gpt_cimg + (size_t)k * I::MAT);
                double pr = 0, pi = 0;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    pr += z.re[0][0][r] * gt.re[0][0][r] - z.im[0][0][r] * gt.im[0][0][r];
                    pi += z.re[0][0][r] * gt.im[0][0][r] + z.im[0][0][r] * gt.re[0][0][r];
                }
                const double g = fma(cs.y, wave_sum(pi), cs.x * wave_sum(pr));  // Re(conj(c) gamma)
#pragma unroll
                for (int kk = 0; kk < QOCX_LINDBLAD_MAX_K; ++kk)
                    if (kk == k) {
                        ga[kk] += (1.0 - ci) * g;
                        gb[kk] += ci * g;
                    }
            }
        }
    }
    if (lane_id() == 0)
#pragma unroll
        for (int kk = 0; kk < QOCX_LINDBLAD_MAX_K; ++kk)
            if (kk < K) {
                a.gsub[(((size_t)b * nsub + q) * 2 + 0) * K + kk] = ga[kk];
                a.gsub[(((size_t)b * nsub + q) * 2 + 1) * K + kk] = gb[kk];
            }
}

}  // namespace

void launch_lindblad_combine(const LindbladArgs& a, int batch, hipStream_t st) {
    if (batch <= 0 || a.nsub <= 0) return;
    if (a.n > 16) {
        launch_lindblad4t_combine(a, batch, st);
        return;
    }
    hipLaunchKernelGGL(lindblad_combine_kernel, dim3(a.nsub, batch), dim3(64), 0, st, a);
}

void launch_lindblad(const LindbladArgs& a, int batch, hipStream_t st) {
    // the two-sided evaluation's stage loop: four waves, constant H0 / G_k, stage values kept
    const bool q2 = a.q2 && a.phase != 0 && a.a0_tab == nullptr && a.gp_tab == nullptr && a.ystages != nullptr &&
                    a.op_tab == nullptr && a.K <= 4;
    if (lindblad4t_supports(a)) launch_lindblad4t(a, batch, st);
    else if (a.n > 16) launch_t<2, true, false>(a, batch, st);
    else if (a.scratch != nullptr) launch_t<1, true, false>(a, batch, st);
#ifdef QOCX_DIAG
    else if (a.multi_wave && a.nops == 2 && a.stamps != nullptr && q2 && a.chain)
        launch_t<1, false, true, true, true, true, true>(a, batch, st);
    else if (a.multi_wave && a.nops == 2 && a.stamps != nullptr && q2)
        launch_t<1, false, true, true, true, true>(a, batch, st);  // stamped builds (qocx_diag.h)
    else if (a.multi_wave && a.nops == 2 && a.stamps != nullptr)
        launch_t<1, false, true, true, true>(a, batch, st);
#endif
    else if (a.multi_wave && a.nops >= 2 && a.nops <= 4 && q2 && a.chain)
        launch_t<1, false, true, true, false, true, true>(a, batch, st);
    else if (a.multi_wave && a.nops == 2 && q2) launch_t<1, false, true, true, false, true>(a, batch, st);
    else if (a.multi_wave && a.nops == 2) launch_t<1, false, true, true>(a, batch, st);
    else if (a.multi_wave) launch_t<1, false, true>(a, batch, st);
    else launch_t<1, false, false>(a, batch, st);
}

// LDS bytes of one seed. mode 0: one wave, everything in LDS; 1: one wave, stage derivatives /
// densities / cotangents in HBM scratch (always for n > 16); 2: nops + 2 waves per seed; 3: as 2
// with the constant generator dumps of K controls cached in LDS
int lindblad_lds_size(int n, int S, int nops, int mode, int K) {
    if (n > 16) return LB<2, true, false>::lds_bytes(S, nops);
    if (mode == 1) return LB<1, true, false>::lds_bytes(S, nops);
    if (mode == 2) return LB<1, false, true>::lds_bytes(S, nops);
    if (mode == 3) return LB<1, false, true>::lds_bytes(S, nops, K);
    return LB<1, false, false>::lds_bytes(S, nops);
}

// complex elements of per-seed HBM scratch when it is used
// (n > 16: a second set of stage dumps - the two passes of the two-sided evaluation run side by side -
// and a third for stage values recomputed from a checkpoint)
size_t lindblad_scratch_elems(int n, int S) {
    return n > 16 ? (size_t)(2 * S + 3 * STAGES) * 1024 : (size_t)(2 * S + STAGES) * 256;
}

}  // namespace qocx
