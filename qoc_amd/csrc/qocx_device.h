// qocx_device.h - argument blocks shared by the kernels (qocx_kernels.hip) and the host side of
// the C ABI (qocx_api.hip). Plain structs of device pointers and sizes.
#ifndef QOCX_DEVICE_H
#define QOCX_DEVICE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace qocx {

#define QOCX_DEV_COST_COHERENT 0
#define QOCX_DEV_COST_INCOHERENT 1
#define QOCX_DEV_COST_FORBID 2
#define QOCX_DEV_COST_TARGET_DENSITY 3
#define QOCX_DEV_COST_FORBID_DENSITY 4
#define QOCX_LINDBLAD_MAX_K 8

// Linear interpolation of the controls at the quadrature time of one propagator step
// (qoc/core/mathmethods.py:33, :54-65): u = y[i1] + ((y[i2] - y[i1]) / dx) * off.
struct StepInterp {
    int i1, i2;
    double dx, off;
};

struct DevCost {
    int kind;
    int step_cost;
    double scale;
    int vec_offset;  // first vector of this cost in the pooled cost_vectors, in units of vectors
    int cnt_offset;  // FORBID: first entry in cost_counts
};

struct FactorArgs {
    // inputs
    const double* controls;    // [B][nc][K]
    const StepInterp* interp;  // [nsteps]
    const double2* h0_cimg;    // [nt] C-images
    const double2* g_cimg;     // [nt][K] C-images
    int K, nc, nsteps, nt;
    int step0, seg_len;        // this launch covers steps [step0, step0 + seg_len) of every seed
    int hermitian;             // every h0[t], g[t][k] is exactly Hermitian
    int n;                     // Hilbert size (used by the sixteen-tile K1a: <= 48 -> nine tiles); 0: unknown
    int pade_policy = 0;       // 0: Pade order by norm (qocx_wave.h), 13: always [13/13]
    int prefer_low = 0;        // the host's bound of ||dt a||_1 is below theta_9 (1) or theta_5 (2): which
                               // paths the four-wave K1a inlines (qocx_pade4.hip)
    int skip_q = 0;            // timing experiment: the two-wave K1a does not store Q
    int herm_tiles = 1;        // four-wave K1a, Hermitian generators, orders 3 / 5: two thirds of the tiles (knob "k1a_herm4")
    // diagnostic build only (qocx_diag.h; knobs "k1a_dbg", "k1a_stamps"): bits 0-1 s_setprio level of
    // the fused factorisation, bit 2 no factorisation at all (garbage), and the cycle sums of the
    // stamped build, [2 waves][8]
    int dbg = 0;
    unsigned long long* stamps = nullptr;
    int lu_mfma = 0;           // fused K1b with its Schur updates on the matrix cores (qocx_lu4.h)
    int lu_dpp = 0;            // fused K1b on the vector unit for provably diagonal pivots (qocx_lu5.h)
    int three_wave = 0;        // orders 3 / 5, Hermitian, step table: one tile per wave on three waves (qocx_pade3.hip)
    int four_steps = 0;     // three-wave K1a: the second halves of the factorisations four to a wave in a kernel of their own (knob "k1a_four"; 1: launch_pq launches it, 2: the caller)
    int pack8 = 0;          // n <= 8, one-wave K1a: two consecutive steps of a seed as the diagonal blocks of ONE 16 x 16 tile (pade_pq8_kernel)
    int gen_share = 2;      // three-wave K1a: which tiles of the generator come from the LDS slot instead of the images (knob "k1a_share")
    // Step table (round 4, launch_step_table): `controls` is [B][nsteps][K] = the interpolated
    // controls u_k(t_mid) of every step (interp unused), and s_arr already holds every step's Pade
    // order and squaring count, taken from the bound dt (||H0||_1 + sum |u_k| ||G_k||_1): the kernel
    // neither interpolates nor forms a norm.
    int direct = 0;
    int* lu_fallbacks = nullptr;  // LuArgs::fallbacks of the fused factorisation
    // two-wave K1a (17 <= n <= 32) with K1b fused in: P stays in LDS, wave 0 factors it, only the
    // factors (and 1/U_kk, the permutation) go to HBM; launch_lu is then not called
    int fuse_lu = 0;
    double2* dinv = nullptr;
    int* perm = nullptr;
    int* iperm = nullptr;
    double dt;
    // outputs, indexed by (b * nsteps + step); column-major NP x NP images
    double2* q_img;
    double2* lu_img;  // receives P; lu_kernel factors it in place
    int* s_arr;
    int* status;
};

// u_k(t_mid) of every (seed, step) by the reference's interpolation formula, and the step's entry
// of s_arr (Pade order, squarings: qocx_wave.h) from the norm bound of that step's generator.
struct StepTableArgs {
    const double* controls;    // [B][nc][K]
    const StepInterp* interp;  // [nsteps]
    int K, nc, nsteps, batch;
    double dt;                 // the bound is |dt| (h0_norm + sum_k |u_k| g_norm[k])
    double h0_norm;            // max over time of ||H0(t)||_1
    const double* g_norm;      // [K] max over time of ||G_k(t)||_1 (device memory)
    int pade_policy;           // 0: order by norm, 13: always [13/13]
    int sq_max = 30;           // the squaring count the host sized the sub-step slots for
    int order_max = 13;        // 5: the three-wave K1a of qocx_pade3.hip runs the evaluation (host bound below theta_5)
    double* ustep;             // out: [B][nsteps][K]
    int* s_arr;                // out: [B][nsteps]
    int* status;               // bit 1: non-finite controls
};
void launch_step_table(const StepTableArgs& a, hipStream_t st);

struct LuArgs {
    double2* lu_img;  // in: P, out: L\\U of the row-permuted P (column-major)
    double2* dinv;    // [NP] 1/U_kk
    int* perm;        // [NP] perm[pos] = original row
    int* iperm;       // [NP] iperm[row] = pos
    int* status;
    int nsteps, step0, seg_len;  // work item w -> matrix (w / seg_len) * nsteps + step0 + w % seg_len
    int n;                       // Hilbert size (sixteen-tile K1b: <= 48 -> 48 elimination steps); 0: unknown
    int dbg = 0;                 // timing experiment (dbg_skip bit 4): loads and stores only
    int inverse = 0;             // lu_img receives P^-1 instead of the factors (n <= 32; qocx_lu.h inv_body)
    int all_dominant = 0;        // every matrix of the launch is diagonally dominant by the margin of qocx_lu5.h
    int pack8 = 0;         // inv16: the image of every EVEN step of the launch holds two steps (FactorArgs::pack8); the inverses go out unpacked
                                 // (the host's bound of the step norm says so): n <= 16 inverses four to a wave
    // 33 <= n <= 64, round 4: [matrices] flags of the MFMA factorisation (qocx_lu4m.hip) - 1: its
    // diagonal-pivot attempt was abandoned, lu4_kernel factors the matrix; nullptr: lu4_kernel factors
    // every matrix
    int* redo = nullptr;
    // how many matrices left the diagonal-pivot MFMA factorisation for the general one (qocx_lu4.h,
    // qocx_lu4m.hip); a device counter the host zeroes per evaluation (qocx_lu_fallbacks), or nullptr
    int* fallbacks = nullptr;
};

struct SweepArgs {
    const double2* q_img;
    const double2* lu_img;
    const double2* dinv;
    const int* perm;
    const int* iperm;
    const int* s_arr;
    const double2* psi0;  // [S][NP]
    int S, nsteps, cost_eval_step, want_grad, has_step_costs;
    int n = 0;  // Hilbert size (sixteen-tile sweep: <= 48 -> nine-tile images in LDS); 0: unknown
    // Time segmentation: one launch runs the forward sweep over steps [j_begin, j_end) (phase
    // bit 0) and / or the adjoint sweep from j_end back to j_begin (phase bit 1); state is carried
    // between launches in states/offs/cost_out (forward) and lam_buf (adjoint).
    int phase, j_begin, j_end;
    int unit_adjoint = 0; // separable final cost: the adjoint runs on lam = targets (qocx_sweep_common.h)
    double2* lam_scale = nullptr;   // [B][S] scalars c_s, written at the end of the forward sweep, read by K3
    int* offs_x = nullptr;  // unit adjoint: [B][nsteps+1] first xs slot of each step, written by the ADJOINT
                          // sweep (it may run before the forward sweep has numbered the sub-steps)
    int batch = 0;        // seeds of the launch (set by the launcher of the two-seeds-per-workgroup form)
    int onebuf = 0;       // one state: one operand set in LDS; 1: one seed per workgroup, 2: two
    int one_state = 0;    // one state, n <= 32: the dedicated kernel of qocx_sweep1.hip (knob "sweep_one")
    int ring2 = 0;        // sweep1, n > 16: two operand sets in LDS, every fetch a whole step ahead (70 KiB per seed)
    const double2* qt_img = nullptr;  // umode: U^T images (launch_umul) - the adjoint sweep copies them straight instead of gathering
    int umode = 0;        // sweepi: q_img holds the propagator U = P^-1 Q (launch_umul): ONE product per sub-step;
                          // the adjoint leaves lambda' in `xs` and K3 forms x = P^-H lambda' (KrylovArgs::umode)
    int loader;           // 1: a dedicated fetch wave per seed issues the LDS-DMA
    int dbg;              // sweep3 timing diagnostics (results are garbage): bit 0 no inversion,
                          // bit 1 no solves, bit 2 no LU fetch, bit 3 no Q fetch, bit 4 no Q touch
    unsigned long long* stamps;  // sweep3 diagnostic build: [B][4 roles][8] cycle sums, or nullptr
    double2* lam_buf;     // [B][S][NP]
    // Externally supplied state cotangents (user Cost plugins whose derivative the host
    // provides): inj_index[step] = row of inj_bars or -1; added to lambda at system step `step`
    // (the states before evolving from it; step N-1 = the final states).
    const int* inj_index;     // [nsteps + 1] or nullptr
    const double2* inj_bars;  // [B][inj_count][S][NP]
    int inj_count;
    size_t slot_cap;      // sub-step slots per seed
    // costs
    int cost_count;
    const DevCost* costs;
    const double2* cost_vectors;  // pooled, padded to NP
    const int* cost_counts;
    // outputs
    double2* states;       // [B][slot_cap][S][NP]  state before sub-step t
    double2* xs;           // [B][slot_cap][S][NP]  x = P^-H lambda' of sub-step t
    int* offs;             // [B][nsteps+1] first sub-step slot of each step
    double* cost_out;      // [B]
    double2* final_out;    // [B][S][NP]
    double2* step_states;  // [B][nsteps+1][S][NP] or nullptr
    int* status;
};

struct KrylovArgs {
    int umode = 0;                       // `xs` holds lambda' (SweepArgs::umode): x = P^-H lambda' is formed here from pinv_img
    const double2* pinv_img = nullptr;   // [B][nsteps] column-major P^-1 images (the LU buffer in inverse mode)
    int lds_pad = 0;  // extra dynamic LDS per workgroup (bytes): fewer K3 waves per CU beside the tail sweeps (knob "k3_lds_pad")
    const double* controls;
    const StepInterp* interp;
    const double2* h0_rimg;  // column-major h0
    const double2* h0_timg;  // column-major h0^T
    const double2* g_rimg;
    const double2* g_timg;
    int K, nc, nsteps, nt, S;
    int direct = 0;  // controls is the step table [B][nsteps][K] (FactorArgs::direct)
    int n = 0;  // Hilbert size (four-wave K3: <= 48 -> the zero pad columns are skipped); 0: unknown
    int step0;  // grid.x covers steps [step0, step0 + gridDim.x)
    int skew;   // 1: every H0(t), G_k(t) is exactly Hermitian (a^H = -a)
    double dt;
    const int* s_arr;
    const int* offs;
    const double2* states;
    const double2* xs;
    size_t slot_cap;
    double* gstep;  // [B][nsteps][K]
    // unit adjoint (qocx_sweep_common.h): x sits at slots of its own (offs_x: first xs slot of each
    // step) and is the back-propagated TARGET; gstep then receives the complex number
    // gamma = sum conj(abar_1) E_k per (step, k) - [B][nsteps][K][2] - and the scatter kernel forms
    // Re(conj(c) gamma) with the cost's scalar c. nullptr: x is the true cotangent, gstep is real.
    const int* offs_x = nullptr;
    // Magnus M4/M6: the generator is read from m_rm (row-major padded NP x NP, unscaled) and the
    // cotangent of M is written to mbar_rm instead of the contraction with G_k. nullptr for M2.
    const double2* m_rm;
    double2* mbar_rm;
};

// Magnus M4 / M6 generator kernels (qocx_magnus.hip)
struct MagnusArgs {
    const double* controls;    // [B][nc][K]
    const StepInterp* interp;  // [nsteps * nodes]
    const double2* h0_cimg;    // [nt] C-images, nt = 1 or nsteps * nodes
    const double2* g_cimg;     // [nt][K]
    int K, nc, nsteps, nt, nodes;
    int step0, seg_len;        // work item w -> (seed w / seg_len, step step0 + w % seg_len)
    int skew;                  // every H0(t), G_k(t) Hermitian: one product per commutator
    double dt;
    double2* m_rm;             // fwd out: [B][nsteps] row-major padded generators
    const double2* mbar_rm;    // vjp in : cotangents of the generators
    double* gstep;             // vjp out: [B][nsteps * nodes][K]
    double2* scratch;          // [blocks][11] lane-linear matrix dumps
    size_t total;              // B * seg_len work items
    int n = 0;                 // Hilbert size (qocx_magnus4w.hip: 33..48 -> the three-wave form)
};

// Magnus M4 with time-independent H0, G_k ("commutator-free" form, qocx_magnus.hip): the step
// generator is LINEAR in effective controls with constant matrices,
//   m4 = -i dt (H0 + sum_k v_k G_k + sum_k w_k A_k + sum_{k<l} z_kl B_kl),
//   A_k = -i [G_k, H0], B_kl = -i [G_k, G_l],
//   v_k = (u1_k + u2_k) / 2, w_k = F0 dt (u2_k - u1_k), z_kl = F0 dt (u2_k u1_l - u2_l u1_k)
// with u1, u2 the controls at the two quadrature nodes, so the M2 kernels run it unchanged on
// Ke = 2 K + K (K - 1) / 2 effective controls given per step.
#define QOCX_M4LIN_MAX_K 8
struct M4LinArgs {
    const double* controls;    // [B][nc][K]
    const StepInterp* interp;  // [nsteps * 2]
    int K, Ke, nc, nsteps, S;
    double f0dt;               // F0 * dt
    double* veff;              // controls kernel out: [B][nsteps][Ke]
    const double* gstep;       // chain kernel in: [B][nsteps][Ke] (x 2: complex, unit adjoint)
    const double2* lam_scale;  // unit adjoint: [B][S] (see ScatterArgs), or nullptr
    double* gnode;             // chain kernel out: [B][nsteps * 2][K]
    size_t total;              // B * nsteps
};

struct ScatterArgs {
    const double* gstep;
    const int* row_ptr;   // [nc+1]
    const int* col_step;  // [nnz]
    const double* weight; // [nnz]
    double* grads;        // [B][nc][K]
    int B, nc, K, nsteps;
    const double2* lam_scale = nullptr;  // unit adjoint: [B][S] (entry of state 0 is used), gstep holds
    int S = 1;                           // complex numbers (see KrylovArgs); nullptr: gstep is real
};

// One sub-interval of the fixed-step Lindblad integrator: [t_a, t_b] inside system step `step`,
// never straddling a control knot; u(t_a), u(t_b) by linear interpolation on the control grid.
struct SubStep {
    double h;
    int ia1, ia2, ib1, ib2;
    double wa1, wa2, wb1, wb2;
    int step;
    int first_of_step;
};

struct LindbladArgs {
    const double* controls;    // [B][nc][K]
    const SubStep* substeps;   // [nsub]
    const double2* a0l_cimg;   // C-dumps (256 complex): A0L, A0R and their conjugate transposes
    const double2* a0r_cimg;
    const double2* a0ld_cimg;
    const double2* a0rd_cimg;
    const double2* gp_cimg;    // [K] Gp_k = -i G_k
    const double2* gpd_cimg;   // [K] Gp_k^H
    const double2* gpt_cimg;   // [K] Gp_k^T
    const double2* a0_tab;     // time-dependent H: [nsub * 12][4] dumps A0L, A0R, A0L^H, A0R^H at the
                               // stage times, or nullptr
    const double2* gp_tab;     // time-dependent G: [nsub * 12][K][3] dumps Gp, Gp^H, Gp^T, or nullptr
    const double2* op_tab;     // time-dependent lindblad_data: [nsub * 12][nops] dumps of L_i at the
                               // stage times (+ gamma_tab [nsub * 12][nops]), or nullptr
    const double* gamma_tab;
    const double2* op_cimg;    // [nops] L_i
    const double* gammas;      // [nops]
    const double2* rho0_cimg;  // [S]
    int n, S, K, nc, nops, nsub, nsteps, cost_eval_step, want_grad, has_step_costs;
    int cost_count;
    const DevCost* costs;
    const double2* cost_matrices;  // pooled C-dumps
    const int* cost_counts;
    double2* checkpoints;      // [B][nsub][S] C-dumps: densities at the start of each sub-interval
    int multi_wave;            // 1: nops + 2 wavefronts per seed (n <= 16, everything in LDS)
    int cache_gen;             // multi_wave: constant generator dumps copied to LDS
    double2* scratch;          // [B][2 S + 12] (n > 16: [B][2 S + 36]) dumps when densities / cotangents / stage
                               // derivatives do not live in LDS (always for n > 16), else nullptr
    double2* ystages;          // [B][nsub][S][12] C-dumps of the stage values, or nullptr: the
                               // adjoint then recomputes them from the checkpoints
    double* gsub;              // [B][nsub][2][K] control cotangents at t_a / t_b
    double* cost_out;          // [B]
    double2* final_out;        // [B][S] C-dumps
    double2* step_densities;   // [B][nsteps+1][S] C-dumps or nullptr
    const int* inj_index;      // host-supplied density cotangents (see SweepArgs): [nsteps + 1]
    const double2* inj_bars;   // [B][inj_count][S] C-dumps
    int inj_count;
    unsigned long long* stamps;  // diagnostic build: [B][6 waves][8] cycle sums, or nullptr
    // Two-sided evaluation (several waves per seed, ONE final TargetDensityInfidelity): the
    // cotangent of a final density is a scalar times its target, and the adjoint of the discrete
    // scheme is linear in it, so it runs on the TARGET beside the forward pass (phase 2 beside
    // phase 1, two launches on two streams) and stores its stage cotangents kbar_i; a third kernel
    // (lindblad_combine) contracts them with the forward stage values and the scalars into the
    // control cotangents. phase 0: the classic forward-then-adjoint launch.
    int phase = 0;
    int hermitian = 0;             // every density and cotangent is Hermitian, A_R = A_L^H, Gp_k^H = -Gp_k (host-checked)
    int tile4 = 1;                 // 17 <= n <= 32: the tile-per-wave kernel (qocx_lindblad4t.hip) where it applies
    int q2 = 0;                    // phases 1 / 2, four waves: the stage loop with 18 MFMAs per wave (substep_q2)
    int chain = 0;                 // with q2, 2 <= nops <= 4: the stage loop with one barrier per stage (substep_chain)
    int ops_real = 0;              // every Lindblad operator has a zero imaginary part (host-checked): chain form only
    double2* kbstages = nullptr;   // [B][nsub][S][12] C-dumps of kbar_i (phase 2 out, combine in)
    double2* lam_scale = nullptr;  // [B][S]: phase 1 out
};

// ---- Hilbert sizes above 64 (qocx_general.hip): row-major padded np x np matrices in HBM, np = 16 ceil(n / 16) ----
struct GeneralArgs {  // K1a + K1b: one work item per (seed, step), `total` = seeds * nsteps
    int np, K, nc, nsteps, nt;
    double dt;
    const double* controls;    // [B][nc][K]
    const StepInterp* interp;  // [nsteps]
    const double2* h0_rm;      // [nt] row-major
    const double2* g_rm;       // [nt][K]
    const double2* gen_rm;     // explicit mode: [B][nsteps] generators sampled by the host, else nullptr
    int pade_policy, sq_max;
    double2* q_img;            // out: Q, row-major
    double2* pinv_img;         // out: P^-1, row-major
    int* s_arr;
    int* status;
    double2* scratch;          // [blocks][7] matrices
    size_t total;
};
struct GeneralSweepArgs {  // K2: one workgroup per seed
    int np, S, nsteps, cost_eval_step, has_step_costs, phase;  // phase bit 0 forward, bit 1 adjoint
    const double2* q_img;
    const double2* pinv_img;
    const int* s_arr;
    const double2* psi0;       // [S][np]
    size_t slot_cap;
    double2* states;           // [B][slot_cap][S][np]
    double2* xs;
    int* offs;                 // [B][nsteps + 1]
    double2* lam_buf;          // [B][S][np]
    int cost_count;
    const DevCost* costs;
    const double2* cost_vectors;
    const int* cost_counts;
    const int* inj_index;
    const double2* inj_bars;
    int inj_count;
    double* cost_out;
    double2* final_out;
    double2* step_states;
    int* status;
};
struct GeneralKrylovArgs {  // K3: one work item per (seed, step)
    int np, S, K, nc, nsteps, nt;
    double dt;
    const double* controls;
    const StepInterp* interp;
    const double2* h0_rm;
    const double2* g_rm;
    const double2* gen_rm;     // explicit mode (then mbar_rm receives the generator cotangents)
    double2* mbar_rm;
    const int* s_arr;
    const int* offs;
    const double2* states;
    const double2* xs;
    size_t slot_cap;
    double* gstep;             // [B][nsteps][K]
    double2* scratch;          // [blocks][3] matrices
    size_t total;
    int skew = 0;              // the generator is skew-Hermitian bit for bit (Hermitian H): both chains run on a^H, one pass each
};
int general_factor_lds(int np);
int general_sweep_lds(int np);
int general_krylov_lds(int np);
int launch_general_factor(const GeneralArgs& a, int blocks, hipStream_t st);
void launch_general_sweep(const GeneralSweepArgs& a, int batch, hipStream_t st);
int launch_general_krylov(const GeneralKrylovArgs& a, int blocks, hipStream_t st);

// Multi-start driver on the device (qocx_optim.hip)
struct OptimArgs {
    int kind;  // 0 SGD, 1 Adam
    double* params;              // [B][per_seed] = the resident controls
    const double* grads;         // [B][per_seed]
    double* moment;              // Adam: [B][per_seed]
    double* square_moment;
    const unsigned char* update; // [B]: seeds that take the step
    size_t per_seed;
    double learning_rate, beta_1, beta_2, one_m_b1, one_m_b2, epsilon, corr_1, corr_2, clip;
    int apply_clip;
};
void launch_clip_controls(double* controls, size_t total, int k, const double* max_norms, hipStream_t st);
void launch_keep_best(const double* controls, double* best_controls, size_t per_seed,
                      const double2* final_states, double2* best_final, size_t final_per_seed,
                      const unsigned char* improved, int batch, hipStream_t st);
void launch_optimizer_update(const OptimArgs& a, int batch, hipStream_t st);

void launch_lindblad(const LindbladArgs& a, int batch, hipStream_t st);
void launch_lindblad_combine(const LindbladArgs& a, int batch, hipStream_t st);
bool lindblad4t_supports(const LindbladArgs& a);
void launch_lindblad4t(const LindbladArgs& a, int batch, hipStream_t st);
void launch_lindblad4t_combine(const LindbladArgs& a, int batch, hipStream_t st);
int lindblad_lds_size(int n, int S, int nops, int mode, int K);
size_t lindblad_scratch_elems(int n, int S);

void launch_pq(int nb, const FactorArgs& a, int nsteps, int batch, hipStream_t st);
void launch_pq_explicit(int nb, const double2* a_in, int n, const FactorArgs& a, int count,
                        hipStream_t st);
// two-wave K1a for nb == 2 (qocx_pade2.hip)
void launch_pq2(const FactorArgs& a, int nsteps, int batch, hipStream_t st);
bool pq3_supports(const FactorArgs& a);
void launch_pq3(const FactorArgs& a, int nsteps, int batch, hipStream_t st);
// q_img := P^-1 Q of `count` steps (work item w -> seed w / seg_len, step step0 + w % seg_len), n <= 32
// (qt_img: where the transposed images go, for the adjoint sweep)
void launch_umul(int nb, const LuArgs& a, double2* q_img, double2* qt_img, size_t count, hipStream_t st);
bool pq3_parks(const FactorArgs& a, int nsteps);
void launch_pq3_second(const FactorArgs& a, int nsteps, int batch, hipStream_t st);
// true: launch_pq left the second halves of the factorisations to launch_pq3_second (FactorArgs::four_steps == 2)
bool pq_second_pending(int nb, const FactorArgs& a, int nsteps);
void launch_pq2_explicit(const double2* a_in, int n, const FactorArgs& a, int count, hipStream_t st);
// 33 <= n <= 64: four-wave workgroups (qocx_pade4.hip)
void launch_pq4(const FactorArgs& a, int nsteps, int batch, hipStream_t st);
void launch_pq4_explicit(const double2* a_in, int n, const FactorArgs& a, int count, hipStream_t st);
// ... and the four-wave K1b / K3 of qocx_big.hip
void launch_lu4(const LuArgs& a, size_t count, hipStream_t st);
void launch_lu4m(const LuArgs& a, size_t count, int* redo, hipStream_t st);  // qocx_lu4m.hip
void launch_krylov4(const KrylovArgs& a, int nsteps, int batch, hipStream_t st);
void launch_lu(int nb, const LuArgs& a, size_t count, hipStream_t st);
void launch_sweep(int nb, const SweepArgs& a, int batch, hipStream_t st);
int sweep_lds_bytes(int nb, int S);
// blocked-inverse sweep (qocx_sweep3.hip): three wavefronts per seed
void launch_sweep3(int nb, const SweepArgs& a, int batch, hipStream_t st);
// dense-state sweep (qocx_sweepd.hip): the states of a seed as GEMM columns; lu_img holds P^-1
bool sweepd_supports(int nb, int S);
void launch_sweepd(const SweepArgs& a, int batch, hipStream_t st);
void launch_krylovd(const KrylovArgs& a, int nsteps, int batch, hipStream_t st);  // K3 on the matrix cores
// inverse-image sweep of latency mode (qocx_sweepi.hip): two matrix-vector products per sub-step
bool sweepi_supports(int nb, int S);
bool sweep1_supports(int nb, int S);
void launch_sweep1(int nb, const SweepArgs& a, int batch, int pack, hipStream_t st);
void launch_sweepi(int nb, const SweepArgs& a, int batch, hipStream_t st);
int sweep3_max_states(int nb);
void launch_krylov(int nb, const KrylovArgs& a, int nsteps, int batch, hipStream_t st);
void launch_scatter(const ScatterArgs& a, hipStream_t st);
// out[0] = sum_b cost[b]; out[1 + j] = sum_b grads[b][j], j < per_seed (seeds in index order)
void launch_reduce_results(const double* cost, const double* grads, int batch, int per_seed,
                           double* out, hipStream_t st);
void launch_mfma_peak(double* out, int blocks, int iters, hipStream_t st);
#ifdef QOCX_DIAG
void launch_pipe_mix(double* out, int blocks, int iters, int mode, hipStream_t st);
#endif
void launch_magnus_fwd(int nb, const MagnusArgs& a, int blocks, hipStream_t st);
void launch_magnus_vjp(int nb, const MagnusArgs& a, int blocks, hipStream_t st);
size_t magnus_scratch_elems(int nb, int blocks);
// multi-wave, LDS-resident forms for 17 <= n <= 48 (qocx_magnus4w.hip); grid = (a.seg_len, batch)
bool magnus4w_supports(int nb, int K, int n);
void launch_magnus4w_fwd(const MagnusArgs& a, int batch, hipStream_t st);
void launch_magnus4w_vjp(const MagnusArgs& a, int batch, hipStream_t st);
void launch_m4lin_controls(const M4LinArgs& a, hipStream_t st);
void launch_m4lin_chain(const M4LinArgs& a, hipStream_t st);
void launch_selftest(double* out, hipStream_t st);

}  // namespace qocx

#endif
