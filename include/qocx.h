/*
 * qocx.h - C ABI of the MI355X GRAPE propagation engine (libqocx.so).
 *
 * The reference (SchusterLab/qoc) is pure Python and has no FFI; the "interface" this
 * library replaces is the Python call pair
 *
 *     error        = _evaluate_schroedinger_discrete(controls, pstate, reporter)
 *                                   qoc/core/schroedingerdiscrete.py:356-438
 *     error, grads = ans_jacobian(_evaluate_schroedinger_discrete, 0)(controls, pstate, reporter)
 *                                   qoc/core/schroedingerdiscrete.py:318-319
 *                                   qoc/standard/utils/autogradutil.py:10-31
 *
 * i.e. "controls in -> total cost, d cost / d controls and final states out", batched here
 * over B independent control arrays (seeds).  Each entry point below cites the reference
 * code whose work it performs.  INTEGRATION.md shows the ctypes stub a maintainer of the
 * reference would add at those call sites.
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error; qocx_last_error() gives the text;
 *   - the caller owns all host buffers (C-contiguous; complex = interleaved re,im float64);
 *   - the library owns all device memory inside the opaque context;
 *   - calls are blocking unless named *_async; one context per (host thread, device);
 *   - no global mutable state besides the per-thread error string.
 */
#ifndef QOCX_H
#define QOCX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct qocx_ctx qocx_ctx;

#define QOCX_OK 0
#define QOCX_ERR_ARG -1       /* invalid argument / unsupported size        */
#define QOCX_ERR_HIP -2       /* HIP runtime error                          */
#define QOCX_ERR_STATE -3     /* call sequence error (no problem set, ...)  */
#define QOCX_ERR_SINGULAR -4  /* Pade denominator numerically singular (numpy.linalg.LinAlgError
                                 in the reference: expm.py:246)             */
#define QOCX_ERR_CAPACITY -5  /* squaring sub-step capacity exceeded        */
#define QOCX_ERR_RCCL -6      /* RCCL error / librccl not loadable          */

/* qoc.models.MagnusPolicy (qoc/models/magnuspolicy.py:8-26) */
#define QOCX_MAGNUS_M2 2
#define QOCX_MAGNUS_M4 4
#define QOCX_MAGNUS_M6 6

/* State-cost kinds evaluated on the device (qoc/standard/costs/). */
#define QOCX_COST_TARGET_COHERENT 0   /* TargetStateInfidelity[Time], neglect_relative_pahse=False:
                                         scale * (1 - |sum_s <t_s|psi_s>|^2 / S^2)
                                         targetstateinfidelity.py:52-56                            */
#define QOCX_COST_TARGET_INCOHERENT 1 /* ... neglect_relative_pahse=True:
                                         scale * (1 - sum_s |<t_s|psi_s>|^2 / S)   :58-61          */
#define QOCX_COST_FORBID 2            /* ForbidStates: scale * sum_s (1/F_s) sum_f |<f_sf|psi_s>|^2
                                         forbidstates.py:64-81 (scale = multiplier / (E * S))      */

#define QOCX_COST_TARGET_DENSITY 3    /* TargetDensityInfidelity[Time]:
                                         scale * (1 - sum_s |tr(T_s^H rho_s)| / (S n))
                                         targetdensityinfidelity.py:41-69; vectors = [S][n][n]      */
#define QOCX_COST_FORBID_DENSITY 4    /* ForbidDensities: scale * sum_s (1/F_s) sum_f |tr(F_sf^H rho_s)/n|^2
                                         forbiddensities.py:53-85; vectors = [sum_s F_s][n][n]       */

typedef struct qocx_cost_desc {
    int32_t kind;          /* QOCX_COST_*                                                         */
    int32_t step_cost;     /* 1: evaluated at system steps j*cost_eval_step, j>=1, before evolving
                              from that step (schroedingerdiscrete.py:412-416); 0: final states
                              only (:429-432)                                                      */
    double scale;          /* cost_multiplier (and 1/cost_eval_count etc.) folded in               */
    const double* vectors; /* TARGET_*: [S][n] complex target states (NOT conjugated)
                              FORBID  : [sum_s F_s][n] complex forbidden states, state-major       */
    const int32_t* counts; /* FORBID: [S] F_s ; TARGET_*: ignored (may be NULL)                    */
} qocx_cost_desc;

/*
 * Static data of one Schroedinger problem; mirrors the fields of
 * GrapeSchroedingerDiscreteState / ProgramState (qoc/models/programstate.py:33-61,
 * qoc/models/schroedingermodels.py:178-206) that the evolve loop reads.
 *
 * The Hamiltonian callable of the reference (schroedingerdiscrete.py:43-46, :485) is passed
 * in structured form   H(u, t) = h0(t) + sum_k u_k g[k](t),  u real (a complex control is two
 * real ones), sampled by the host at the quadrature times of the Magnus
 * policy: nt = 1 (time independent) or nt = (system_eval_count-1) * nodes.
 */
typedef struct qocx_schroedinger_problem {
    int32_t struct_size;         /* sizeof(qocx_schroedinger_problem) of the CALLER's header: a
                                    stale binding is rejected (QOCX_ERR_ARG) instead of over-read  */
    int32_t hilbert_size;        /* n, 1..1024 (65..1024: the general path, qocx_general.hip)              */
    int32_t state_count;         /* S >= 1                                                          */
    int32_t control_count;       /* K real controls, >= 0                                           */
    int32_t control_eval_count;  /* Nc (>= 2 when K > 0)                                            */
    int32_t system_eval_count;   /* N >= 2; N-1 propagator steps                                    */
    int32_t cost_eval_step;      /* >= 1                                                            */
    int32_t magnus_policy;       /* QOCX_MAGNUS_M2 / _M4 / _M6 (mathmethods.py:72-164)               */
    int32_t nt;                  /* time samples of h0/g: 1 or (N-1)*nodes, nodes = 1/2/3 for M2/4/6,
                                    ordered [step][node], t = step*dt + c_node*dt                   */
    double evolution_time;       /* T; dt = T/(N-1), control_eval_times = linspace(0,T,Nc)          */
    const double* h0;            /* [nt][n][n] complex                                              */
    const double* g;             /* [nt][K][n][n] complex                                           */
    const double* initial_states;/* [S][n] complex                                                  */
    int32_t cost_count;
    const qocx_cost_desc* costs; /* [cost_count]                                                    */
} qocx_schroedinger_problem;

/*
 * Static data of one Lindblad problem; mirrors the fields of GrapeLindbladDiscreteState
 * (qoc/models/lindbladmodels.py:125-203) that _evaluate_lindblad_discrete reads
 * (qoc/core/lindbladdiscrete.py:357-441). hamiltonian(u, t) = h0(t) + sum_k u_k g[k](t):
 * time independent (h0, g; fixed_subdivision = 0) or sampled at the integrator's stage times
 * (fixed_subdivision > 0, h0_stages, g_stages). lindblad_data(t) = (dissipators, operators) is
 * constant (dissipators, operators) or sampled the same way (diss_stages, op_stages; the reference
 * calls it at every right-hand side, lindbladdiscrete.py:486-492).
 */
typedef struct qocx_lindblad_problem {
    int32_t struct_size;          /* sizeof(qocx_lindblad_problem) of the CALLER's header         */
    int32_t hilbert_size;         /* n, 1..32 (n > 16: four tiles per matrix, HBM scratch)         */
    int32_t density_count;        /* S >= 1                                                        */
    int32_t control_count;        /* K real controls, 0..8                                         */
    int32_t control_eval_count;   /* Nc                                                            */
    int32_t system_eval_count;    /* N >= 2                                                        */
    int32_t cost_eval_step;
    int32_t operator_count;       /* L Lindblad operators, 0..8 (17 <= n <= 32: as many as fit the LDS, 5) */
    double evolution_time;
    const double* h0;             /* [n][n] complex                                                */
    const double* g;              /* [K][n][n] complex                                             */
    const double* dissipators;    /* [L] float64 (gamma_i)                                         */
    const double* operators;      /* [L][n][n] complex                                             */
    const double* initial_densities; /* [S][n][n] complex                                          */
    int32_t cost_count;
    const qocx_cost_desc* costs;  /* kinds QOCX_COST_TARGET_DENSITY / QOCX_COST_FORBID_DENSITY     */
    /* Hamiltonian with explicit time dependence: every seed then uses `fixed_subdivision` pieces
     * per system step and h0 / g are given at the stage times of that grid
     * (qocx_lindblad_stage_times, same order). 0: time independent, sub-division chosen per seed. */
    int32_t fixed_subdivision;
    const double* h0_stages;      /* [count][n][n] complex                                         */
    const double* g_stages;       /* [count][K][n][n] complex, or NULL when g is constant          */
    /* time-dependent lindblad_data (needs fixed_subdivision > 0 and h0_stages): both or neither   */
    const double* diss_stages;    /* [count][L] float64 gamma_i(t), or NULL                        */
    const double* op_stages;      /* [count][L][n][n] complex L_i(t), or NULL                      */
} qocx_lindblad_problem;

const char* qocx_last_error(void);
int qocx_version(void);

/* Device discovery / context. device < 0: use LOCAL_RANK (or 0). */
int qocx_device_count(int* count);
int qocx_create(int device, qocx_ctx** out);
int qocx_destroy(qocx_ctx* ctx);
int qocx_synchronize(qocx_ctx* ctx);

/* Upload a problem (replaces any previous one). Performs what the construction of
 * GrapeSchroedingerDiscreteState does for the evolve loop: dt, control_eval_times,
 * step-cost selection (programstate.py:41-61). */
int qocx_set_schroedinger_problem(qocx_ctx* ctx, const qocx_schroedinger_problem* problem);

/*
 * One batched evaluation == B calls of _evaluate_schroedinger_discrete (+ its reverse-mode
 * gradient when want_grad != 0), schroedingerdiscrete.py:318-324, :356-438.
 *   controls      [B][Nc][K] float64 (ignored when K == 0)
 *   cost_out      [B]
 *   grad_out      [B][Nc][K] float64   (d cost / d controls; NULL allowed when !want_grad)
 *   final_out     [B][S][n] complex    (reporter.final_states, :435-436; NULL allowed)
 * Host buffers in, host buffers out (H2D/D2H inside the call).
 */
int qocx_eval_schroedinger(qocx_ctx* ctx, int32_t batch, const double* controls,
                           int32_t want_grad, double* cost_out, double* grad_out,
                           double* final_out);

/* Same evaluation split so that inputs can be made resident first (bench.py times this). */
int qocx_upload_controls(qocx_ctx* ctx, int32_t batch, const double* controls);
int qocx_eval_resident(qocx_ctx* ctx, int32_t want_grad);
int qocx_download_results(qocx_ctx* ctx, double* cost_out, double* grad_out, double* final_out);

/*
 * Opaque Hamiltonians. The reference calls hamiltonian(controls, time) - arbitrary Python, not
 * necessarily linear in the controls (schroedingerdiscrete.py:483-486) - at every step. For a
 * callable the structured form above cannot represent, the host samples the step generators
 * itself, M_j = -i dt H(u(t_j + dt/2), t_j + dt/2) (mathmethods.py:90-93, magnus_m2), and the
 * engine takes them as they are (problem set with control_count = 0, magnus_policy M2):
 *   generators  [B][N-1][n][n] complex, row-major
 *   qocx_eval_resident / qocx_download_results as usual (cost_out, final_out);
 *   cotangents  [B][N-1][n][n] complex: Mbar_j = d cost / d Re(M_j) + i d cost / d Im(M_j), from
 *               which the host forms d cost / d u_k = Re sum_rc conj(Mbar_j[r][c]) (dM_j/du_k)[r][c]
 *               (what autograd's tape does at :318-319 through the user's callable).
 * qocx_upload_controls switches back to the structured form.
 */
int qocx_upload_generators(qocx_ctx* ctx, int32_t batch, const double* generators);
int qocx_download_generator_cotangents(qocx_ctx* ctx, double* cotangents_out);

/* Optional: all system-step states of the last evaluation, [B][N][S][n] complex
 * (what save_intermediate_states persists, schroedingerdiscrete.py:395-402). */
int qocx_set_keep_step_states(qocx_ctx* ctx, int32_t keep); /* before the evaluation */
int qocx_download_step_states(qocx_ctx* ctx, double* states_out);

/*
 * Cotangents of the states supplied by the host, for Cost plugins whose derivative the engine
 * does not know (the reference obtains it from autograd, schroedingerdiscrete.py:412-416,
 * :429-432): bars [B][count][S][n] complex = d cost / d Re(psi) + i d cost / d Im(psi) of the
 * states at system steps steps[c] (1..N-1; N-1 = final states). They are added to the adjoint
 * sweep of every following evaluation of the same batch size; count = 0 clears them.
 */
int qocx_set_state_cotangents(qocx_ctx* ctx, int32_t batch, int32_t count, const int32_t* steps,
                              const double* bars);

/*
 * Lindblad path: B calls of _evaluate_lindblad_discrete (+ gradient), lindbladdiscrete.py:321-322,
 * :357-441. The device integrates the same master equation with a fixed-step DOP853 scheme
 * and its exact discrete adjoint (parity tolerances there).
 *   controls [B][Nc][K]; cost_out [B]; grad_out [B][Nc][K]; final_out [B][S][n][n] complex.
 * qocx_download_step_densities: [B][N][S][n][n] complex after qocx_set_keep_step_states(ctx, 1).
 */
int qocx_set_lindblad_problem(qocx_ctx* ctx, const qocx_lindblad_problem* problem);
/* The times at which a time-dependent Hamiltonian must be sampled for `subdivision` pieces per
 * system step: for every sub-interval [t_a, t_b] (uniform pieces cut at the control knots) the 12
 * DOP853 stage times t_a + c_i (t_b - t_a). times_out may be NULL to query *count_out. */
int qocx_lindblad_stage_times(double evolution_time, int32_t system_eval_count,
                              int32_t control_eval_count, int32_t control_count,
                              int32_t subdivision, double* times_out, int64_t capacity,
                              int64_t* count_out);
int qocx_eval_lindblad(qocx_ctx* ctx, int32_t batch, const double* controls, int32_t want_grad,
                       double* cost_out, double* grad_out, double* final_out);
int qocx_download_step_densities(qocx_ctx* ctx, double* densities_out);
/* Sub-intervals (12-stage DOP853 steps) the last qocx_eval_lindblad integrated, summed over its
 * seeds: the unit of the Lindblad kernel's work (bench.py prices its roofline with it). */
int qocx_lindblad_last_subintervals(qocx_ctx* ctx, int64_t* total);
/* The Lindblad twin of qocx_set_state_cotangents: bars [B][count][S][n][n] complex. */
int qocx_set_density_cotangents(qocx_ctx* ctx, int32_t batch, int32_t count, const int32_t* steps,
                                const double* bars);

/* Per-kernel timing, measured with HIP events on the launch streams.
 * enable: 0 off, 1 every launch, 2 + k the launches of kernel k only (two events per timed launch
 * cost the evaluation 2-3 % when all ~45 launches of it carry them; bench.py times its roofline
 * kernel only inside the timed region). After evaluations, qocx_get_timing returns for kernel `which`
 * (0 pade_pq, 1 sweep, 2 krylov_grad, 3 scatter, 4 lu, 5 lindblad, 6 lindblad_combine) the launch
 * count and total ms
 * since the last reset. */
int qocx_set_timing(qocx_ctx* ctx, int32_t enable);
int qocx_get_timing(qocx_ctx* ctx, int32_t which, int64_t* launches, double* total_ms);
int qocx_reset_timing(qocx_ctx* ctx);

/* Seeds per memory chunk (0 = auto from free HBM) and number of time segments of the pipeline
 * (0 = auto: 8 for large evaluations; 1 = no overlap of the sweep with the other kernels). Results
 * do not depend on either. */
int qocx_set_chunk(qocx_ctx* ctx, int32_t seeds_per_chunk);
int qocx_set_pipeline(qocx_ctx* ctx, int32_t time_segments);

/*
 * Multi-GPU: one process per GPU; the seed axis is sharded by the caller, the summed
 * cost/gradient is one RCCL all-reduce (no reference counterpart: the reference is single
 * process).  unique_id is ncclUniqueId bytes (128) produced by rank 0.
 */
int qocx_comm_unique_id(uint8_t* id128);
int qocx_comm_init(qocx_ctx* ctx, const uint8_t* id128, int32_t rank, int32_t world);
int qocx_comm_allreduce_sum(qocx_ctx* ctx, double* buf_host, int64_t count);
/* The path's single collective, device resident: out[0] = sum over the seeds of the last
 * evaluation of the cost, out[1 ..] = sum over the seeds of the gradient ([nc][K]); the sums are
 * formed on the device (fixed order: deterministic), all-reduced over the ranks of the
 * communicator with ONE ncclAllReduce on the device buffer when allreduce != 0 (qocx_comm_init
 * must have run), and only 8 (1 + nc K) bytes travel to the host. count = 1 + nc * K, or 1 for
 * the cost alone. */
int qocx_reduce_results(qocx_ctx* ctx, int32_t allreduce, double* out, int64_t count);
int qocx_comm_allreduce_max(qocx_ctx* ctx, double* buf_host, int64_t count);
int qocx_comm_barrier(qocx_ctx* ctx);
int qocx_comm_destroy(qocx_ctx* ctx);

/*
 * Debug / unit-test entry points (used by tests/, not by the host package).
 *   qocx_debug_pade_factor: K1 on explicit generator matrices a [count][n][n] complex:
 *     q_out, lu_out [count][n][n] complex (row-major, lu = L\U of the row-permuted P),
 *     perm_out [count][n], dinv_out [count][n] complex (1/U_kk), s_out [count].
 *   qocx_debug_selftest: wave-level primitives (DPP reductions, MFMA layout); returns the
 *     number of failed checks in *failures.
 */
int qocx_debug_pade_factor(qocx_ctx* ctx, int32_t count, int32_t n, const double* a,
                           double* q_out, double* lu_out, int32_t* perm_out,
                           double* dinv_out, int32_t* s_out);
int qocx_debug_selftest(qocx_ctx* ctx, int32_t* failures, char* report, int32_t report_len);
/* Kernel-variant switches for A/B measurements in one process and for tests that pin a variant.
 * Every knob libqocx.so accepts selects between paths that give the same numbers to rounding;
 * no environment variable changes which kernels the product library runs. Known names:
 *   "sweep_impl"    1 (default): column-chain sweep (one wavefront per seed; best inside the
 *                   segmented pipeline of large batches); 3: blocked-inverse sweep (four wavefronts
 *                   per seed: compute | inverter L | loader | inverter U'; qocx_sweep3.hip), 1.5x
 *                   faster per step when the sweep has the chip to itself (<= 128 seeds). One
 *                   implementation serves all batch sizes of a context, so results stay bit
 *                   identical across batching.
 *   "sweep3_phases" with the blocked sweep selected: bit 0 forward launches, bit 1 adjoint.
 *   "sweep_loader"  column-chain sweep only. 0 (default): the compute wave issues the LDS-DMA from
 *                   inside its triangular solves; 1: a dedicated fetch wave per seed does.
 *   "pade_order"    0 (default): Pade order 3 / 5 / 7 / 9 / 13 by the 1-norm of the step generator;
 *                   13: always [13/13], as the reference executes it (qocx_pade_orders).
 *   "unit_adjoint"  one final TargetStateInfidelity: the adjoint sweep back-propagates the targets.
 *   "bidir"         with it: forward and adjoint sweep side by side, factorisation from both ends.
 *   "fuse_lu"       17 <= n <= 32: the LU factorisation runs inside the Pade kernel;
 *   "lu_mfma"       (round 4) ... and takes its Schur updates to the matrix cores (qocx_lu4.h).
 *   "latency"       0 (default; the host sets 1 for entry points that evaluate ONE control set):
 *                   four time segments, the inverse-image sweep.
 *   "sweep_inverse" latency mode: sub-steps as two matrix-vector products with P^-1 (qocx_sweepi.hip);
 *   "sweep_inverse_small": the same sweep for every batch at n <= 16.
 *   "sweep_dense"   8 <= S <= 32 states at 17 <= n <= 32 as MFMA GEMM columns (qocx_sweepd.hip);
 *   "krylov_dense"  (default 0) K3 on the matrix cores for those problems; "lu_inverse": the debug
 *                   factor entry point returns P^-1.
 *   "m4_linear"     MagnusPolicy.M4 with time-independent H0, G_k on the M2 kernels (commutators
 *                   hoisted into constant matrices); "magnus_4w": four-wave LDS-resident Magnus
 *                   kernels at 17 <= n <= 32; "magnus_general": the general commutator forms.
 *   "lindblad_two_sided", "lindblad_side_limit": forward and unit-adjoint Lindblad passes side by side.
 *   "k1a_herm4": four-wave K1a, Hermitian generators, orders 3..9: two thirds of the tiles (0: all tiles).
 *   "lu_stream": n > 32, K1b of a time segment on a stream of its own beside K1a of the next (0: one stream).
 *   "lindblad_q2": their stage loop with 18 of the 72 MFMAs of a right-hand side per wave (0: the quarter-split loops).
 *   "lindblad_4t": Lindblad at 17 <= n <= 32 on the tile-per-wave kernel where it applies (0: one wave per seed).
 *   "lindblad_hermitian": that kernel's shorter stages for Hermitian problems (Y A_R = (A_L Y)^H; 0: the general stages).
 *   "lindblad_pad_operator": Lindblad with ONE operator at n <= 16 on the four-wave launches of two (the second zero);
 *                   read when the problem is set (0: the three-wave form).
 *   "sweep_onebuf", "k3_split": launch shapes of the sweep / of K3.
 * Diagnostic knobs - libqocx_diag.so only (make diag, -DQOCX_DIAG; the product library answers
 * QOCX_ERR_ARG): "dbg_skip", "sweep3_dbg", "k1a_dbg" switch parts of an evaluation off for timing
 * (results are garbage); "sweep3_stamps", "lindblad_stamps", "k1a_stamps" run kernel builds that
 * execute in-kernel clock stamps (qocx_debug_read_stamps). The diagnostic library also reads the
 * QOCX_* environment switches of earlier experiments (qoc_amd/csrc/qocx_diag.h). */
int qocx_debug_set_knob(qocx_ctx* ctx, const char* name, int64_t value);
/* 1: a variant knob (every build), 2: a diagnostic knob this library accepts, -2: a diagnostic
 * knob this (product) library rejects, 0: unknown. No context, no GPU needed. */
int qocx_knob_kind(const char* name);
/* 1 for libqocx_diag.so, 0 for the product library. */
int qocx_build_is_diag(void);
/* After evaluations with the knob "sweep3_stamps" = 1 (a diagnostic build of the sweep that
 * executes in-kernel clock stamps; never the product kernel): per seed, per role (compute |
 * inverter L | loader | inverter U'), 8 sums - shader-clock cycles per phase of the role's step loop, [7] = the
 * 100 MHz real-time counter over the role's life. out: [batch][4][8]. */
int qocx_debug_read_stamps(qocx_ctx* ctx, uint64_t* out, int64_t count);
/* Pade orders of the last Schroedinger evaluation (its last memory chunk): counts[0..4] = number of
 * propagator steps evaluated with the [3/3], [5/5], [7/7], [9/9], [13/13] approximant. The engine
 * takes the order from the 1-norm of the step generator by the thresholds of Higham 2005,
 * Algorithm 2.3 - the table /root/reference/qoc/standard/functions/expm.py:194-209 carries; the
 * reference itself always evaluates [13/13] (expm.py:230-233: its selection loop has no break),
 * which the knob "pade_order" = 13 reproduces. Same matrix and same derivative to rounding. */
int qocx_pade_orders(qocx_ctx* ctx, int64_t* counts);
/* Matrices of the last Schroedinger evaluation (or qocx_debug_pade_factor call) whose LU factorisation
 * left the diagonal-pivot form with MFMA Schur updates (knob "lu_mfma"; qoc_amd/csrc/qocx_lu4.h,
 * qocx_lu4m.hip) for the general partial-pivoting elimination: LAPACK's pivot rule asked for a row
 * interchange somewhere. 0 for the Pade denominators of well-scaled generators. */
int qocx_lu_fallbacks(qocx_ctx* ctx, int64_t* count);
/* With qocx_set_timing on: the kernel launches of the LAST evaluation as (which, start_ms, end_ms)
 * triples relative to its first launch (HIP events on the launch streams; which = the index of
 * qocx_get_timing: 0 K1a, 1 sweep, 2 K3, 3 scatter, 4 K1b, 5 Lindblad, 6 its combine kernel). out
 * holds up to `capacity`
 * triples, *count receives how many there were. A view of how the pipeline overlaps. */
int qocx_debug_timeline(qocx_ctx* ctx, double* out, int64_t capacity, int64_t* count);
/* Force the variants of the Lindblad launch that large batches / little free HBM select:
 *   stage_budget_seeds  seeds whose forward stage values may be kept for the adjoint (0: as many
 *                       as fit 45 % of free HBM); a larger group is launched in pieces;
 *   min_piece           below this many seeds per piece the adjoint recomputes the stage values
 *                       from the checkpoints instead (default 256);
 *   wave_mode           0 auto (several waves per seed, batches beyond the CU count in rounds of
 *                       one seed per CU), 1 one wave per seed, 2 several waves per seed in ONE
 *                       launch whatever the batch size.
 * Results do not depend on any of them (tests/test_gpu_lindblad.py). */
int qocx_debug_lindblad_knobs(qocx_ctx* ctx, int64_t stage_budget_seeds, int32_t min_piece,
                              int32_t wave_mode);
/* Sustained FP64 MFMA rate of the device: waves_per_simd x 4 x CUs waves issue `iters` rounds of
 * 8 independent v_mfma_f64_16x16x4_f64 from registers (no memory traffic). bench.py reports it
 * next to the spec peak. */
int qocx_debug_mfma_peak(qocx_ctx* ctx, int32_t waves_per_simd, int32_t iters, double* tflops);

/* ---- multi-start GRAPE with the optimizer states resident on the device ------------------------
 * The reference's driver iteration - clip the controls, evaluate, keep the best so far, apply the
 * optimizer (qoc/core/schroedingerdiscrete.py:293-353, common.py:8-30, optimizers/adam.py:110-165,
 * sgd.py) - for the B control sets that qocx_upload_controls left in HBM, without a trip to the host:
 * per iteration 8 B bytes of costs come back, B flags go in. Real controls, built-in Adam / SGD
 * (no scale_grads). IEEE operations in the reference's order, no contraction: every seed walks its
 * single-seed trajectory bit for bit.
 *   qocx_opt_begin          after qocx_upload_controls: zero the Adam moments, allocate the
 *                           best-so-far buffers for the current batch.
 *   qocx_opt_clip           clip_control_norms on the resident controls, in place; the squaring
 *                           capacity of the next evaluation is re-derived from max_norms [K].
 *   qocx_download_costs     the B costs of the last evaluation.
 *   qocx_opt_step           improved[b] != 0: best controls / final states of seed b := those of the
 *                           last evaluation; then update[b] != 0: seed b takes the optimizer step
 *                           (kind 0 SGD, 1 Adam) with the gradients of the last evaluation;
 *                           corr_i = 1 - beta_i^step and the learning rate are the caller's scalars.
 *   qocx_opt_download_best  best controls [B][Nc][K] and best final states [B][S][n] complex. */
int qocx_opt_begin(qocx_ctx* ctx);
int qocx_opt_clip(qocx_ctx* ctx, const double* max_norms);
int qocx_download_costs(qocx_ctx* ctx, double* cost_out);
int qocx_opt_step(qocx_ctx* ctx, int32_t kind, const uint8_t* improved, const uint8_t* update,
                  double learning_rate, double beta_1, double beta_2, double epsilon, double corr_1,
                  double corr_2, int32_t apply_clip_grads, double clip_grads);
int qocx_opt_download_best(qocx_ctx* ctx, double* controls_out, double* final_out);

/* ---- host-side helpers of the multi-start GRAPE driver (no GPU work, no context) --------------
 * The reference's driver loop clips the controls and applies its optimizer plugin to ONE control
 * set per process (qoc/core/common.py:8-30, qoc/standard/optimizers/adam.py:110-165, sgd.py). The
 * batched driver holds B optimizer states as [B][P] arrays; these two functions are that loop's
 * arithmetic over the rows `rows[0..row_count)` on a few host threads - IEEE double operations in
 * the reference's order (no contraction), so every seed walks its single-seed trajectory bit for
 * bit. */
/* controls [B][Nc][K] real, in place: an entry whose modulus exceeds max_norms[k] becomes
 * (entry / modulus) * max_norms[k] (clip_control_norms). */
int qocx_host_clip_controls(double* controls, int64_t batch, int64_t nc, int32_t k,
                            const double* max_norms);
/* Adam.update on rows of params / grads / moment / square_moment (all [B][P]):
 *   g' = clip(g, -clip_grads, clip_grads) if apply_clip_grads
 *   m = beta_1 m + (1 - beta_1) g' ; v = beta_2 v + (1 - beta_2) g'^2
 *   params -= learning_rate * ((m / corr_1) / (sqrt(v / corr_2) + epsilon))
 * corr_i = 1 - beta_i^step and the (possibly decayed) learning rate are the caller's scalars.
 * kind 0 = SGD (params -= learning_rate * g; the other arguments are ignored). */
int qocx_host_optimizer_update(int32_t kind, double* params, const double* grads, double* moment,
                               double* square_moment, int64_t p, const int64_t* rows,
                               int64_t row_count, double learning_rate, double beta_1,
                               double beta_2, double epsilon, double corr_1, double corr_2,
                               int32_t apply_clip_grads, double clip_grads);

#ifdef __cplusplus
}
#endif
#endif /* QOCX_H */
