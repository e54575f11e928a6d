"""
bench.py - propagator-steps/sec (fwd+grad) of the GRAPE hot path on MI355X.

Workload (BASELINE.json configs[2]): dim=32 Schroedinger, 1000 propagator steps,
256 random control seeds PER GPU (weak scaling), K=2 real controls, S=1 state,
TargetStateInfidelity cost; synthetic GUE-like H0 and G_k of unit 2-norm, dt = 0.05,
controls ~ N(0, 0.1^2), all seeded.

One "step" = one evaluation (cost, d cost / d controls, final states) of every seed of every
rank, controls resident in HBM when the clock starts (`value`, as the bench contract defines it).
With --gpus N > 1 the driver launches one process per GPU (torchrun env); the seed axis is
sharded and each step ends with the path's single RCCL all-reduce of [sum cost, sum gradient].

The line also carries, measured in the same run on rank 0:
  host_to_host  the same evaluation through qocx_eval_schroedinger - FRESH controls from host
                memory every step, costs + gradients + final states back in host memory (PCIe
                inclusive; SURVEY.md 8d's "controls in, results out");
  secondary     BASELINE.json configs[3]: dim=16 Lindblad, 500 steps, 64 seeds - ms per
                evaluation, steps/s and the roofline of the Lindblad kernel;
  general_path  the path for Hilbert sizes above 64 (qocx_general.hip) at dim = 128, 16 seeds x 250 steps;
  cpu_baseline  the oracle on the host cores (all cores, and one core: the reference's execution
                model).

    python bench.py --gpus 1 --steps 5 --warmup 2
"""

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

DIM, N_EVAL, SEEDS_PER_GPU, K_CTRL, DT = 32, 1001, 256, 2, 0.05
FP64_MFMA_PEAK_TFLOPS = 78.6  # MI355X FP64 matrix peak (spec; SURVEY.md 8d / BASELINE.md 3)
# The rate a register-only loop of v_mfma_f64_16x16x4_f64 sustains on the box is measured live
# (qocx_debug_mfma_peak) and reported as roofline.peak_sustained_measured: ~68 TFLOP/s on one accumulation chain (48 on eight).


def gue(rng, n):
    g = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
    h = (g + g.conj().T) / 2
    return h / np.linalg.norm(h, 2)


def make_problem():
    rng = np.random.default_rng(2003)
    h0 = gue(rng, DIM)
    g = [gue(rng, DIM) for _ in range(K_CTRL)]
    psi0 = np.eye(DIM, dtype=np.complex128)[:1]
    target = np.eye(DIM, dtype=np.complex128)[1:2]
    return h0, g, psi0, target


def make_controls(first_seed, count):
    out = np.empty((count, N_EVAL, K_CTRL))
    for b in range(count):
        out[b] = 0.1 * np.random.default_rng(1000 + first_seed + b).standard_normal((N_EVAL, K_CTRL))
    return out


PMC_SUMMARY = "profiles/r05_pmc_hbm.json"
PMC_SQ_SUMMARY = "profiles/r05_pmc_sq.json"  # SQ counters of the same command (tools/profile_round.sh)
PMC_META = "profiles/r05_profile_meta.json"  # what the counter passes were taken from (kernel sources)
K1A_KERNEL = "qocx::pade3::pade_pq3_kernel<false, false>"  # the roofline kernel of the headline workload
# the kernel sources whose change makes the committed counters of K1A_KERNEL stale
K1A_SOURCES = ("qoc_amd/csrc/qocx_pade3.hip", "qoc_amd/csrc/qocx_lu5.h", "qoc_amd/csrc/qocx_wave.h")


def source_digest(paths=K1A_SOURCES):
    import hashlib
    h = hashlib.sha256()
    for rel in paths:
        try:
            with open(os.path.join(ROOT, rel), "rb") as f:
                h.update(f.read())
        except OSError:
            h.update(b"missing:" + rel.encode())
    return h.hexdigest()[:16]


def counters_current():
    """True if the committed counter summaries were taken from the kernel sources of this tree
    (tools/profile_round.sh records their digest in PMC_META)."""
    try:
        with open(os.path.join(ROOT, PMC_META)) as f:
            return json.load(f).get("k1a_sources_sha16") == source_digest()
    except (OSError, ValueError):
        return False


def pmc_executed_mfma_flops(kernel, units_per_dispatch=32000):
    """
    Executed FP64 MFMA flops per propagator step of `kernel` (exact name) from the committed SQ
    counter pass (SQ_INSTS_VALU_MFMA_MOPS_F64 per dispatch; one v_mfma_f64_16x16x4_f64 = 2048 flop
    counts 4, i.e. 512 flop per counted operation - calibrated in round 2 against the instruction
    count of the kernel). None if the summary or the kernel is absent.
    """
    try:
        with open(os.path.join(ROOT, PMC_SQ_SUMMARY)) as f:
            summary = json.load(f)
        return summary[kernel]["SQ_INSTS_VALU_MFMA_MOPS_F64"] * 512.0 / units_per_dispatch
    except (OSError, KeyError, ValueError, TypeError):
        return None


def pmc_traffic_bytes(kernel, units_per_launch):
    """
    HBM bytes per launch of `kernel` (exact name) from the committed rocprofv3 PMC summary
    (PMC_SUMMARY: FETCH_SIZE x2 + WRITE_SIZE, separate --pmc passes of this bench command, gfx950
    correction applied; tools/profile_round.sh + tools/pmc_summary.py), rescaled to this run's
    units per launch. PMC counters cannot be collected from inside the timed run; None if the
    summary or the kernel is absent.
    """
    path = os.path.join(ROOT, PMC_SUMMARY)
    try:
        with open(path) as f:
            summary = json.load(f)
        entry = summary["kernels"][kernel]
        return (entry["hbm_bytes_per_dispatch_corrected"] * units_per_launch
                / summary["units_per_dispatch"])
    except (OSError, KeyError, ValueError):
        return None


# kernels of one headline evaluation: timing class (qocx_set_timing) -> kernel names in the profiles
EVAL_KERNELS = {
    "pade_pq": (K1A_KERNEL,),
    "sweep": ("qocx::sweep1::sweep1_kernel<2, 1>",),
    "krylov_grad": ("qocx::krylov_grad_skew_kernel<2, false>",),
    "scatter": ("qocx::scatter_kernel",),
}


def pmc_evaluation_bytes(launches_per_evaluation):
    """HBM bytes of ONE headline evaluation from the committed PMC summary: the average bytes per
    dispatch of every kernel of the evaluation x its launches per evaluation (this run's count),
    plus the step table and the reduction (one launch each). None if a kernel is missing."""
    try:
        with open(os.path.join(ROOT, PMC_SUMMARY)) as f:
            kernels = json.load(f)["kernels"]
        total, parts = 0.0, {}
        for cls, names in EVAL_KERNELS.items():
            per = sum(kernels[n]["hbm_bytes_per_dispatch_corrected"] for n in names)
            parts[cls] = per * launches_per_evaluation[cls]
            total += parts[cls]
        for name in ("qocx::step_table_kernel", "qocx::reduce_results_kernel"):
            parts[name] = kernels[name]["hbm_bytes_per_dispatch_corrected"]
            total += parts[name]
        return total, parts
    except (OSError, KeyError, ValueError):
        return None, None


# ---- CPU baseline: the oracle (NumPy restatement of the reference + hand adjoint) ----------

def _cpu_worker(seed_ids):
    try:
        from threadpoolctl import threadpool_limits
        limiter = threadpool_limits(1)
    except Exception:  # pragma: no cover
        limiter = None
    from oracle import qoc_numpy as onp
    h0, g, psi0, target = make_problem()
    problem = onp.SchroedingerProblem(
        DT * (N_EVAL - 1), lambda u, t: h0 + u[0] * g[0] + u[1] * g[1], psi0[:, :, None], N_EVAL,
        control_eval_count=N_EVAL, costs=[onp.TargetStateInfidelity(target[:, :, None])],
        control_count=K_CTRL)
    t0 = time.perf_counter()
    for sid in seed_ids:
        onp.evaluate_with_grad(problem, make_controls(sid, 1)[0])
    del limiter
    return time.perf_counter() - t0


def host_cores():
    """(cores this process may use, cores of the machine, how the first number was found). The GPU
    box gives a one-GPU job a CPU share - a cgroup quota of 16 CPUs on a 256-core host whose scheduler
    affinity still lists every core - so the quota, not os.cpu_count(), is what can be used."""
    total = os.cpu_count() or 1
    usable, how = total, "os.cpu_count()"
    try:
        usable, how = len(os.sched_getaffinity(0)), "scheduler affinity"
    except (AttributeError, OSError):  # pragma: no cover
        pass
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: t.split()),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", lambda t: (t.strip(), None))):
        try:
            with open(path) as f:
                quota, period = parse(f.read())
            if period is None:
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                    period = f.read().strip()
            if quota not in ("max", "-1") and float(period) > 0:
                share = max(1, int(float(quota) / float(period) + 0.5))
                if share < usable:
                    usable, how = share, "cgroup CPU quota ({})".format(path)
            break
        except (OSError, ValueError):
            continue
    return max(1, usable), total, how


def cpu_baseline(budget_seconds=12.0):
    """Oracle fwd+grad on a bounded sample: one worker process per usable core (all of them: the
    CPU share of the box), each with a block of seeds sized from a timed
    probe so that the sample takes about `budget_seconds`; plus the single-core figure (one seed
    at a time, the reference's execution model)."""
    import multiprocessing as mp
    cores, total, how = host_cores()
    ctx = mp.get_context("spawn")
    with ctx.Pool(cores) as pool:
        pool.map(_cpu_worker, [[0]] * cores)  # process start, imports, BLAS warm-up: not timed
        probe = max(pool.map(_cpu_worker, [[1]] * cores))  # one seed per worker, all cores busy
        seeds_per_worker = int(max(4, min(64, round(budget_seconds / max(probe, 1e-3)))))
        jobs = [[w * seeds_per_worker + i for i in range(seeds_per_worker)] for w in range(cores)]
        t0 = time.perf_counter()
        pool.map(_cpu_worker, jobs)
        wall = time.perf_counter() - t0
        single_seconds = pool.apply(_cpu_worker, (list(range(1000, 1012)),))  # others idle
    steps = cores * seeds_per_worker * (N_EVAL - 1)
    return dict(value=steps / wall, unit="propagator-steps/s", cores=cores, cores_available=total,
                cores_note="cores = every CPU this job may use ({}), one worker process each; "
                           "cores_available = os.cpu_count() of the host".format(how),
                kind="port",
                sample="{} seeds x {} steps (oracle/qoc_numpy.py fwd+grad, {} processes, "
                       "1 BLAS thread each, {:.1f}s wall, warm processes)".format(
                           cores * seeds_per_worker, N_EVAL - 1, cores, wall),
                single_core=dict(value=12 * (N_EVAL - 1) / single_seconds,
                                 unit="propagator-steps/s", cores=1,
                                 sample="12 seeds x {} steps, one process, 1 BLAS thread, "
                                        "{:.1f}s".format(N_EVAL - 1, single_seconds)))


# ---- secondary: BASELINE.json configs[3], the Lindblad path ------------------------------------

LB_DIM, LB_EVAL, LB_SEEDS, LB_OPS = 16, 501, 64, 2


def lindblad_problem():
    rng = np.random.default_rng(2004)
    h0 = gue(rng, LB_DIM)
    g = [gue(rng, LB_DIM) for _ in range(K_CTRL)]
    a = np.diag(np.sqrt(np.arange(1, LB_DIM)), 1).astype(np.complex128)
    ops = np.stack([a, a.conj().T @ a])
    gam = np.array([0.05, 0.02])
    rho0 = np.zeros((1, LB_DIM, LB_DIM), dtype=np.complex128)
    rho0[0, 0, 0] = 1
    target = np.zeros((1, LB_DIM, LB_DIM), dtype=np.complex128)
    target[0, 1, 1] = 1
    return h0, g, gam, ops, rho0, target


def latency_secondary(engine, reps=200):
    """ms per forward + gradient evaluation of ONE control set: BASELINE configs[1] (dim = 8 single
    transmon, 500 steps, 1 seed) and the headline shape with one seed - what a GRAPE iteration of
    an ordinary qoc script waits for (knob "latency", as the single-control-set entry points set it)."""
    from qoc_amd.engine import COST_TARGET_COHERENT
    out = []
    engine.set_timing(False)
    for label, n, steps in (("configs[1]: dim=8 transmon, 500 steps, 1 seed", 8, 500),
                            ("dim=32 (headline shape), 1000 steps, 1 seed", DIM, N_EVAL - 1)):
        if n == 8:  # a + a^dagger / i (a - a^dagger) drives on a weakly anharmonic oscillator
            a = np.diag(np.sqrt(np.arange(1, n)), 1).astype(np.complex128)
            ad = a.conj().T
            h0 = 2 * np.pi * 0.05 * ad @ a + 0.5 * 2 * np.pi * (-0.2) * ad @ ad @ a @ a
            g = [a + ad, 1j * (a - ad)]
            psi0 = np.eye(n, dtype=np.complex128)[:1]
            target = np.eye(n, dtype=np.complex128)[1:2]
        else:
            h0, g, psi0, target = make_problem()
        engine.set_schroedinger_problem(
            n, 1, K_CTRL, steps + 1, steps + 1, DT * steps, h0[None], np.stack(g)[None], psi0,
            costs=[dict(kind=COST_TARGET_COHERENT, step_cost=0, scale=1.0, vectors=target)])
        u = 0.1 * np.random.default_rng(77).standard_normal((1, steps + 1, K_CTRL))
        engine.set_knob("latency", 1)
        engine.set_knob("sweep_impl", 3)
        for _ in range(5):
            engine.evaluate(u, True)
        engine.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            engine.evaluate(u, True)
        wall = (time.perf_counter() - t0) / reps
        engine.set_knob("latency", 0)
        engine.set_knob("sweep_impl", 1)
        out.append(dict(workload=label, ms_per_evaluation=wall * 1e3,
                        value=steps / wall, unit="propagator-steps/s"))
    return out


def general_secondary(engine, n=128, seeds=16, steps=250):
    """The path for Hilbert sizes above 64 (qoc_amd/csrc/qocx_general.hip): the headline's GUE problem at
    dim = 128, forward + gradient, with the flops of its Pade products and inversion against the FP64 peak."""
    global DIM
    from qoc_amd.engine import COST_TARGET_COHERENT
    keep = DIM
    DIM = n
    try:
        h0, g, psi0, target = make_problem()
        controls = make_controls(0, seeds)[:, :steps + 1]
    finally:
        DIM = keep
    engine.set_schroedinger_problem(
        n, 1, K_CTRL, steps + 1, steps + 1, DT * steps, h0[None], np.stack(g)[None], psi0,
        costs=[dict(kind=COST_TARGET_COHERENT, step_cost=0, scale=1.0, vectors=target)])
    engine.upload_controls(np.ascontiguousarray(controls))
    engine.set_timing(True)
    engine.eval_resident(True)
    engine.synchronize()
    engine.reset_timing()
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        engine.eval_resident(True)
    engine.synchronize()
    wall = (time.perf_counter() - t0) / reps
    orders = engine.pade_orders()
    kernels = {k: v[1] / reps for k, v in engine.timing().items() if v[0]}
    engine.set_timing(False)
    products = {3: 2, 5: 3, 7: 4, 9: 5, 13: 6}
    total = max(1, sum(orders.values()))
    # complex n^3 products of the Pade chain + the n^3 of the explicit inverse, 8 flops per complex multiply-add
    flops_per_step = 8.0 * n ** 3 * (sum(products[o] * c for o, c in orders.items()) / total + 1.0)
    factor_ms = kernels.get("pade_pq", 0.0)
    return dict(workload="dim={} Schroedinger (general path, qocx_general.hip), {} steps, {} seeds".format(n, steps, seeds),
                ms_per_evaluation=wall * 1e3, value=seeds * steps / wall, unit="propagator-steps/s",
                pade_orders={str(o): c for o, c in orders.items() if c},
                kernel_ms_per_evaluation=kernels,
                roofline=dict(bound="mfma", kernel="qocx::general::factor_kernel", flops_per_step=flops_per_step,
                              achieved=flops_per_step * seeds * steps / (factor_ms * 1e-3) / 1e12 if factor_ms else None,
                              peak=FP64_MFMA_PEAK_TFLOPS, unit="TFLOP/s"))


# the two-sided launches of configs[3] (chain form of the stage loop), as the profiles name them
LB_KERNEL = "qocx::lindblad_kernel<1, false, true, true, false, true, true>"


def lindblad_secondary(engine, reps=5):
    """ms per fwd+grad evaluation of configs[3] and the roofline of the Lindblad kernel.
    Algorithmic work per sub-interval (one 12-stage DOP853 step of one seed): the
    right-hand side is (2 + 2 L) n^3 complex MACs (A_L rho, rho A_R, L_i rho, (.) L_i^H), 12
    stages forward; the adjoint recomputes nothing (stage values are kept) but applies the
    transposed right-hand side (the same count) and the control cotangent products
    Y k^H - k^H Y (2 n^3): 12 x ((2 + 2 L) x 2 + 2) n^3 complex MACs x 8 flops."""
    from qoc_amd.engine import COST_TARGET_DENSITY
    h0, g, gam, ops, rho0, target = lindblad_problem()
    engine.set_lindblad_problem(
        LB_DIM, 1, K_CTRL, LB_EVAL, LB_EVAL, DT * (LB_EVAL - 1), h0, g, gam, ops, rho0,
        costs=[dict(kind=COST_TARGET_DENSITY, step_cost=0, scale=1.0, vectors=target)])
    u = np.empty((LB_SEEDS, LB_EVAL, K_CTRL))
    for b in range(LB_SEEDS):
        u[b] = 0.1 * np.random.default_rng(1000 + b).standard_normal((LB_EVAL, K_CTRL))
    engine.evaluate_lindblad(u)  # warm-up
    engine.reset_timing()
    t0 = time.perf_counter()
    for _ in range(reps):
        cost, grads, final = engine.evaluate_lindblad(u)
    wall = (time.perf_counter() - t0) / reps
    timing = engine.timing()
    launches, total_ms = timing["lindblad"]
    subs = engine.lindblad_last_subintervals()
    flops_per_sub = 8.0 * 12 * ((2 + 2 * LB_OPS) * 2 + 2) * LB_DIM ** 3
    # Round 3: the evaluation is three kernels - forward pass and unit adjoint side by side on two
    # streams (the same lindblad_kernel, LindbladArgs::phase 1 / 2), then lindblad_combine. The
    # roofline divides the evaluation's algorithmic flops by the device time from the first launch
    # to the end of the combine kernel (HIP events, qocx_debug_timeline of the last evaluation).
    spans = [(a, b) for w, a, b in engine.timeline() if int(w) in (5, 6)]
    kernel_s = (max(b for _, b in spans) - min(a for a, _ in spans)) * 1e-3 if spans else 0.0
    achieved = flops_per_sub * subs / kernel_s / 1e12 if kernel_s > 0 else 0.0
    combine_launches, combine_ms = timing.get("lindblad_combine", (0, 0.0))
    return {
        "config": {"workload": "configs[3]: dim=16 Lindblad, 500 system steps, 64 seeds, L=2 "
                               "operators, K=2 real controls, S=1, fixed-step DOP853 + exact "
                               "discrete adjoint"},
        "metric": "propagator-steps/sec (fwd+grad), dim=16 Lindblad, 500 steps x 64 seeds",
        "value": LB_SEEDS * (LB_EVAL - 1) / wall, "unit": "propagator-steps/s",
        "ms_per_eval": wall * 1e3, "subintervals_per_step": subs / (LB_SEEDS * (LB_EVAL - 1.0)),
        "cus_occupied": {"busy": 2 * LB_SEEDS, "of": 256,
                         "note": "one workgroup (one CU) per seed and pass: forward pass and unit "
                                 "adjoint side by side = 2 x 64 of the 256 CUs; the roofline fraction "
                                 "below is against the whole chip"},
        "roofline": {"bound": "mfma",
                     "kernel": LB_KERNEL + " (forward || unit adjoint) + lindblad_combine",
                     "achieved": achieved,
                     "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": achieved / FP64_MFMA_PEAK_TFLOPS, "device_ms_per_eval": kernel_s * 1e3,
                     "avg_launch_ms": {"lindblad_kernel": total_ms / max(launches, 1),
                                       "lindblad_combine": combine_ms / max(combine_launches, 1)},
                     "flops_per_subinterval": flops_per_sub,
                     "note": "algorithmic flops: the complex count of the right-hand side. configs[3]'s "
                             "Lindblad operators (a, a^dagger a) are REAL matrices: the kernel notices and "
                             "spends two real products per complex one on them (16 MFMAs per operator and "
                             "stage instead of 24); complex operators take the general path "
                             "(knob lindblad_real_ops, tests/test_gpu_lindblad.py)",
                     "traffic": pmc_traffic_bytes(LB_KERNEL, 32000),
                     "traffic_source": "committed rocprofv3 --pmc passes ({}), per launch of this "
                                       "workload".format(PMC_SUMMARY)},
        "check": {"sum_cost": float(cost.sum()),
                  "trace_defect": float(np.max(np.abs(np.trace(final[:, 0], axis1=-2, axis2=-1) - 1)))},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--seeds-per-gpu", type=int, default=SEEDS_PER_GPU)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-reference-order", action="store_true",
                    help="skip the extra pass with the Pade order pinned to 13 (profiling runs: the "
                         "kernel statistics then hold the timed configuration only)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the configs[3] Lindblad measurement")
    ap.add_argument("--no-latency", action="store_true",
                    help="skip the one-control-set latency block (profiling runs: its 400 small "
                         "evaluations would swamp the per-kernel averages)")
    ap.add_argument("--time-segments", type=int, default=0,
                    help="tuning knob: time segments of the pipeline (0 = the engine's choice)")
    ap.add_argument("--standin-engine", default=None, metavar="MODULE",
                    help="TESTS ONLY (tests/test_bench_main.py): a stand-in engine module in the "
                         "place of libqocx, to rehearse the multi-rank control flow without a GPU")
    args = ap.parse_args()

    from qoc_amd import parallel
    standin = args.standin_engine
    if standin:
        # CPU rehearsal of the multi-rank control flow (tests/test_bench_main.py): a stand-in engine
        # module in the place of libqocx, gloo in the place of RCCL. The line says so in `data`;
        # nothing of the hot path is computed or measured.
        import importlib
        mod = importlib.import_module(standin)
        Engine, COST_TARGET_COHERENT = mod.Engine, mod.COST_TARGET_COHERENT
    else:
        from qoc_amd.engine import Engine, COST_TARGET_COHERENT

    rank, world, local_rank = parallel.env_world()
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node {}".format(args.gpus))
        raise SystemExit("--gpus {} but WORLD_SIZE={}".format(args.gpus, world))
    seeds = args.seeds_per_gpu

    # (a launcher that shows every rank one device only: local rank 3 still finds device 0)
    visible = Engine.device_count() if hasattr(Engine, "device_count") else 0
    engine = Engine(local_rank % visible if visible > 0 else local_rank)
    comm = parallel.RcclComm(engine, rank, world) if world > 1 else parallel.SingleComm()
    h0, g, psi0, target = make_problem()
    engine.set_schroedinger_problem(
        DIM, 1, K_CTRL, N_EVAL, N_EVAL, DT * (N_EVAL - 1), h0[None], np.stack(g)[None], psi0,
        costs=[dict(kind=COST_TARGET_COHERENT, step_cost=0, scale=1.0, vectors=target)])
    controls = make_controls(rank * seeds, seeds)
    if args.time_segments > 0:
        engine.set_pipeline(args.time_segments)
    engine.upload_controls(controls)  # resident in HBM before the clock starts

    def one_step():
        # evaluation, then the path's single collective: sum over this rank's seeds on the device,
        # one ncclAllReduce over the ranks (N > 1), 16 KB to the host
        engine.eval_resident(True)
        return parallel.summed_results_resident(engine, comm)

    # kernel timing (HIP events on the launch streams) is already on during the warm-up, so that
    # the event pool and the runtime's signal pools exist before the clock starts
    engine.set_timing(True)
    for _ in range(3):  # set-up: first-touch of the result buffers, event / signal pools
        one_step()
    engine.reset_timing()
    for _ in range(max(1, args.warmup)):
        one_step()
    warm_timing = engine.timing()  # every kernel, from the warm-up steps
    # inside the timed region only the roofline kernel carries events (two events on each of the
    # ~45 launches of an evaluation cost 2-3 % of it)
    engine.set_timing(True, only="pade_pq")
    engine.reset_timing()
    comm.barrier()
    engine.synchronize()
    t0 = time.perf_counter()
    step_ms = []
    for _ in range(args.steps):
        t_step = time.perf_counter()
        total_cost, total_grad = one_step()
        step_ms.append(round((time.perf_counter() - t_step) * 1e3, 2))
    engine.synchronize()
    comm.barrier()
    elapsed = time.perf_counter() - t0
    elapsed = float(comm.allreduce_max(np.array([elapsed]))[0])
    timing = engine.timing()

    # the same evaluation host buffer to host buffer (rank 0 of a single-GPU run): fresh controls
    # from host memory every step, costs + gradients + final states back in host memory
    host_to_host = None
    if world == 1 and not standin:
        fresh = [make_controls(5000 + 300 * k, seeds) for k in range(3)]
        for warm in fresh:  # first-touch of the result staging on this route
            engine.evaluate(warm, want_grad=True)
        engine.synchronize()
        t_h = time.perf_counter()
        for k in range(args.steps):
            engine.evaluate(fresh[k % 3], want_grad=True)
        h2h = (time.perf_counter() - t_h) / args.steps
        host_to_host = dict(ms_per_step=h2h * 1e3, value=seeds * (N_EVAL - 1) / h2h,
                            unit="propagator-steps/s",
                            note="qocx_eval_schroedinger: H2D of fresh controls (4 MB), "
                                 "evaluation, D2H of costs + gradients + final states, per step")
    # Pade orders the engine chose on this workload (by the 1-norm of the step generators, Higham
    # 2005 Algorithm 2.3 - the thresholds the reference's expm.py carries), and the same step with
    # the order pinned to 13, which is what the reference itself always executes
    orders = engine.pade_orders()
    reference_order = None
    if world == 1 and not args.no_reference_order and not standin:
        engine.set_knob("pade_order", 13)
        one_step()
        engine.synchronize()
        t_r = time.perf_counter()
        for _ in range(args.steps):
            one_step()
        engine.synchronize()
        r13 = (time.perf_counter() - t_r) / args.steps
        engine.set_knob("pade_order", 0)
        one_step()
        reference_order = dict(ms_per_step=r13 * 1e3, value=seeds * (N_EVAL - 1) / r13,
                               unit="propagator-steps/s",
                               note="knob pade_order = 13: every step on the [13/13] approximant, as "
                                    "the reference executes it (expm.py:230-233)")
    secondary = None
    latency = None
    general = None
    if world == 1 and not args.no_secondary and not standin:
        engine.set_timing(True)
        secondary = lindblad_secondary(engine)
        latency = None if args.no_latency else latency_secondary(engine)
        general = None if args.no_latency else general_secondary(engine)
    engine.set_timing(False)

    units_per_step = world * seeds * (N_EVAL - 1)
    value = units_per_step * args.steps / elapsed

    # roofline of the dominant kernel (pade_pq): algorithmic flops = the complex n^3 GEMMs of the
    # Pade chain per propagator step - 2 / 3 / 4 / 5 / 6 products for order 3 / 5 / 7 / 9 / 13, as
    # executed on this workload - at 8 real flops per complex MAC. The LU
    # factorisation fused into the kernel (8/3 n^3 VALU flops) is not counted.
    launches, total_ms = timing["pade_pq"]
    products = {3: 2, 5: 3, 7: 4, 9: 5, 13: 6}
    steps_counted = max(1, sum(orders.values()))
    mean_products = sum(products[o] * c for o, c in orders.items()) / steps_counted
    k1_flops_per_unit = 8.0 * mean_products * DIM ** 3
    roofline = None
    current = counters_current()
    stale_note = ("" if current else " - STALE: the kernel sources have changed since those passes "
                                     "(profiles/r05_profile_meta.json), re-run tools/profile_round.sh")
    if launches > 0 and total_ms > 0:
        avg_s = total_ms / launches * 1e-3
        units_per_launch = seeds * (N_EVAL - 1) * args.steps / launches
        achieved = k1_flops_per_unit * units_per_launch / avg_s / 1e12
        roofline = dict(bound="mfma", kernel=K1A_KERNEL, achieved=achieved,
                        peak=FP64_MFMA_PEAK_TFLOPS, unit="TFLOP/s",
                        frac=achieved / FP64_MFMA_PEAK_TFLOPS,
                        peak_sustained_measured=engine.mfma_peak(2, 20000),
                        peak_sustained_chains=1,
                        peak_sustained_note="register-only loop of v_mfma_f64_16x16x4_f64, ONE "
                                            "accumulation chain per wave, two waves per SIMD "
                                            "(qocx_debug_mfma_peak; eight chains sustain ~48)",
                        traffic=pmc_traffic_bytes(K1A_KERNEL, units_per_launch),
                        traffic_source="committed rocprofv3 --pmc passes of this command "
                                       "({}; FETCH_SIZE x2 + WRITE_SIZE), not collected in "
                                       "this run{}".format(PMC_SUMMARY, stale_note),
                        avg_launch_ms=total_ms / launches,
                        gemm_products_per_step=mean_products,
                        flops_per_step=k1_flops_per_unit,
                        note="FP64 MFMA and FP64 vector instructions share one pipe on this part "
                             "(SQ_VALU_MFMA_COEXEC_CYCLES = 0, profiles/r05_pmc_k1a_two_vs_three.json): "
                             "the kernel's 232 MFMA + ~3 900 vector instructions per step keep that "
                             "pipe busy ~75 % of the launch")
        # what the matrix cores executed (3M scheme, Hermitian tiles, the Schur update of the
        # factorisation), from the committed SQ counter pass: MFMA operations per step x 512 flop
        executed = pmc_executed_mfma_flops(K1A_KERNEL)
        if executed is not None:
            roofline["executed_mfma_flops_per_step"] = executed
            roofline["frac_executed"] = executed * units_per_launch / avg_s / 1e12 / FP64_MFMA_PEAK_TFLOPS
            roofline["frac_executed_source"] = ("SQ_INSTS_VALU_MFMA_MOPS_F64 of the committed rocprofv3 "
                                                "--pmc pass ({}) over this run's launch time{}".format(
                                                    PMC_SQ_SUMMARY, stale_note))
            roofline["frac_executed_current"] = current
    kernel_ms = {k: (v[1] / v[0] if v[0] else 0.0) for k, v in warm_timing.items()}
    kernel_ms["pade_pq"] = total_ms / launches if launches else kernel_ms["pade_pq"]
    # The whole path under the BUILD'S OWN operation count (SURVEY.md 8d's
    # 21.33 n^3 is the count of the reference's formulation - U_j formed and squared, the dense
    # reverse rules of the tape - which this algorithm does not execute): per propagator step
    #   K1a   m_p complex n^3 products of the Pade chain (m_p = 2 / 3 / 4 / 5 / 6 at order 3..13)
    #   K1b   n^3 / 3 complex MACs (LU of P)
    #   K2    forward + adjoint sweep: Q psi and two triangular solves each = 4 n^2 complex MACs
    #   K3    2 (m - 1) matrix-vector products, m rank-1 updates, K contractions = (3 m - 2 + K) n^2
    # at 8 real flops per complex MAC, m = the mean Pade order of the workload.
    mean_order = sum(o * c for o, c in orders.items()) / steps_counted
    path_flops_per_unit = 8.0 * (mean_products * DIM ** 3 + DIM ** 3 / 3.0 + 4 * DIM ** 2
                                 + (3 * mean_order - 2 + K_CTRL) * DIM ** 2)
    path_tflops = path_flops_per_unit * units_per_step * args.steps / elapsed / 1e12 / world
    # HBM traffic of one evaluation (committed PMC passes x this run's launch counts) over this
    # run's time per evaluation
    per_eval_launches = {k: (v[0] / max(1, args.warmup)) for k, v in warm_timing.items()}
    hbm = None
    eval_bytes, eval_parts = pmc_evaluation_bytes(per_eval_launches) if all(
        k in per_eval_launches for k in EVAL_KERNELS) else (None, None)
    if eval_bytes is not None:
        tbs = eval_bytes / (elapsed / args.steps) / 1e12
        hbm = dict(bytes_per_evaluation=eval_bytes, achieved=tbs, peak=8.0, unit="TB/s", frac=tbs / 8.0,
                   algorithmic_bytes_per_evaluation=float(seeds) * (N_EVAL * K_CTRL * 8 * 2 + DIM * 16 + 8),
                   by_kernel=eval_parts,
                   note="Q and the LU factors of every step are written once (K1a) and read twice "
                        "(forward and adjoint sweep): 33.8 KB per step and pass by design; the "
                        "algorithmic bytes are controls in, gradients + final states + costs out",
                   source="committed rocprofv3 --pmc passes ({}) x the launches per evaluation of "
                          "this run{}".format(PMC_SUMMARY, stale_note))

    line = {
        "metric": "propagator-steps/sec (fwd+grad), dim=32 Schroedinger, 1000 steps x 256 seeds",
        "value": value, "unit": "propagator-steps/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic" if not standin else "STAND-IN ENGINE ({}): control-flow rehearsal, no "
                                                "measurement".format(standin),
        "config": {"workload": "configs[2]: dim=32 Schroedinger, 1000 propagator steps, "
                               "{} seeds per GPU, K=2 real controls, S=1, M2".format(seeds),
                   "seeds_per_gpu": seeds, "hilbert_size": DIM, "system_eval_count": N_EVAL,
                   "pade_orders": {str(o): c for o, c in orders.items() if c},
                   "pade_policy": "order by ||dt H||_1 (Higham 2005 alg. 2.3 thresholds, "
                                  "expm.py:194-209); reference_order has the always-13 rate",
                   "parallelism": "seed-sharded x{}".format(world)},
        "roofline": roofline,
        "kernel_ms_per_launch": kernel_ms,
        "kernel_ms_source": "pade_pq: HIP events inside the timed region; the others: HIP events of "
                            "the warm-up steps (all launches timed there)",
        "step_ms": step_ms,
        "flops_per_step_path": path_flops_per_unit,
        "path_tflops_per_gpu": path_tflops,
        "frac_path": path_tflops / FP64_MFMA_PEAK_TFLOPS,
        "frac_path_note": "whole evaluation under the build's own operation count (K1a products + LU "
                          "+ both sweeps + K3, bench.py) over the FP64 peak of the chip (78.6 TFLOP/s, "
                          "vector and matrix alike)",
        "hbm": hbm,
        "check": {"sum_cost": total_cost, "grad_l2": float(np.linalg.norm(total_grad))},
        "value_definition": "controls resident in HBM when the clock starts (bench contract), Pade "
                            "order by norm; value_host_to_host: fresh controls from host memory and "
                            "results back in host memory every step (SURVEY 8d); value_reference_order: "
                            "every step on the [13/13] approximant as the reference executes it",
        "value_host_to_host": host_to_host["value"] if host_to_host else None,
        "value_reference_order": reference_order["value"] if reference_order else None,
        "host_to_host": host_to_host,
        "reference_order": reference_order,
        "secondary": secondary,
        "latency": latency,
        "general_path": general,
    }
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline and not standin:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line))
    engine.close()


if __name__ == "__main__":
    main()
