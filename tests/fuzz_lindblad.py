"""
fuzz_lindblad.py - TEST TOOLING (parity checker; lives under tests/ because it drives the oracle): random Lindblad problem shapes, engine (through the C ABI)
against the NumPy model of the device algorithm (tests/lindblad_model.py), which the CPU suite
holds against the reference fixtures.

    python -m tests.fuzz_lindblad [count] [seed] [lmin lmax]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import qoc_lindblad_numpy as ol  # noqa: E402
from qoc_amd.engine import Engine, QocxError, COST_FORBID_DENSITY, COST_TARGET_DENSITY  # noqa: E402
from tests import lindblad_model as lm  # noqa: E402
from tests.cases import gue, random_density  # noqa: E402


def one(engine, rng, index, lmin=0, lmax=3):
    n = int(rng.integers(1, 33)) if rng.random() < 0.5 else int(rng.integers(1, 17))
    S = int(rng.integers(1, 4))
    K = int(rng.integers(0, 4))
    L = int(rng.integers(lmin, lmax + 1))
    N = int(rng.integers(2, 6))
    Nc = int(rng.integers(2, 9)) if K else 0
    ces = int(rng.integers(1, 3))
    batch = int(rng.integers(1, 4))
    h0 = gue(rng, n) * float(10 ** rng.uniform(-0.3, 0.5))
    g = [gue(rng, n) for _ in range(K)]
    ops = np.stack([gue(rng, n) + 0.5j * gue(rng, n) for _ in range(L)]) if L else None
    gam = rng.uniform(0.02, 0.4, L) if L else None
    rho0 = np.stack([random_density(rng, n) for _ in range(S)])
    targ = np.stack([random_density(rng, n) for _ in range(S)])
    forb = np.stack([random_density(rng, n) for _ in range(2 * S)])
    T = float(10 ** rng.uniform(-1.2, -0.2)) * (N - 1)
    count = (N - 1) // ces
    descs = [dict(kind=COST_TARGET_DENSITY, step_cost=0, scale=0.8, vectors=targ)]
    costs = [ol.TargetDensityInfidelity(targ, cost_multiplier=0.8)]
    if count > 0:
        descs.append(dict(kind=COST_FORBID_DENSITY, step_cost=1, scale=1.5 / (count * S),
                          vectors=forb, counts=[2] * S))
        costs.append(ol.ForbidDensities(forb.reshape(S, 2, n, n), N, cost_eval_step=ces,
                                        cost_multiplier=1.5))
    tag = "n={} S={} K={} L={} N={} Nc={} ces={} B={}".format(n, S, K, L, N, Nc, ces, batch)
    try:
        engine.set_lindblad_problem(n, S, K, Nc, N, T, h0, g, gam, ops, rho0, costs=descs,
                                    cost_eval_step=ces)
    except QocxError as exc:
        if L > 4 and "LDS" in exc.message:  # 17 <= n <= 32: the operator images of more than five do not fit
            return 0.0, tag + " (rejected: operators beyond the LDS, not compared)"
        raise
    controls = (float(10 ** rng.uniform(-1, 0.3)) * rng.standard_normal((batch, Nc, K))
                if K else None)
    cost, grads, final = engine.evaluate_lindblad(controls if K else batch, want_grad=K > 0)
    system = lm.StructuredLindblad(h0, g, gam, ops)
    worst = 0.0
    for b in range(batch):
        u = controls[b] if K else np.zeros((2, 0))
        m_err, m_grads, m_final = lm.evaluate_with_grad(system, u, rho0, T, N, costs, ces,
                                                        want_grad=K > 0)
        worst = max(worst, abs(cost[b] - m_err) / 1e-11, np.max(np.abs(final[b] - m_final)) / 1e-11)
        if K:
            worst = max(worst, np.max(np.abs(grads[b] - m_grads))
                        / (1e-9 * max(np.max(np.abs(m_grads)), 1e-3)))
    tag = "n={} S={} K={} L={} N={} Nc={} ces={} B={}".format(n, S, K, L, N, Nc, ces, batch)
    return worst, tag


def one_two_sided(engine, rng, index):
    """Shapes of the two-sided evaluation at n <= 16 (one final target cost: forward pass || unit adjoint,
    the chain form of the stage loop): 1 .. 4 Lindblad operators, complex or real, 1 .. 3 densities, 1 .. 4
    controls, control magnitudes that mix sub-division counts inside a batch."""
    n = int(rng.integers(2, 17))
    S = int(rng.integers(1, 4))
    K = int(rng.integers(1, 5))
    L = int(rng.integers(1, 5))
    N = int(rng.integers(2, 8))
    Nc = int(rng.integers(2, 9))
    batch = int(rng.integers(1, 5))
    h0 = gue(rng, n) * float(10 ** rng.uniform(-0.3, 0.5))
    g = [gue(rng, n) for _ in range(K)]
    if rng.random() < 0.5:
        ops = np.stack([rng.standard_normal((n, n)) / np.sqrt(n) + 0j for _ in range(L)])
    else:
        ops = np.stack([gue(rng, n) + 0.5j * gue(rng, n) for _ in range(L)])
    gam = rng.uniform(0.02, 0.4, L)
    rho0 = np.stack([random_density(rng, n) for _ in range(S)])
    targ = np.stack([random_density(rng, n) for _ in range(S)])
    T = float(10 ** rng.uniform(-1.2, -0.2)) * (N - 1)
    descs = [dict(kind=COST_TARGET_DENSITY, step_cost=0, scale=0.8, vectors=targ)]
    costs = [ol.TargetDensityInfidelity(targ, cost_multiplier=0.8)]
    engine.set_lindblad_problem(n, S, K, Nc, N, T, h0, g, gam, ops, rho0, costs=descs)
    controls = (10 ** rng.uniform(-1, 0.3, (batch, 1, 1))) * rng.standard_normal((batch, Nc, K))
    cost, grads, final = engine.evaluate_lindblad(controls, want_grad=True)
    system = lm.StructuredLindblad(h0, g, gam, ops)
    tag = "two-sided n={} S={} K={} L={} N={} Nc={} B={} real_ops={}".format(
        n, S, K, L, N, Nc, batch, bool(np.all(ops.imag == 0)))
    # The sub-division count is a ceil of (norm bound x step / 0.4): the engine's bound comes from a power
    # iteration, the model's from the exact norm - on the rare case that falls between the two they
    # integrate on different meshes (both right to the integrator's own ~1e-10) and there is nothing to compare.
    expected = 0
    for b in range(batch):
        grid = lm.substep_grid(T, N, Nc, system.norm_bound(np.max(np.abs(controls[b]), axis=0)))
        expected += sum(len(step) for step in grid)
    if expected != engine.lindblad_last_subintervals():
        return 0.0, tag + " (meshes differ: not compared)"
    worst = 0.0
    for b in range(batch):
        m_err, m_grads, m_final = lm.evaluate_with_grad(system, controls[b], rho0, T, N, costs, 1, want_grad=True)
        worst = max(worst, abs(cost[b] - m_err) / 1e-11, np.max(np.abs(final[b] - m_final)) / 1e-11)
        worst = max(worst, np.max(np.abs(grads[b] - m_grads)) / (1e-9 * max(np.max(np.abs(m_grads)), 1e-3)))
    return worst, tag


def main():
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    lmin = int(sys.argv[3]) if len(sys.argv) > 3 else None  # [lmin lmax]: operator counts of one()
    lmax = int(sys.argv[4]) if len(sys.argv) > 4 else lmin
    rng = np.random.default_rng(seed)
    engine = Engine(0)
    bad, skipped, overall = 0, 0, 0.0
    for index in range(count):
        # (every other case from the two-sided shapes)
        if lmin is not None:
            worst, tag = one(engine, rng, index, lmin, lmax)
        else:
            worst, tag = (one_two_sided if index % 2 else one)(engine, rng, index)
        overall = max(overall, worst)
        skipped += tag.endswith("not compared)")
        if worst > 1.0:
            bad += 1
            print("FAIL x{:.2f} of tolerance: {}".format(worst, tag), flush=True)
    print("{} cases ({} on meshes that differ from the model's: not compared), {} failures, worst {:.2e} of "
          "tolerance (cost/densities 1e-11, grads 1e-9)".format(count, skipped, bad, overall))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
