"""
GPU test (-m gpu): one engine context re-used for a long run of problems of changing size (all three
tile counts), state count, control count, Magnus policy and batch size, Schroedinger and Lindblad in
turn - buffers are re-sized up and down, kernels of every family alternate. Results stay finite and
the host process does not grow.
"""

import numpy as np
import pytest

from tests import cases as cm

pytestmark = pytest.mark.gpu


def _rss_mb():
    for line in open("/proc/self/status"):
        if line.startswith("VmRSS"):
            return int(line.split()[1]) // 1024
    return 0


def test_one_context_many_problems():
    from qoc_amd.engine import Engine, COST_TARGET_COHERENT, COST_TARGET_DENSITY
    rng = np.random.default_rng(1)
    engine = Engine(0)
    at_40 = None
    try:
        for it in range(240):
            n = [3, 8, 16, 20, 32, 40, 64][it % 7]
            S, K, N = 1 + it % 3, 1 + it % 3, 20 + (it % 5) * 7
            B = [1, 5, 33, 128][it % 4]
            h0, g = cm.gue(rng, n), [cm.gue(rng, n) for _ in range(K)]
            psi = np.eye(n, dtype=complex)[:S]
            target = np.eye(n, dtype=complex)[::-1][:S]
            engine.set_schroedinger_problem(
                n, S, K, N, N, 0.05 * (N - 1), h0[None], np.stack(g)[None], psi,
                costs=[dict(kind=COST_TARGET_COHERENT, step_cost=0, scale=1.0, vectors=target)],
                magnus_policy=["M2", "M4", "M6"][it % 3])
            cost, grads, _ = engine.evaluate(0.3 * rng.standard_normal((B, N, K)), want_grad=True)
            assert np.all(np.isfinite(cost)) and np.all(np.isfinite(grads))
            assert np.all(cost > -1e-12) and np.all(cost < 1 + 1e-12)
            if it % 6 == 0 and n <= 16:
                a = np.diag(np.sqrt(np.arange(1, n)), 1).astype(complex)
                rho0 = np.zeros((1, n, n), complex)
                rho0[0, 0, 0] = 1
                rho1 = np.zeros((1, n, n), complex)
                rho1[0, -1, -1] = 1
                engine.set_lindblad_problem(
                    n, 1, K, N, N, 0.05 * (N - 1), h0, g, np.array([0.05]), a[None], rho0,
                    costs=[dict(kind=COST_TARGET_DENSITY, step_cost=0, scale=1.0, vectors=rho1)])
                cost, grads, final = engine.evaluate_lindblad(0.3 * rng.standard_normal((B, N, K)))
                assert np.all(np.isfinite(cost)) and np.all(np.isfinite(grads))
                assert np.max(np.abs(np.trace(final[:, 0], axis1=-2, axis2=-1) - 1)) < 1e-10
            if it == 40:
                at_40 = _rss_mb()
        assert _rss_mb() <= at_40 + 64  # no growth with the number of evaluations
    finally:
        engine.close()
