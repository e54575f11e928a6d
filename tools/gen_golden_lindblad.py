"""
gen_golden_lindblad.py - BUILD-CONTAINER TOOLING: mint tests/golden/lindblad_*.npz.

Forward values come from the REFERENCE's own evolve_lindblad_discrete (imported under the
forward-only stand-ins of tools/refstubs); gradients from tools/torch_ad_lindblad.py (AD of a
restatement of the same adaptive integrator, checked here against the reference forward to
1e-9, the reproducibility of the adaptive mesh). Only the .npz files travel.
"""

import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, "refstubs"))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import qoc  # noqa: E402
from qoc.standard import costs as ref_costs  # noqa: E402

from tests import cases as cases_mod  # noqa: E402
import torch_ad_lindblad  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def ref_forward(case, controls):
    costs = [getattr(ref_costs, k)(**kw) for k, kw in case.cost_specs]
    r = qoc.evolve_lindblad_discrete(case.T, case.initial_densities, case.N, controls=controls,
                                     cost_eval_step=case.cost_eval_step, costs=costs,
                                     hamiltonian=case.hamiltonian(),
                                     lindblad_data=case.lindblad_data())
    return r.error, np.asarray(r.final_densities)


def richardson_fd(case, controls, index, h):
    """d error / d(Re, Im) of controls.flat[index] from the REFERENCE's own forward
    (qoc.evolve_lindblad_discrete(...).error), Richardson-extrapolated central differences: the
    same cross-check tools/gen_golden.py applies to the Schroedinger fixtures."""
    def f(c):
        return ref_forward(case, c)[0]

    def central(step, direction):
        cp = controls.copy()
        cm = controls.copy()
        cp.flat[index] += step * direction
        cm.flat[index] -= step * direction
        return (f(cp) - f(cm)) / (2 * step)

    def rich(direction):
        return (4 * central(h / 2, direction) - central(h, direction)) / 3

    g = rich(1.0)
    if np.iscomplexobj(controls):
        g = g + 1j * rich(1.0j)
    return g


# AD (mesh frozen) against finite differences of the reference forward. The forward is an
# adaptive integration that reproduces itself to ~1e-10 (more over 500 steps), so a difference
# quotient with step h carries noise ~1e-10 / h: the gate is 2e-7 relative where the gradient
# is large (the well-conditioned fixtures), and the measured number is printed and stored.
FD_COUNT = {"lindblad_c4_full": 2, "lindblad_bench_c4": 2, "lindblad_wc_c4": 3}
FD_STEP = {"lindblad_wc_n4": 4e-2, "lindblad_wc_n16": 4e-2, "lindblad_timedep_data": 4e-2,
           "lindblad_bench_c4": 4e-2, "lindblad_opaque_wc": 1.5e-2, "lindblad_wc_c4": 4e-2,
           "lindblad_wc_l3": 4e-2, "lindblad_wc_l4": 4e-2}


def main(only=None):
    for case in (cases_mod.lindblad_cases() + cases_mod.lindblad_extra_cases()
                 + cases_mod.lindblad_opaque_grad_cases()):
        if only and case.name not in only:
            continue
        t0 = time.time()
        errors, finals, grads, traced, fd_index, grads_fd, tight = [], [], [], [], [], [], []
        for b, controls in enumerate(case.controls):
            err, final = ref_forward(case, controls)
            # AD with the mesh frozen (step sizes are constants of the tape): the gradient of the
            # discrete scheme on the mesh the reference chose
            err_ad, g_frozen, final_ad = torch_ad_lindblad.ad_eval(case, controls, freeze_mesh=True)
            # AD that also differentiates the step-size controller, as autograd does in the
            # reference (kept for the record: it carries an O(1e-3) relative spurious component)
            _, g_traced, _ = torch_ad_lindblad.ad_eval(case, controls, freeze_mesh=False)
            # the adaptive mesh amplifies rounding-level differences (NumPy vs PyTorch matmul)
            # to the integrator's own accuracy: that is the reproducibility of the reference's
            # result, and the floor of any parity claim for this path
            dev = max(abs(err - err_ad), float(np.max(np.abs(final - final_ad))))
            assert dev < 1e-8, (case.name, dev)
            print("  forward reproducibility {:.1e}; traced-vs-frozen gradient {:.1e} rel".format(
                dev, np.max(np.abs(g_traced - g_frozen)) / np.max(np.abs(g_frozen))))
            rng = np.random.default_rng(9100 + b)
            count = min(FD_COUNT.get(case.name, 4), controls.size)
            idx = rng.choice(controls.size, size=count, replace=False)
            g_fd = np.array([richardson_fd(case, controls, i, FD_STEP.get(case.name, 2e-2))
                             for i in idx])
            scale = np.max(np.abs(g_frozen))
            fd_dev = np.max(np.abs(g_fd - g_frozen.flat[idx])) / scale
            fd_gate = 2e-7 if getattr(case, "grad_rtol", None) else 1e-5
            print("  AD(frozen mesh)-vs-FD(reference forward) {:.1e} rel of max|g| = {:.2e}".format(
                fd_dev, scale))
            # absolute floor: the noise of a difference quotient of the adaptive forward
            # (~1e-10 reproducibility / h), which dominates where max|g| is 1e-5 .. 1e-4
            assert fd_dev * scale < max(fd_gate * scale, 1e-9), (case.name, b, fd_dev)
            if getattr(case, "tight_atol", None):
                # the same integrator with a tighter local tolerance (NOT what the reference's entry
                # point runs): over 500 system steps the default atol = 1e-12 leaves ~1e-8 relative in
                # the gradient; this shows what it converges to
                _, g_tight, _ = torch_ad_lindblad.ad_eval(case, controls, freeze_mesh=True,
                                                           atol=case.tight_atol)
                print("  frozen-mesh AD at atol {:.0e} vs the reference's atol 1e-12: {:.1e} rel".format(
                    case.tight_atol, np.max(np.abs(g_tight - g_frozen)) / scale))
                tight.append(g_tight)
            errors.append(err)
            finals.append(final)
            grads.append(g_frozen)
            traced.append(g_traced)
            fd_index.append(idx)
            grads_fd.append(g_fd)
        extra = dict(grads_ad_tight=np.stack(tight)) if tight else {}
        np.savez_compressed(os.path.join(GOLDEN, case.name + ".npz"), error=np.array(errors),
                            final_densities=np.stack(finals), grads_ad=np.stack(grads),
                            grads_ad_traced_controller=np.stack(traced), controls=case.controls,
                            fd_index=np.stack(fd_index), grads_fd=np.stack(grads_fd), **extra)
        print("{:24s} errors {} ({:.1f}s)".format(case.name, errors, time.time() - t0))
    # Hamiltonians that are not linear in the controls: forward only (errors, final densities)
    for case in cases_mod.lindblad_opaque_cases():
        if only and case.name not in only:
            continue
        out = [ref_forward(case, controls) for controls in case.controls]
        np.savez_compressed(os.path.join(GOLDEN, case.name + ".npz"),
                            error=np.array([o[0] for o in out]),
                            final_densities=np.stack([o[1] for o in out]), controls=case.controls)
        print("{:24s} errors {}".format(case.name, [o[0] for o in out]))
    # the reference's analytic known answers (tests/test_core.py:82-148) are asserted directly in
    # tests/test_lindblad_oracle.py; no fixture needed.


if __name__ == "__main__":
    main(only=set(sys.argv[1:]))
