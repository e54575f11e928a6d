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


def main():
    for case in cases_mod.lindblad_cases():
        t0 = time.time()
        errors, finals, grads, traced = [], [], [], []
        for controls in case.controls:
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
            errors.append(err)
            finals.append(final)
            grads.append(g_frozen)
            traced.append(g_traced)
        np.savez_compressed(os.path.join(GOLDEN, case.name + ".npz"), error=np.array(errors),
                            final_densities=np.stack(finals), grads_ad=np.stack(grads),
                            grads_ad_traced_controller=np.stack(traced), controls=case.controls)
        print("{:24s} errors {} ({:.1f}s)".format(case.name, errors, time.time() - t0))
    # the reference's analytic known answers (tests/test_core.py:82-148) are asserted directly in
    # tests/test_lindblad_oracle.py; no fixture needed.


if __name__ == "__main__":
    main()
