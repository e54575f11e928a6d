"""
gen_golden.py - BUILD-CONTAINER TOOLING: mint tests/golden/*.npz.

Imports the REFERENCE (/root/reference, read-only) under the forward-only stand-ins of
tools/refstubs and runs its own, unmodified, forward path on the seeded problems of
tests/cases.py.  Gradients cannot come from the reference here (HIPS autograd is absent),
so every gradient fixture is produced twice, independently:
  * `grads_ad`  - reverse-mode AD of the same op sequence (tools/torch_ad.py);
  * `grads_fd`  - Richardson-extrapolated central differences of the REFERENCE forward's
                  `.error`, on a random subset of control entries (`fd_index`).
The script asserts the two agree before writing anything.

Only the resulting .npz data files travel; nothing here runs on the GPU box.

    python tools/gen_golden.py            # all cases
    python tools/gen_golden.py c2_random  # one case
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

import qoc  # noqa: E402  (the reference)
from qoc.core.common import clip_control_norms, slap_controls, strip_controls  # noqa: E402
from qoc.core.mathmethods import (interpolate_linear_set, magnus_m2, magnus_m4,  # noqa: E402
                                  magnus_m6)
from qoc.models import MagnusPolicy  # noqa: E402
from qoc.standard import costs as ref_costs  # noqa: E402
from qoc.standard.functions.expm import expm_pade  # noqa: E402
from qoc.standard.optimizers.adam import Adam  # noqa: E402

from tests import cases as cases_mod  # noqa: E402
import torch_ad  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
POLICY = {"M2": MagnusPolicy.M2, "M4": MagnusPolicy.M4, "M6": MagnusPolicy.M6}


def ref_costs_for(case):
    out = []
    for kind, kw in case.cost_specs:
        out.append(getattr(ref_costs, kind)(**kw))
    return out


def ref_forward(case, controls):
    result = qoc.evolve_schroedinger_discrete(
        case.T, case.hamiltonian(), case.initial_states, case.N,
        controls=controls, cost_eval_step=case.cost_eval_step, costs=ref_costs_for(case),
        magnus_policy=POLICY[case.magnus])
    return result.error, np.asarray(result.final_states)


def richardson_fd(case, controls, index, h=1e-3):
    """d error / d(Re, Im) of controls.flat[index] from the reference forward."""
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


def do_case(case, fd_count):
    t0 = time.time()
    out = dict(name=case.name)
    if case.controls is None:
        err, final = ref_forward(case, None)
        out.update(error=np.array([err]), final_states=final[None])
        np.savez_compressed(os.path.join(GOLDEN, case.name + ".npz"), **out)
        print("{:22s} forward only  ({:.1f}s)".format(case.name, time.time() - t0))
        return
    errors, finals, grads_ad, fd_index, grads_fd = [], [], [], [], []
    for b, controls in enumerate(case.controls):
        err, final = ref_forward(case, controls)
        err_ad, g_ad, final_ad = torch_ad.ad_eval(case, controls)
        assert abs(err - err_ad) <= 1e-12 * max(1, abs(err)), (case.name, err, err_ad)
        assert np.allclose(final, final_ad, rtol=0, atol=1e-11), case.name
        rng = np.random.default_rng(9000 + b)
        idx = rng.choice(controls.size, size=min(fd_count, controls.size), replace=False)
        g_fd = np.array([richardson_fd(case, controls, i) for i in idx])
        scale = np.max(np.abs(g_ad))
        dev = np.max(np.abs(g_fd - g_ad.flat[idx])) / scale
        assert dev < 1e-7, (case.name, b, dev)
        errors.append(err)
        finals.append(final)
        grads_ad.append(g_ad)
        fd_index.append(idx)
        grads_fd.append(g_fd)
        print("{:22s} seed {} error {:.12f}  AD-vs-FD(ref fwd) rel {:.2e}".format(
            case.name, b, err, dev))
    out.update(error=np.array(errors), final_states=np.stack(finals),
               grads_ad=np.stack(grads_ad), fd_index=np.stack(fd_index),
               grads_fd=np.stack(grads_fd), controls=case.controls)
    np.savez_compressed(os.path.join(GOLDEN, case.name + ".npz"), **out)
    print("{:22s} done ({:.1f}s)".format(case.name, time.time() - t0))


def do_units():
    """Unit-level vectors from the reference's own functions."""
    rng = np.random.default_rng(123)
    out = {}
    # expm_pade: Hermitian-generated and general complex inputs, norms on both sides of
    # theta_13 (s = 0 .. 4), sizes 2 .. 32.
    mats, exps = [], []
    for i, (n, scale, skew) in enumerate([(2, 0.3, True), (4, 1.0, True), (8, 3.0, True),
                                          (8, 7.0, True), (8, 40.0, False), (16, 2.0, False),
                                          (32, 1.0, True), (32, 12.0, True), (32, 80.0, False)]):
        g = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
        if skew:
            a = -1j * (g + g.conj().T) / 2
        else:
            a = g
        a = a * (scale / np.max(np.sum(np.abs(a), axis=0)))
        out["expm_in_%d" % i] = a
        out["expm_out_%d" % i] = expm_pade(a)
    out["expm_count"] = np.array(9)
    # interpolation incl. both extrapolation branches (mathmethods.py:54-59)
    xs = np.linspace(0, 2.0, 9)
    ys = rng.standard_normal((9, 3)) + 1j * rng.standard_normal((9, 3))
    xq = np.array([-0.3, 0.0, 0.1, 0.25, 0.26, 1.0, 1.99, 2.0, 2.4])
    out["interp_xs"], out["interp_ys"], out["interp_xq"] = xs, ys, xq
    out["interp_out"] = np.stack([interpolate_linear_set(x, xs, ys) for x in xq])
    # Magnus with an explicitly time-dependent, non-commuting generator
    m0 = rng.standard_normal((5, 5)) + 1j * rng.standard_normal((5, 5))
    m1 = rng.standard_normal((5, 5)) + 1j * rng.standard_normal((5, 5))
    m2 = rng.standard_normal((5, 5)) + 1j * rng.standard_normal((5, 5))
    gen = lambda t: m0 + t * m1 + np.sin(3 * t) * m2
    out["magnus_m0"], out["magnus_m1"], out["magnus_m2"] = m0, m1, m2
    out["magnus_dt"], out["magnus_t"] = np.array(0.37), np.array(1.1)
    out["magnus_out_M2"] = magnus_m2(gen, 0.37, 1.1)
    out["magnus_out_M4"] = magnus_m4(gen, 0.37, 1.1)
    out["magnus_out_M6"] = magnus_m6(gen, 0.37, 1.1)
    # clip / strip / slap (common.py)
    cr = rng.standard_normal((6, 2)) * 2
    cc = (rng.standard_normal((6, 2)) + 1j * rng.standard_normal((6, 2))) * 2
    norms = np.array([1.0, 2.5])
    out["clip_norms"] = norms
    out["clip_in_real"], out["clip_in_complex"] = cr.copy(), cc.copy()
    clip_control_norms(cr, norms)
    clip_control_norms(cc, norms)
    out["clip_out_real"], out["clip_out_complex"] = cr, cc
    out["strip_complex"] = strip_controls(True, out["clip_in_complex"])
    out["slap_complex"] = slap_controls(True, out["strip_complex"], (6, 2))
    # Adam trajectory with every knob on (adam.py:110-165)
    adam = Adam(learning_rate=0.05, learning_rate_decay=7.0, clip_grads=0.6, scale_grads=1.5)
    params = rng.standard_normal(10)
    adam.run(None, 0, params, None, None)
    traj, gs = [params], []
    for _ in range(5):
        g = rng.standard_normal(10)
        gs.append(g)
        traj.append(adam.update(g, traj[-1]))
    out["adam_grads"], out["adam_traj"] = np.stack(gs), np.stack(traj)
    # cost values on random inputs, every cost class in its Schroedinger flavour
    np.savez_compressed(os.path.join(GOLDEN, "units.npz"), **out)
    print("units done")


def main():
    os.makedirs(GOLDEN, exist_ok=True)
    only = sys.argv[1:]
    if not only or "units" in only:
        do_units()
    for case in cases_mod.all_cases() + cases_mod.opaque_cases():
        if only and case.name not in only:
            continue
        big = case.n * case.N > 4000
        do_case(case, fd_count=6 if big else 24)


if __name__ == "__main__":
    main()
