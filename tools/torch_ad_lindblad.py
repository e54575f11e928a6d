"""
torch_ad_lindblad.py - BUILD-CONTAINER TOOLING (never imported by the product, the tests or
bench.py): reverse-mode AD (PyTorch CPU complex128) over a restatement of the reference's
discrete Lindblad evolve loop INCLUDING its adaptive RKDP5(4) integrator. Step sizes, mesh
positions and the dense-output abscissa stay in the graph, exactly as they are traced by
autograd in the reference (qoc/core/mathmethods.py:352-480 uses autograd.numpy throughout);
only the accept/reject comparisons are decisions. HIPS autograd is not installed here, and the
reference pins no gradient value, so this plays the role of its AD for the golden vectors.
"""

import os

import numpy as np
import torch

import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "refstubs"))
sys.path.insert(0, "/root/reference")
from qoc.core import mathmethods as ol  # noqa: E402  the REFERENCE's own RKDP5(4) tableau

ol.P_ORDER = ol.P  # (the name this file uses for the order)

C = torch.complex128


def interp(x, xs, ys):
    xf = float(x)
    if xf <= xs[0]:
        i1, i2 = 0, 1
    elif xf >= xs[-1]:
        i1, i2 = len(xs) - 2, len(xs) - 1
    else:
        i2 = int(np.argmax(xf <= xs))
        i1 = i2 - 1
    return ys[i1] + (((ys[i2] - ys[i1]) / (xs[i2] - xs[i1])) * (x - xs[i1]))


def hconj(x):
    return torch.conj(x.transpose(-1, -2))


def rms_norm(a):
    return torch.sqrt(torch.sum(a * torch.conj(a)) / a.numel())


def ad_eval(case, controls_np, freeze_mesh=False, atol=1e-12):
    # (atol: the reference's integrate_rkdp5 default is 1e-12, mathmethods.py:353; a tighter value is
    # used ONLY to show which of two gradients the integrator converges to - fixture grads_ad_tight)
    h0 = torch.tensor(case.h0, dtype=C)
    g_re = [torch.tensor(g, dtype=C) for g in case.g_re]
    g_im = [torch.tensor(g, dtype=C) for g in case.g_im] if case.g_im is not None else None
    gam = torch.tensor(case.dissipators)
    ops = torch.tensor(case.operators, dtype=C)
    ops_d = hconj(ops)
    ops_p = torch.matmul(ops_d, ops)
    controls = torch.tensor(controls_np, requires_grad=True)
    xs = np.linspace(0, case.T, case.Nc)
    dt = case.T / (case.N - 1)

    def rhs(t, rho):
        u = interp(t, xs, controls)
        h = h0
        if getattr(case, "time_mod", None) is not None:  # tests/cases.py Case.hamiltonian
            h = h0 * (1 + 0.3 * torch.cos(case.time_mod * torch.as_tensor(t)))
        for k in range(case.K):
            if case.complex_controls:
                h = h + torch.real(u[k]) * g_re[k] + torch.imag(u[k]) * g_im[k]
            else:
                h = h + u[k] * g_re[k]
            if getattr(case, "quad", None) is not None:  # tests/cases.py Case.hamiltonian
                mod2 = torch.real(u[k]) ** 2 + torch.imag(u[k]) ** 2 if case.complex_controls \
                    else u[k] ** 2
                h = h + mod2 * torch.tensor(case.quad[k], dtype=C)
        out = -1j * (torch.matmul(h, rho) - torch.matmul(rho, h))
        gs, os_, osd, osp = gam, ops, ops_d, ops_p
        if getattr(case, "data_mod", None) is not None:  # tests/cases.py LindbladCase.lindblad_data
            wg, wo = case.data_mod
            tt = torch.as_tensor(t)
            gs = gam * (1 + 0.5 * torch.sin(wg * tt))
            fo = 1 + 0.2 * torch.cos(wo * tt)
            os_, osd, osp = ops * fo, ops_d * fo, ops_p * (fo * fo)
        for i in range(ops.shape[0]):
            out = out + gs[i] * (torch.matmul(torch.matmul(os_[i], rho), osd[i])
                                 - 0.5 * torch.matmul(osp[i], rho)
                                 - 0.5 * torch.matmul(rho, osp[i]))
        return out

    def rk_step(h, x0, y0, k1):
        k2 = rhs(x0 + ol.C2 * h, y0 + h * ol.A21 * k1)
        k3 = rhs(x0 + ol.C3 * h, y0 + h * (ol.A31 * k1 + ol.A32 * k2))
        k4 = rhs(x0 + ol.C4 * h, y0 + h * (ol.A41 * k1 + ol.A42 * k2 + ol.A43 * k3))
        k5 = rhs(x0 + ol.C5 * h, y0 + h * (ol.A51 * k1 + ol.A52 * k2 + ol.A53 * k3 + ol.A54 * k4))
        k6 = rhs(x0 + h, y0 + h * (ol.A61 * k1 + ol.A62 * k2 + ol.A63 * k3 + ol.A64 * k4
                                   + ol.A65 * k5))
        y1 = y0 + h * (ol.B1 * k1 + ol.B3 * k3 + ol.B4 * k4 + ol.B5 * k5 + ol.B6 * k6)
        k7 = rhs(x0 + h, y1)
        y1h = y0 + h * (ol.B1H * k1 + ol.B3H * k3 + ol.B4H * k4 + ol.B5H * k5 + ol.B6H * k6
                        + ol.B7H * k7)
        return (k1, k2, k3, k4, k5, k6, k7), y1, y1h

    def integrate(x_final, x_initial, y_initial):
        f0 = rhs(x_initial, y_initial)
        d0, d1 = rms_norm(y_initial), rms_norm(f0)
        if float(d0.real) < 1e-5 or float(d1.real) < 1e-5:
            h0_ = torch.tensor(1e-6)
        else:
            h0_ = torch.real(0.01 * d0 / d1)
        y1 = y_initial + h0_ * f0
        f1 = rhs(x_initial + h0_, y1)
        d2 = rms_norm(f1 - f0) / h0_
        big = torch.maximum(torch.real(d1), torch.real(d2))
        if float(big) <= 1e-15:
            h1 = torch.maximum(torch.tensor(1e-6), h0_ * 1e-3)
        else:
            h1 = torch.pow(0.01 / big, 1 / (ol.P_ORDER + 1))
        step = torch.minimum(100 * h0_, h1)
        if freeze_mesh:
            step = step.detach()
        x_cur, y_cur, k1 = x_initial, y_initial, f0
        result = None
        while float(x_cur) <= float(x_final):
            rejected, accepted = False, False
            while not accepted:
                ks, y1, y1h = rk_step(step, x_cur, y_cur, k1)
                x_new = x_cur + step
                err = torch.real(rms_norm((y1 - y1h) / atol))
                if float(err) < 1:
                    accepted = True
                    if float(err) == 0:
                        factor = torch.tensor(10.0)
                    else:
                        factor = torch.minimum(torch.tensor(10.0),
                                               0.9 * torch.pow(err, float(ol.ERROR_EXP)))
                    if rejected:
                        factor = torch.minimum(torch.tensor(1.0), factor)
                    step = step * (factor.detach() if freeze_mesh else factor)
                else:
                    rejected = True
                    factor = torch.maximum(torch.tensor(0.2),
                                           0.9 * torch.pow(err, float(ol.ERROR_EXP)))
                    step = step * (factor.detach() if freeze_mesh else factor)
            if float(x_cur) <= float(x_final) <= float(x_new):
                hh = x_new - x_cur
                r1, r2 = y_cur, y1 - y_cur
                r3 = y_cur + hh * ks[0] - y1
                r4 = 2 * (y1 - y_cur) - hh * (ks[0] + ks[6])
                r5 = hh * (ol.D1 * ks[0] + ol.D3 * ks[2] + ol.D4 * ks[3] + ol.D5 * ks[4]
                           + ol.D6 * ks[5] + ol.D7 * ks[6])
                th = (x_final - x_cur) / hh
                th2, th3 = th ** 2, th ** 3
                th4 = th2 ** 2
                result = (r1 + th * (r2 + r3) - th2 * (r3 - r4 - r5) - th3 * (r4 + 2 * r5)
                          + th4 * r5)
            x_cur, y_cur, k1 = x_new, y1, ks[6]
        return result

    def cost_value(spec, rho):
        kind, kw = spec
        m = kw.get("cost_multiplier", 1.)
        if kind in ("TargetDensityInfidelity", "TargetDensityInfidelityTime"):
            t = torch.tensor(np.stack(kw["target_densities"]), dtype=C)
            prods = torch.matmul(hconj(t), rho)
            fid = 0
            for p in prods:
                fid = fid + torch.abs(torch.trace(p))
            val = 1 - fid / (t.shape[0] * t.shape[1])
            if kind == "TargetDensityInfidelityTime":
                val = val / ((kw["system_eval_count"] - 1) // kw.get("cost_eval_step", 1))
            return val * m
        forb = kw["forbidden_densities"]
        n = forb.shape[3]
        count = (kw["system_eval_count"] - 1) // kw.get("cost_eval_step", 1)
        total = 0
        for i, fs in enumerate(forb):
            dc = 0
            for f in fs:
                ip = torch.trace(torch.matmul(hconj(torch.tensor(f, dtype=C)), rho[i])) / n
                dc = dc + torch.real(ip * torch.conj(ip))
            total = total + dc / len(fs)
        return total / (count * len(forb)) * m

    rho = torch.tensor(case.initial_densities, dtype=C)
    error = 0
    for step in range(case.N):
        if step % case.cost_eval_step == 0 and step != 0:
            for spec in case.cost_specs:
                if spec[0] == "ForbidDensities":
                    error = error + cost_value(spec, rho)
        if step != case.N - 1:
            t0 = step * dt
            rho = integrate(torch.tensor(t0 + dt), torch.tensor(t0), rho)
    for spec in case.cost_specs:
        if spec[0] != "ForbidDensities":
            error = error + cost_value(spec, rho)
    error.backward()
    return float(error), controls.grad.numpy().copy(), rho.detach().numpy().copy()
