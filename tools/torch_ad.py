"""
torch_ad.py - BUILD-CONTAINER TOOLING (never imported by the product, the tests or bench.py).

An independent reverse-mode AD of the reference's op sequence for the discrete Schroedinger
evolve loop, written against PyTorch-CPU complex128.  It exists because HIPS autograd (the
reference's AD engine) is not installed in this image and no reference test pins a gradient
value.  tools/gen_golden.py stores its gradients as golden vectors
and cross-checks them against Richardson finite differences of the REFERENCE forward.

PyTorch's complex gradient convention for a real loss, grad = dL/dRe + i dL/dIm, equals
qoc's convention after its conjugation step (qoc/core/schroedingerdiscrete.py:320-324).
"""

import numpy as np
import torch

B = (64764752532480000, 32382376266240000, 7771770303897600, 1187353796428800,
     129060195264000, 10559470521600, 670442572800, 33522128640, 1323241920,
     40840800, 960960, 16380, 182, 1)
THETA_13 = 5.371920351148152

C = torch.complex128


def expm_pade(a):
    n = a.shape[0]
    norm1 = float(torch.max(torch.sum(torch.abs(a), dim=0)))
    s = 0
    if not norm1 < THETA_13:
        s = max(0, int(np.ceil(np.log2(norm1 / THETA_13))))
        a = a * (2 ** -s)
    ident = torch.eye(n, dtype=C)
    a2 = a @ a
    a4 = a2 @ a2
    a6 = a2 @ a4
    u = a @ (a6 @ (B[13] * a6 + B[11] * a4 + B[9] * a2) + B[7] * a6 + B[5] * a4 + B[3] * a2) + B[1] * a
    v = a6 @ (B[12] * a6 + B[10] * a4 + B[8] * a2) + B[6] * a6 + B[4] * a4 + B[2] * a2 + B[0] * ident
    r = torch.linalg.solve(-u + v, u + v)
    for _ in range(s):
        r = r @ r
    return r


def interp(x, xs, ys):
    if x <= xs[0]:
        i1, i2 = 0, 1
    elif x >= xs[-1]:
        i1, i2 = len(xs) - 2, len(xs) - 1
    else:
        i2 = int(np.argmax(x <= xs))
        i1 = i2 - 1
    return ys[i1] + (((ys[i2] - ys[i1]) / (xs[i2] - xs[i1])) * (x - xs[i1]))


def comm(a, b):
    return a @ b - b @ a


def magnus(policy, gen, dt, t):
    s3, s15 = np.sqrt(3), np.sqrt(15)
    if policy == "M2":
        return dt * gen(t + dt * 0.5)
    if policy == "M4":
        a1 = gen(t + dt * (0.5 - s3 / 6))
        a2 = gen(t + dt * (0.5 + s3 / 6))
        return (dt / 2) * (a1 + a2) + (s3 / 12) * (dt ** 2) * comm(a2, a1)
    a1 = gen(t + dt * (0.5 - s15 / 10))
    a2 = gen(t + dt * 0.5)
    a3 = gen(t + dt * (0.5 + s15 / 10))
    b1 = dt * a2
    b2 = (s15 / 3) * dt * (a3 - a1)
    b3 = (10 / 3) * dt * (a3 - 2 * a2 + a1)
    c12 = comm(b1, b2)
    return (b1 + 0.5 * b3 + (1 / 240)
            * comm(-20 * b1 - b3 + c12, b2 - (1 / 60) * comm(b1, 2 * b3 + c12)))


def abs2(z):
    return torch.real(z * torch.conj(z))


def cost_value(spec, controls, states, case):
    kind, kw = spec
    m = kw.get("cost_multiplier", 1.)
    if kind in ("TargetStateInfidelity", "TargetStateInfidelityTime"):
        t = torch.tensor(np.stack(kw["target_states"]), dtype=C)
        s_count = t.shape[0]
        ip = torch.matmul(torch.conj(t.transpose(-1, -2)), states)[:, 0, 0]
        if not kw.get("neglect_relative_pahse", False):
            fid = abs2(torch.sum(ip)) / s_count ** 2
        else:
            fid = torch.sum(abs2(ip)) / s_count
        val = 1 - fid
        if kind == "TargetStateInfidelityTime":
            val = val / ((kw["system_eval_count"] - 1) // kw.get("cost_eval_step", 1))
        return val * m
    if kind == "ForbidStates":
        forb = kw["forbidden_states"]
        count = (kw["system_eval_count"] - 1) // kw.get("cost_eval_step", 1)
        total = 0
        for i, fs in enumerate(forb):
            sc = 0
            for f in fs:
                ft = torch.tensor(f, dtype=C)
                sc = sc + abs2(torch.matmul(torch.conj(ft.transpose(-1, -2)), states[i])[0, 0])
            total = total + sc / len(fs)
        return total / (count * len(forb)) * m
    nc, k = controls.shape
    if kind == "ControlNorm":
        u = controls
        if kw.get("max_control_norms") is not None:
            u = u / torch.tensor(kw["max_control_norms"])
        if kw.get("control_weights") is not None:
            u = u * torch.tensor(kw["control_weights"])
        return torch.sum(abs2(u)) / (nc * k) * m
    if kind == "ControlVariation":
        u = controls
        order = kw.get("order", 1)
        if kw.get("max_control_norms") is not None:
            u = u / torch.tensor(kw["max_control_norms"])
        d = u
        for _ in range(order):
            d = d[1:] - d[:-1]
        return torch.sum(abs2(d)) / (k * (nc - order) * 2 ** order) * m
    if kind == "ControlArea":
        u = controls / torch.tensor(kw["max_control_norms"])
        return torch.sum(torch.abs(torch.sum(u, dim=0))) / (nc * k) * m
    if kind == "ControlBandwidthMax":
        dtc = kw["evolution_time"] / (nc - 1)
        freqs = np.fft.fftfreq(nc, d=dtc)
        total = 0
        for i, bw in enumerate(kw["max_bandwidths"]):
            mags = torch.abs(torch.fft.fft(controls[:, i]))
            idx = np.nonzero(freqs >= bw)[0]
            pen = mags[idx]
            total = total + torch.sum(pen) / (len(idx) * torch.max(pen))
        return total / k * m
    raise KeyError(kind)


STEP_KINDS = ("TargetStateInfidelityTime", "ForbidStates")


def ad_eval(case, controls_np):
    """Returns (error, grads, final_states) as NumPy."""
    h0 = torch.tensor(case.h0, dtype=C)
    g_re = [torch.tensor(g, dtype=C) for g in case.g_re]
    g_im = [torch.tensor(g, dtype=C) for g in case.g_im] if case.g_im is not None else None
    controls = torch.tensor(controls_np, requires_grad=True)
    xs = np.linspace(0, case.T, case.Nc)
    dt = case.T / (case.N - 1)

    def gen(t):
        base = h0 if case.time_mod is None else h0 * (1 + 0.3 * np.cos(case.time_mod * t))
        u = interp(t, xs, controls)
        h = base
        for k in range(case.K):
            if case.complex_controls:
                h = h + torch.real(u[k]) * g_re[k] + torch.imag(u[k]) * g_im[k]
            else:
                h = h + u[k] * g_re[k]
            if getattr(case, "quad", None) is not None:  # tests/cases.py Case.hamiltonian
                mod2 = (torch.real(u[k]) ** 2 + torch.imag(u[k]) ** 2) if case.complex_controls \
                    else u[k] ** 2
                h = h + mod2 * torch.tensor(case.quad[k], dtype=C)
        return -1j * h

    states = torch.tensor(case.initial_states, dtype=C)
    error = 0
    for step in range(case.N):
        if step % case.cost_eval_step == 0 and step != 0:
            for spec in case.cost_specs:
                if spec[0] in STEP_KINDS:
                    error = error + cost_value(spec, controls, states, case)
        if step != case.N - 1:
            states = torch.matmul(expm_pade(magnus(case.magnus, gen, dt, step * dt)), states)
    for spec in case.cost_specs:
        if spec[0] not in STEP_KINDS:
            error = error + cost_value(spec, controls, states, case)
    error.backward()
    return float(error), controls.grad.numpy().copy(), states.detach().numpy().copy()
