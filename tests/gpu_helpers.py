"""Helpers for the -m gpu tests: turn a tests/cases.py Case into an engine problem."""

import numpy as np

from qoc_amd import engine as eng


def sample_hamiltonian(case):
    """h0[nt][n][n], g[nt][Kr][n][n] at the quadrature times of case.magnus (Kr real controls)."""
    nsteps = case.N - 1
    dt = case.T / (case.N - 1)
    g_list = []
    for k in range(case.K):
        g_list.append(case.g_re[k])
        if case.complex_controls:
            g_list.append(case.g_im[k])
    if case.time_mod is None:
        h0 = np.asarray(case.h0, dtype=np.complex128)[None]
        g = np.asarray(g_list, dtype=np.complex128).reshape(1, len(g_list), case.n, case.n)
    else:
        nodes = {"M2": (0.5,), "M4": (0.5 - 3 ** 0.5 / 6, 0.5 + 3 ** 0.5 / 6),
                 "M6": (0.5 - 15 ** 0.5 / 10, 0.5, 0.5 + 15 ** 0.5 / 10)}[case.magnus]
        times = [j * dt + dt * c for j in range(nsteps) for c in nodes]
        h0 = np.stack([case.h0 * (1 + 0.3 * np.cos(case.time_mod * t)) for t in times])
        g = np.stack([np.asarray(g_list, dtype=np.complex128) for _ in times])
    return h0, g


def device_costs(case):
    """State costs as device descriptors; returns (descs, host_specs) - host_specs are the
    control-only costs the host evaluates."""
    descs, host = [], []
    for kind, kw in case.cost_specs:
        m = kw.get("cost_multiplier", 1.)
        if kind in ("TargetStateInfidelity", "TargetStateInfidelityTime"):
            targets = np.stack(kw["target_states"])[:, :, 0]
            scale = m
            step = 0
            if kind == "TargetStateInfidelityTime":
                scale = m / ((kw["system_eval_count"] - 1) // kw.get("cost_eval_step", 1))
                step = 1
            descs.append(dict(kind=eng.COST_TARGET_INCOHERENT if kw.get("neglect_relative_pahse", False)
                              else eng.COST_TARGET_COHERENT, step_cost=step, scale=scale,
                              vectors=targets))
        elif kind == "ForbidStates":
            forb = kw["forbidden_states"]
            count = (kw["system_eval_count"] - 1) // kw.get("cost_eval_step", 1)
            vecs = np.concatenate([np.asarray(f)[:, :, 0] for f in forb])
            descs.append(dict(kind=eng.COST_FORBID, step_cost=1, scale=m / (count * len(forb)),
                              vectors=vecs, counts=[len(f) for f in forb]))
        else:
            host.append((kind, kw))
    return descs, host


def real_controls(case, controls):
    """(B, Nc, K) complex or real -> (B, Nc, Kr) float64 with (re, im) interleaved."""
    controls = np.asarray(controls)
    if case.complex_controls:
        out = np.empty(controls.shape[:-1] + (2 * case.K,), dtype=np.float64)
        out[..., 0::2] = controls.real
        out[..., 1::2] = controls.imag
        return out
    return controls.astype(np.float64)


def complex_grads(case, grads):
    if case.complex_controls:
        return grads[..., 0::2] + 1j * grads[..., 1::2]
    return grads


def setup_engine(engine, case):
    h0, g = sample_hamiltonian(case)
    descs, host = device_costs(case)
    kr = case.K * (2 if case.complex_controls else 1)
    engine.set_schroedinger_problem(
        case.n, case.S, kr, case.Nc, case.N, case.T, h0, g, case.initial_states[:, :, 0],
        costs=descs, cost_eval_step=case.cost_eval_step, magnus_policy=case.magnus)
    return host


# ---- Lindblad -----------------------------------------------------------------------------------

def lindblad_device_costs(case):
    """Density costs as device descriptors (qocx.h kinds 3/4)."""
    descs = []
    for kind, kw in case.cost_specs:
        m = kw.get("cost_multiplier", 1.)
        if kind == "TargetDensityInfidelity":
            descs.append(dict(kind=eng.COST_TARGET_DENSITY, step_cost=0, scale=m,
                              vectors=np.stack(kw["target_densities"])))
        elif kind == "TargetDensityInfidelityTime":
            # requires_step_evaluation is False in the reference (targetdensityinfidelitytime.py:43)
            count = (kw["system_eval_count"] - 1) // kw.get("cost_eval_step", 1)
            descs.append(dict(kind=eng.COST_TARGET_DENSITY, step_cost=0, scale=m / count,
                              vectors=np.stack(kw["target_densities"])))
        elif kind == "ForbidDensities":
            forb = kw["forbidden_densities"]
            count = (kw["system_eval_count"] - 1) // kw.get("cost_eval_step", 1)
            mats = np.concatenate([np.asarray(f) for f in forb])
            descs.append(dict(kind=eng.COST_FORBID_DENSITY, step_cost=1,
                              scale=m / (count * len(forb)), vectors=mats,
                              counts=[len(f) for f in forb]))
        else:
            raise ValueError(kind)
    return descs


def lindblad_generators(case):
    g = []
    for k in range(case.K):
        g.append(case.g_re[k])
        if case.complex_controls:
            g.append(case.g_im[k])
    return g


def setup_lindblad_engine(engine, case):
    g = lindblad_generators(case)
    engine.set_lindblad_problem(
        case.n, case.initial_densities.shape[0], len(g), case.Nc, case.N, case.T, case.h0, g,
        case.dissipators, case.operators, case.initial_densities,
        costs=lindblad_device_costs(case), cost_eval_step=case.cost_eval_step)
