"""
lindblad_model.py - TEST INFRASTRUCTURE: NumPy model of the device algorithm for the Lindblad
path: the matrix-form master equation

    d rho / dt = A(t) rho + rho A(t)^H + sum_i gamma_i L_i rho L_i^H,
    A(t) = -i (H0 + sum_k u_k(t) G_k) - 1/2 sum_i gamma_i L_i^H L_i

integrated with a FIXED-step explicit Runge-Kutta scheme (Dormand-Prince 8(5,3), 12 stages,
coefficients from scipy) on sub-intervals that never straddle a control knot, and its exact
discrete adjoint with per-substep recomputation. The reference integrates the same equation
with an adaptive RK5(4) whose own accuracy is ~1e-10 (tests/test_lindblad_oracle.py), which
bounds the achievable parity.

Nothing here is imported by the product.
"""

import numpy as np
from scipy.integrate._ivp import dop853_coefficients as _dc

from oracle import qoc_numpy as onp

RK_A = np.array(_dc.A[:_dc.N_STAGES, :_dc.N_STAGES])
RK_B = np.array(_dc.B)
RK_C = np.array(_dc.C[:_dc.N_STAGES])
STAGES = len(RK_B)


def h(x):
    return np.conjugate(np.swapaxes(x, -1, -2))


def substep_grid(evolution_time, system_eval_count, control_eval_count, norm_bound,
                 max_phase=0.4, subdivision=None):
    """
    Sub-intervals of every system step: uniform pieces short enough that
    norm_bound * length <= max_phase, further cut at control knots. Returns a list (per system
    step) of lists of (t_a, t_b).
    """
    n_steps = system_eval_count - 1
    dt = evolution_time / n_steps
    knots = (np.linspace(0, evolution_time, control_eval_count) if control_eval_count > 1
             else np.array([]))
    ksub = max(1, int(np.ceil(norm_bound * dt / max_phase)))
    if subdivision is not None:  # time-dependent Hamiltonian: the grid it was sampled for
        assert ksub <= subdivision
        ksub = subdivision
    out = []
    for step in range(n_steps):
        t0, t1 = step * dt, (step + 1) * dt
        cuts = [t0 + (t1 - t0) * q / ksub for q in range(ksub)] + [t1]
        inner = [k for k in knots if t0 + 1e-12 * dt < k < t1 - 1e-12 * dt]
        pts = sorted(set(cuts + inner))
        out.append([(pts[i], pts[i + 1]) for i in range(len(pts) - 1)])
    return out


class StructuredLindblad(object):
    """h0, g (real controls: list of K matrices), gammas (L,), ops (L x n x n). A Hamiltonian
    with explicit time dependence is given by the callables h0_of_t(t) -> (n x n) and
    g_of_t(t) -> list of K matrices; h0 / g then only provide shapes and norm bounds. Likewise
    data_of_t(t) -> (gammas, ops) for a time-dependent lindblad_data (gammas / ops then provide
    the norm bound only)."""

    def __init__(self, h0, g, gammas, ops, h0_of_t=None, g_of_t=None, data_of_t=None):
        self.data_of_t = data_of_t
        self.h0 = np.asarray(h0, dtype=np.complex128)
        self.g = [np.asarray(x, dtype=np.complex128) for x in g]
        self.h0_of_t, self.g_of_t = h0_of_t, g_of_t
        self.gammas = np.zeros(0) if gammas is None else np.asarray(gammas, dtype=np.float64)
        self.ops = (np.zeros((0,) + self.h0.shape, dtype=np.complex128) if ops is None
                    else np.asarray(ops, dtype=np.complex128))
        self.decay = sum((gm * (h(op) @ op) for gm, op in zip(self.gammas, self.ops)),
                         np.zeros_like(self.h0))

    def g_at(self, t):
        return self.g if self.g_of_t is None else self.g_of_t(t)

    def generator(self, u, t=None):
        """(A, gammas, ops) of one stage: A = -i H(u, t) - 1/2 sum gamma_i L_i^H L_i."""
        h0 = self.h0 if self.h0_of_t is None else self.h0_of_t(t)
        ham = h0 + sum((uk * gk for uk, gk in zip(u, self.g_at(t))), np.zeros_like(self.h0))
        if self.data_of_t is None:
            return -1j * ham - 0.5 * self.decay, self.gammas, self.ops
        gammas, ops = self.data_of_t(t)
        decay = sum((gm * (h(op) @ op) for gm, op in zip(gammas, ops)), np.zeros_like(self.h0))
        return -1j * ham - 0.5 * decay, gammas, ops

    def rhs(self, gen, rho):
        a, gammas, ops = gen
        out = a @ rho + rho @ h(a)
        for gm, op in zip(gammas, ops):
            out = out + gm * (op @ rho @ h(op))
        return out

    def rhs_adjoint(self, gen, x):
        """adjoint of rho -> rhs(gen, rho) w.r.t. Re tr(X^H Y)."""
        a, gammas, ops = gen
        out = h(a) @ x + x @ a
        for gm, op in zip(gammas, ops):
            out = out + gm * (h(op) @ x @ op)
        return out

    def norm_bound(self, umax):
        # like the engine: the spectral norm of the control-free Liouvillian
        # rho -> -i [H0, rho] + sum_i gamma_i (L_i rho L_i^H - {L_i^H L_i, rho} / 2) as a whole
        # (the engine: matrix-free power iteration + 2 % margin, capped by the sum of the parts'
        # bounds) plus 2 |u_k| ||G_k||_2 per control. A Hamiltonian / lindblad_data with explicit
        # time dependence keeps the sum of the parts' bounds over its samples.
        ctl = sum(um * 2 * np.linalg.norm(gk, 2) for um, gk in zip(umax, self.g))
        simple = 2 * np.linalg.norm(self.h0, 2) + 2 * sum(
            abs(gm) * np.linalg.norm(op, 2) ** 2 for gm, op in zip(self.gammas, self.ops))
        if self.h0_of_t is not None or self.g_of_t is not None or self.data_of_t is not None:
            return 1.02 * (simple + ctl)
        n = self.h0.shape[0]
        eye = np.eye(n)
        sup = -1j * (np.kron(self.h0, eye) - np.kron(eye, self.h0.T))
        for gm, op in zip(self.gammas, self.ops):
            ld = h(op) @ op
            sup = sup + gm * (np.kron(op, op.conj()) - 0.5 * np.kron(ld, eye)
                              - 0.5 * np.kron(eye, ld.T))
        return 1.02 * (min(np.linalg.norm(sup, 2), simple) + ctl)


def evaluate_with_grad(system, controls, initial_densities, evolution_time, system_eval_count,
                       costs, cost_eval_step=1, want_grad=True, stop_step=None,
                       subdivision=None):
    """
    controls :: (Nc x K) real. costs :: oracle cost objects (cost / states_bar on
    (S x n x n) densities). Returns (error, grads (Nc x K), final_densities).
    """
    controls = np.asarray(controls, dtype=np.float64)
    nc, k = controls.shape
    xs = np.linspace(0, evolution_time, nc)
    n_steps = system_eval_count - 1
    umax = np.max(np.abs(controls), axis=0) if k else []
    grid = substep_grid(evolution_time, system_eval_count, nc, system.norm_bound(umax),
                        subdivision=subdivision)
    step_costs = [c for c in costs if c.requires_step_evaluation]

    def control(t):
        return onp.interpolate_linear_set(t, xs, controls)

    def run_substep(rho, ta, tb, keep=False):
        hh = tb - ta
        ua, ub = control(ta), control(tb)
        ks, ys, gens = [], [], []
        for i in range(STAGES):
            u = (1 - RK_C[i]) * ua + RK_C[i] * ub  # linear inside a sub-interval
            a = system.generator(u, ta + RK_C[i] * hh)
            yi = rho + hh * sum((RK_A[i, j] * ks[j] for j in range(i)), np.zeros_like(rho))
            ks.append(system.rhs(a, yi))
            ys.append(yi)
            gens.append(a)
        new = rho + hh * sum((RK_B[i] * ks[i] for i in range(STAGES)), np.zeros_like(rho))
        return (new, ys, gens) if keep else new

    rho = np.asarray(initial_densities, dtype=np.complex128)
    error = 0.0
    checkpoints, hits = [], {}
    for step in range(system_eval_count):
        if step % cost_eval_step == 0 and step != 0:
            for c in step_costs:
                error = error + c.cost(controls, rho, step)
                hits.setdefault(step, []).append(c)
        if step == n_steps:
            break
        if stop_step is not None and step == stop_step:  # densities at a system step (tests)
            return error, None, rho
        for ta, tb in grid[step]:
            checkpoints.append((step, ta, tb, rho))
            rho = run_substep(rho, ta, tb)
    final = rho
    for c in costs:
        if not c.requires_step_evaluation:
            error = error + c.cost(controls, final, n_steps)
            hits.setdefault(n_steps, []).append(c)
    if not want_grad:
        return error, None, final

    grads = np.zeros((nc, k))
    lam = np.zeros_like(final)
    for c in hits.get(n_steps, []):
        lam = lam + c.states_bar(controls, final, n_steps)
    for step, ta, tb, rho0 in reversed(checkpoints):
        hh = tb - ta
        _, ys, gens = run_substep(rho0, ta, tb, keep=True)
        ia1, wa1, ia2, wa2 = onp.interpolation_weights(ta, xs)
        ib1, wb1, ib2, wb2 = onp.interpolation_weights(tb, xs)
        ybar_stage = [None] * STAGES
        lam_new = lam.copy()
        for i in range(STAGES - 1, -1, -1):
            kb = hh * RK_B[i] * lam
            for j in range(i + 1, STAGES):
                if RK_A[j, i] != 0:
                    kb = kb + hh * RK_A[j, i] * ybar_stage[j]
            ybar_stage[i] = system.rhs_adjoint(gens[i], kb)
            lam_new = lam_new + ybar_stage[i]
            # d rhs / d u_k = -i [G_k, Y_i]  ->  ubar_k = sum_s Re tr(kb_s^H (-i)(G_k Y_s - Y_s G_k))
            g_now = system.g_at(ta + RK_C[i] * hh)
            for kk in range(k):
                val = 0.0
                for s in range(ys[i].shape[0]):
                    comm = g_now[kk] @ ys[i][s] - ys[i][s] @ g_now[kk]
                    val += np.real(np.trace(h(kb[s]) @ (-1j * comm)))
                ca, cb = (1 - RK_C[i]) * val, RK_C[i] * val
                grads[ia1, kk] += wa1 * ca
                grads[ia2, kk] += wa2 * ca
                grads[ib1, kk] += wb1 * cb
                grads[ib2, kk] += wb2 * cb
        lam = lam_new
        # step costs are evaluated on the densities at the START of system step `step`
        if abs(ta - step * (evolution_time / n_steps)) < 1e-14 * max(1.0, evolution_time):
            for c in hits.get(step, []):
                lam = lam + c.states_bar(controls, rho0, step)
    return error, grads, final
