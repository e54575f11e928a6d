"""
qoc_lindblad_numpy.py -- TEST INFRASTRUCTURE: CPU restatement of the reference's discrete
Lindblad evolve loop: matrix-form master equation integrated with the reference's adaptive
Dormand-Prince RK5(4) (atol = 1e-12, rtol = 0, restarted at every system step, quartic dense
output), and the three density costs with hand cotangents.

Citations are `path:line` under /root/reference. Parity status: forward PINNED by
tests/golden/lindblad_*.npz minted from the reference's own forward path
(tools/gen_golden_lindblad.py); gradients pinned by Richardson finite differences of the
reference forward and by an independent AD of the same adaptive integrator
(tools/torch_ad_lindblad.py).
"""

import numpy as np

from oracle.qoc_numpy import OracleCost, _h, interpolate_linear_set

# ----------------------------------------------------------------------------------
# Lindbladian  (qoc/core/mathmethods.py:169-206)
# ----------------------------------------------------------------------------------


def commutator(a, b):
    return np.matmul(a, b) - np.matmul(b, a)


def get_lindbladian(densities, dissipators=None, hamiltonian=None, operators=None):
    if hamiltonian is not None:
        lindbladian = -1j * commutator(hamiltonian, densities)
    else:
        lindbladian = 0
    if dissipators is not None and operators is not None:
        operators_dagger = _h(operators)
        operators_product = np.matmul(operators_dagger, operators)
        for i, operator in enumerate(operators):
            lindbladian = (lindbladian
                           + (dissipators[i]
                              * (np.matmul(np.matmul(operator, densities), operators_dagger[i])
                                 - 0.5 * np.matmul(operators_product[i], densities)
                                 - 0.5 * np.matmul(densities, operators_product[i]))))
    return lindbladian


# ----------------------------------------------------------------------------------
# RKDP5(4)  (qoc/core/mathmethods.py:211-480)
# ----------------------------------------------------------------------------------

C2, C3, C4, C5 = 1 / 5, 3 / 10, 4 / 5, 8 / 9
A21 = 1 / 5
A31, A32 = 3 / 40, 9 / 40
A41, A42, A43 = 44 / 45, -56 / 15, 32 / 9
A51, A52, A53, A54 = 19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729
A61, A62, A63, A64, A65 = 9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656
B1, B3, B4, B5, B6 = 35 / 384, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84
B1H, B3H, B4H, B5H, B6H, B7H = (5179 / 57600, 7571 / 16695, 393 / 640, -92097 / 339200,
                                187 / 2100, 1 / 40)
D1 = -12715105075 / 11282082432
D3 = 87487479700 / 32700410799
D4 = -10690763975 / 1880347072
D5 = 701980252875 / 199316789632
D6 = -1453857185 / 822651844
D7 = 69997945 / 29380423
P_ORDER = 5
ERROR_EXP = -1 / (np.minimum(5, 4) + 1)


def rms_norm(array):
    """qoc/standard/functions/convenience.py:77-91."""
    return np.sqrt(np.sum(array * np.conjugate(array)) / np.prod(np.shape(array)))


def rkdp5_dense(ks, x0, x1, x_eval_step, y0, y1):
    """mathmethods.py:263-304."""
    h = x1 - x0
    r1 = y0
    r2 = y1 - y0
    r3 = y0 + h * ks[0] - y1
    r4 = 2 * (y1 - y0) - h * (ks[0] + ks[6])
    r5 = h * (D1 * ks[0] + D3 * ks[2] + D4 * ks[3] + D5 * ks[4] + D6 * ks[5] + D7 * ks[6])
    theta = (x_eval_step - x0) / h
    theta2 = theta ** 2
    theta3 = theta ** 3
    theta4 = theta2 ** 2
    return (r1 + theta * (r2 + r3) - theta2 * (r3 - r4 - r5) - theta3 * (r4 + 2 * r5)
            + theta4 * r5)


def integrate_rkdp5_step(h, rhs, x0, y0, k1=None):
    """mathmethods.py:307-349."""
    if k1 is None:
        k1 = rhs(x0, y0)
    k2 = rhs(x0 + C2 * h, y0 + h * A21 * k1)
    k3 = rhs(x0 + C3 * h, y0 + h * (A31 * k1 + A32 * k2))
    k4 = rhs(x0 + C4 * h, y0 + h * (A41 * k1 + A42 * k2 + A43 * k3))
    k5 = rhs(x0 + C5 * h, y0 + h * (A51 * k1 + A52 * k2 + A53 * k3 + A54 * k4))
    k6 = rhs(x0 + h, y0 + h * (A61 * k1 + A62 * k2 + A63 * k3 + A64 * k4 + A65 * k5))
    y1 = y0 + h * (B1 * k1 + B3 * k3 + B4 * k4 + B5 * k5 + B6 * k6)
    k7 = rhs(x0 + h, y1)
    y1h = y0 + h * (B1H * k1 + B3H * k3 + B4H * k4 + B5H * k5 + B6H * k6 + B7H * k7)
    return (k1, k2, k3, k4, k5, k6, k7), y1, y1h


def integrate_rkdp5(rhs, x_eval, x_initial, y_initial, atol=1e-12, rtol=0.,
                    step_safety_factor=0.9, step_update_factor_max=10,
                    step_update_factor_min=2e-1, stats=None):
    """mathmethods.py:352-480 (same accept/reject logic, initial step and dense output)."""
    if len(x_eval) == 0:
        raise ValueError("No output was specified.")
    x_final = x_eval[-1]
    f0 = rhs(x_initial, y_initial)
    d0 = rms_norm(y_initial)
    d1 = rms_norm(f0)
    if d0 < 1e-5 or d1 < 1e-5:
        h0 = 1e-6
    else:
        h0 = 0.01 * d0 / d1
    y1 = y_initial + h0 * f0
    f1 = rhs(x_initial + h0, y1)
    d2 = rms_norm(f1 - f0) / h0
    if np.maximum(d1, d2) <= 1e-15:
        h1 = np.maximum(1e-6, h0 * 1e-3)
    else:
        h1 = np.power(0.01 / np.maximum(d1, d2), 1 / (P_ORDER + 1))
    step_current = np.minimum(100 * h0, h1)
    y_eval_list = list()
    x_current = x_initial
    y_current = y_initial
    k1 = f0
    while x_current <= x_final:
        step_rejected = False
        step_accepted = False
        while not step_accepted:
            ks, y1, y1h = integrate_rkdp5_step(step_current, rhs, x_current, y_current, k1=k1)
            if stats is not None:
                stats["rhs"] = stats.get("rhs", 0) + 6
            x_new = x_current + step_current
            scale = atol + np.maximum(np.abs(y1), np.abs(y1h)) * rtol
            error_norm = rms_norm((y1 - y1h) / scale)
            if error_norm < 1:
                step_accepted = True
                if error_norm == 0:
                    factor = step_update_factor_max
                else:
                    factor = np.minimum(step_update_factor_max,
                                        step_safety_factor * np.power(error_norm, ERROR_EXP))
                if step_rejected:
                    factor = np.minimum(1, factor)
                step_current = step_current * factor
            else:
                step_rejected = True
                factor = np.maximum(step_update_factor_min,
                                    step_safety_factor * np.power(error_norm, ERROR_EXP))
                step_current = step_current * factor
        idx = np.nonzero(np.logical_and(x_current <= x_eval, x_eval <= x_new))[0]
        x_eval_step = x_eval[idx]
        if len(x_eval_step) != 0:
            y_eval_step = rkdp5_dense(ks, x_current, x_new, x_eval_step, y_current, y1)
            for y_eval_ in y_eval_step:
                y_eval_list.append(y_eval_)
        x_current = x_new
        y_current = y1
        k1 = ks[6]
    return np.stack(y_eval_list)


# ----------------------------------------------------------------------------------
# density costs  (qoc/standard/costs/targetdensityinfidelity.py:41-69,
#                 targetdensityinfidelitytime.py:47-76, forbiddensities.py:53-85)
# ----------------------------------------------------------------------------------

class TargetDensityInfidelity(OracleCost):
    name = "target_density_infidelity"

    def __init__(self, target_densities, cost_multiplier=1.):
        super().__init__(cost_multiplier)
        self.target_densities = np.asarray(target_densities, dtype=np.complex128)
        self.density_count = target_densities.shape[0]
        self.hilbert_size = target_densities.shape[1]
        self.norm = 1.0

    def _traces(self, densities):
        return np.array([np.trace(np.matmul(_h(t), d))
                         for t, d in zip(self.target_densities, densities)])

    def cost(self, controls, densities, step):
        z = self._traces(densities)
        fid = np.sum(np.abs(z)) / (self.density_count * self.hilbert_size)
        return (1 - fid) / self.norm * self.cost_multiplier

    def states_bar(self, controls, densities, step):
        z = self._traces(densities)
        mag = np.abs(z)
        phase = np.where(mag > 0, z / np.where(mag > 0, mag, 1), 0)
        f = -self.cost_multiplier / (self.norm * self.density_count * self.hilbert_size)
        return f * phase[:, None, None] * self.target_densities


class TargetDensityInfidelityTime(TargetDensityInfidelity):
    """requires_step_evaluation is False in the reference although it divides by the count."""
    name = "target_density_infidelity_time"
    requires_step_evaluation = False

    def __init__(self, system_eval_count, target_densities, cost_eval_step=1, cost_multiplier=1.):
        super().__init__(np.stack(target_densities), cost_multiplier)
        self.cost_eval_count, _ = np.divmod(system_eval_count - 1, cost_eval_step)
        self.norm = self.cost_eval_count


class ForbidDensities(OracleCost):
    name = "forbid_densities"
    requires_step_evaluation = True

    def __init__(self, forbidden_densities, system_eval_count, cost_eval_step=1,
                 cost_multiplier=1.):
        super().__init__(cost_multiplier)
        density_count = forbidden_densities.shape[0]
        count, _ = np.divmod(system_eval_count - 1, cost_eval_step)
        self.cost_normalization_constant = count * density_count
        self.forbidden = [np.asarray(f, dtype=np.complex128) for f in forbidden_densities]
        self.forbidden_densities_count = np.array([f.shape[0] for f in forbidden_densities])
        self.hilbert_size = forbidden_densities.shape[3]

    def cost(self, controls, densities, step):
        cost = 0
        for i, fs in enumerate(self.forbidden):
            dc = 0
            for f in fs:
                ip = np.trace(np.matmul(_h(f), densities[i])) / self.hilbert_size
                dc = dc + np.real(ip * np.conjugate(ip))
            cost = cost + dc / self.forbidden_densities_count[i]
        return cost / self.cost_normalization_constant * self.cost_multiplier

    def states_bar(self, controls, densities, step):
        out = np.zeros_like(np.asarray(densities, dtype=np.complex128))
        for i, fs in enumerate(self.forbidden):
            scale = 2 * self.cost_multiplier / (self.cost_normalization_constant
                                                * self.forbidden_densities_count[i]
                                                * self.hilbert_size)
            for f in fs:
                ip = np.trace(np.matmul(_h(f), densities[i])) / self.hilbert_size
                out[i] = out[i] + scale * ip * f
        return out


# ----------------------------------------------------------------------------------
# evolve loop  (qoc/core/lindbladdiscrete.py:357-495)
# ----------------------------------------------------------------------------------

class LindbladProblem(object):
    def __init__(self, evolution_time, initial_densities, system_eval_count, hamiltonian=None,
                 lindblad_data=None, control_eval_count=0, costs=(), cost_eval_step=1,
                 complex_controls=False, control_count=0):
        self.evolution_time = evolution_time
        self.initial_densities = np.asarray(initial_densities, dtype=np.complex128)
        self.system_eval_count = system_eval_count
        self.hamiltonian = hamiltonian
        self.lindblad_data = lindblad_data
        self.control_eval_count = control_eval_count
        self.control_eval_times = np.linspace(0, evolution_time, control_eval_count)
        self.costs = list(costs)
        self.step_costs = [c for c in self.costs if c.requires_step_evaluation]
        self.cost_eval_step = cost_eval_step
        self.dt = evolution_time / (system_eval_count - 1)
        self.final_system_eval_step = system_eval_count - 1
        self.complex_controls = complex_controls
        self.control_count = control_count


def rhs_lindbladian(problem, controls):
    """lindbladdiscrete.py:444-495."""
    def rhs(time, densities):
        if controls is not None and problem.control_eval_count > 0:
            u = interpolate_linear_set(time, problem.control_eval_times, controls)
        else:
            u = None
        h = problem.hamiltonian(u, time) if problem.hamiltonian is not None else None
        if problem.lindblad_data is not None:
            dissipators, operators = problem.lindblad_data(time)
        else:
            dissipators, operators = None, None
        return get_lindbladian(densities, dissipators, h, operators)
    return rhs


def evaluate(problem, controls, intermediate=None, stats=None):
    """_evaluate_lindblad_discrete, lindbladdiscrete.py:357-441. Returns (error, densities)."""
    densities = problem.initial_densities
    error = 0
    rhs = rhs_lindbladian(problem, controls)
    step = 0
    for step in range(problem.system_eval_count):
        if intermediate is not None:
            intermediate.append(densities)
        _, rem = divmod(step, problem.cost_eval_step)
        time = step * problem.dt
        if rem == 0 and step != 0:
            for c in problem.step_costs:
                error = error + c.cost(controls, densities, step)
        if step != problem.final_system_eval_step:
            # one evaluation point: the dense output broadcasts theta over the last axis and
            # the loop at mathmethods.py:471-472 unpacks the density axis, so the result has
            # the shape of `densities` again
            densities = integrate_rkdp5(rhs, np.array([time + problem.dt]), time, densities,
                                        stats=stats)
    for c in problem.costs:
        if not c.requires_step_evaluation:
            error = error + c.cost(controls, densities, step)
    return error, densities
