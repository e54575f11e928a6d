"""
qoc_numpy.py -- TEST INFRASTRUCTURE: CPU restatement of the reference's discrete
Schroedinger GRAPE hot path (forward, op for op) and a hand-derived dense
reverse-mode adjoint standing in for HIPS autograd.

Citations are `path:line` under /root/reference.  Parity status: forward PINNED by
tests/golden/*.npz (minted from the reference's own forward path, see
tools/gen_golden.py); gradients pinned by finite differences of the reference forward
and an independent AD, because the reference tests pin no gradient value.

Convention for every cotangent: for the real scalar cost C and a
complex array X,  Xbar := dC/dRe(X) + i dC/dIm(X), so dC = Re tr(Xbar^H dX).  This is
the convention of qoc's returned `grads` (qoc/core/schroedingerdiscrete.py:320-324).
"""

import numpy as np

# ----------------------------------------------------------------------------------
# expm  (qoc/standard/functions/expm.py)
# ----------------------------------------------------------------------------------

# Pade-13 coefficients, expm.py:86-101 (Higham 2005, algorithm 2.3).
PADE_B = (
    64764752532480000, 32382376266240000, 7771770303897600, 1187353796428800,
    129060195264000, 10559470521600, 670442572800, 33522128640, 1323241920,
    40840800, 960960, 16380, 182, 1,
)
# expm.py:192-207: only theta_13 ever matters, because the order-selection loop keeps the
# LAST order whose theta exceeds the norm (expm.py:230-234) and theta is increasing.
THETA_13 = 5.371920351148152


def one_norm(a):
    """expm.py:103-116 -- max column sum of complex moduli."""
    return np.max(np.sum(np.abs(a), axis=0))


def pade_scale_count(norm1):
    """expm.py:230-241 -- number of squarings; 0 whenever norm1 < theta_13."""
    if norm1 < THETA_13:
        return 0
    return max(0, int(np.ceil(np.log2(norm1 / THETA_13))))


def expm_pade_cached(a):
    """
    expm.py:210-252 with the always-13 order.
    Returns (r, cache); cache holds what the reverse sweep needs.
    """
    b = PADE_B
    n = a.shape[0]
    norm1 = one_norm(a)
    s = pade_scale_count(norm1)
    if norm1 < THETA_13:
        a_s = a
    else:
        a_s = a * (2 ** -s)  # expm.py:238-241
    ident = np.eye(n)
    # pade13, expm.py:153-159, same association order.
    a2 = np.matmul(a_s, a_s)
    a4 = np.matmul(a2, a2)
    a6 = np.matmul(a2, a4)
    w1 = b[13] * a6 + b[11] * a4 + b[9] * a2
    w2 = np.matmul(a6, w1) + b[7] * a6 + b[5] * a4 + b[3] * a2
    u = np.matmul(a_s, w2) + b[1] * a_s
    w3 = b[12] * a6 + b[10] * a4 + b[8] * a2
    v = np.matmul(a6, w3) + b[6] * a6 + b[4] * a4 + b[2] * a2 + b[0] * ident
    p = -u + v
    q = u + v
    r = np.linalg.solve(p, q)  # expm.py:246
    squares = [r]
    for _ in range(s):  # expm.py:249-250
        r = np.matmul(r, r)
        squares.append(r)
    cache = dict(s=s, a=a_s, a2=a2, a4=a4, a6=a6, w1=w1, w2=w2, w3=w3,
                 p=p, squares=squares)
    return r, cache


def expm_pade(a):
    return expm_pade_cached(a)[0]


def _h(x):
    return np.conjugate(np.swapaxes(x, -1, -2))


def expm_pade_vjp(cache, rbar):
    """Reverse rule of expm_pade; returns abar for the UNSCALED input."""
    b = PADE_B
    s = cache["s"]
    a, a2, a4, a6 = cache["a"], cache["a2"], cache["a4"], cache["a6"]
    w1, w2, w3 = cache["w1"], cache["w2"], cache["w3"]
    squares = cache["squares"]
    # r <- r r, s times
    for k in range(s, 0, -1):
        rk = squares[k - 1]
        rbar = np.matmul(rbar, _h(rk)) + np.matmul(_h(rk), rbar)
    r0 = squares[0]
    # r = solve(p, q)
    qbar = np.linalg.solve(_h(cache["p"]), rbar)
    pbar = -np.matmul(qbar, _h(r0))
    ubar = qbar - pbar
    vbar = qbar + pbar
    # u = a w2 + b1 a
    abar = np.matmul(ubar, _h(w2)) + b[1] * ubar
    w2bar = np.matmul(_h(a), ubar)
    # w2 = a6 w1 + b7 a6 + b5 a4 + b3 a2
    a6bar = np.matmul(w2bar, _h(w1)) + b[7] * w2bar
    w1bar = np.matmul(_h(a6), w2bar)
    a4bar = b[5] * w2bar
    a2bar = b[3] * w2bar
    # w1 = b13 a6 + b11 a4 + b9 a2
    a6bar = a6bar + b[13] * w1bar
    a4bar = a4bar + b[11] * w1bar
    a2bar = a2bar + b[9] * w1bar
    # v = a6 w3 + b6 a6 + b4 a4 + b2 a2 + b0 I
    a6bar = a6bar + np.matmul(vbar, _h(w3)) + b[6] * vbar
    w3bar = np.matmul(_h(a6), vbar)
    a4bar = a4bar + b[4] * vbar
    a2bar = a2bar + b[2] * vbar
    # w3 = b12 a6 + b10 a4 + b8 a2
    a6bar = a6bar + b[12] * w3bar
    a4bar = a4bar + b[10] * w3bar
    a2bar = a2bar + b[8] * w3bar
    # a6 = a2 a4 ; a4 = a2 a2 ; a2 = a a
    a2bar = a2bar + np.matmul(a6bar, _h(a4))
    a4bar = a4bar + np.matmul(_h(a2), a6bar)
    a2bar = a2bar + np.matmul(a4bar, _h(a2)) + np.matmul(_h(a2), a4bar)
    abar = abar + np.matmul(a2bar, _h(a)) + np.matmul(_h(a), a2bar)
    # a <- a 2^-s
    return abar * (2 ** -s)


# ----------------------------------------------------------------------------------
# interpolation and Magnus generators  (qoc/core/mathmethods.py)
# ----------------------------------------------------------------------------------

def interpolate_linear_points(x1, x2, x3, y1, y2):
    """mathmethods.py:14-33."""
    return y1 + (((y2 - y1) / (x2 - x1)) * (x3 - x1))


def interpolation_bracket(x, xs):
    """Index pair (i1, i2) used by mathmethods.py:54-65 for abscissa x."""
    if x <= xs[0]:
        return 0, 1
    if x >= xs[-1]:
        return len(xs) - 2, len(xs) - 1
    index = int(np.argmax(x <= xs))
    return index - 1, index


def interpolate_linear_set(x, xs, ys):
    """mathmethods.py:36-67."""
    i1, i2 = interpolation_bracket(x, xs)
    return interpolate_linear_points(xs[i1], xs[i2], x, ys[i1], ys[i2])


def interpolation_weights(x, xs):
    """(i1, w1, i2, w2) with y(x) = w1 ys[i1] + w2 ys[i2] -- the transpose used backwards."""
    i1, i2 = interpolation_bracket(x, xs)
    theta = (x - xs[i1]) / (xs[i2] - xs[i1])
    return i1, 1.0 - theta, i2, theta


def commutator(a, b):
    """qoc/standard/functions/convenience.py:16-29."""
    return np.matmul(a, b) - np.matmul(b, a)


def _commutator_vjp(x, y, zbar):
    """Z = [X, Y] -> (Xbar, Ybar)."""
    xbar = np.matmul(zbar, _h(y)) - np.matmul(_h(y), zbar)
    ybar = np.matmul(_h(x), zbar) - np.matmul(zbar, _h(x))
    return xbar, ybar


# mathmethods.py:72, 96-98, 125-132
M2_C1 = 0.5
M4_C1 = 0.5 - np.divide(np.sqrt(3), 6)
M4_C2 = 0.5 + np.divide(np.sqrt(3), 6)
M4_F0 = np.divide(np.sqrt(3), 12)
M6_C1 = 0.5 - np.divide(np.sqrt(15), 10)
M6_C2 = 0.5
M6_C3 = 0.5 + np.divide(np.sqrt(15), 10)
M6_F0 = np.divide(np.sqrt(15), 3)
M6_F1 = np.divide(10, 3)
M6_F2 = np.divide(1, 2)
M6_F3 = np.divide(1, 240)
M6_F4 = np.divide(1, 60)

MAGNUS_NODES = {"M2": (M2_C1,), "M4": (M4_C1, M4_C2), "M6": (M6_C1, M6_C2, M6_C3)}


def magnus_combine(policy, dt, gens):
    """
    mathmethods.py:74-164 given the generator samples a(t + c_i dt).
    Returns (m, cache).
    """
    if policy == "M2":
        (a1,) = gens
        return dt * a1, None
    if policy == "M4":
        a1, a2 = gens
        m = ((dt / 2) * (a1 + a2) + M4_F0 * (dt ** 2) * commutator(a2, a1))
        return m, None
    if policy == "M6":
        a1, a2, a3 = gens
        b1 = dt * a2
        b2 = M6_F0 * dt * (a3 - a1)
        b3 = M6_F1 * dt * (a3 - 2 * a2 + a1)
        c12 = commutator(b1, b2)
        x = -20 * b1 - b3 + c12
        w = 2 * b3 + c12
        y = b2 - M6_F4 * commutator(b1, w)
        m = b1 + M6_F2 * b3 + M6_F3 * commutator(x, y)
        return m, dict(b1=b1, b2=b2, b3=b3, x=x, w=w, y=y)
    raise ValueError("Unrecognized magnus policy {}.".format(policy))


def magnus_combine_vjp(policy, dt, gens, cache, mbar):
    """Cotangents of the generator samples."""
    if policy == "M2":
        return (dt * mbar,)
    if policy == "M4":
        a1, a2 = gens
        cbar = M4_F0 * (dt ** 2) * mbar
        a2bar, a1bar = _commutator_vjp(a2, a1, cbar)
        return ((dt / 2) * mbar + a1bar, (dt / 2) * mbar + a2bar)
    if policy == "M6":
        b1, b2, b3 = cache["b1"], cache["b2"], cache["b3"]
        x, w, y = cache["x"], cache["w"], cache["y"]
        b1bar = mbar.copy()
        b3bar = M6_F2 * mbar
        xbar, ybar = _commutator_vjp(x, y, M6_F3 * mbar)
        b1bar = b1bar - 20 * xbar
        b3bar = b3bar - xbar
        c12bar = xbar.copy()
        b2bar = ybar.copy()
        innerbar = -M6_F4 * ybar
        d1, wbar = _commutator_vjp(b1, w, innerbar)
        b1bar = b1bar + d1
        b3bar = b3bar + 2 * wbar
        c12bar = c12bar + wbar
        d1, d2 = _commutator_vjp(b1, b2, c12bar)
        b1bar = b1bar + d1
        b2bar = b2bar + d2
        a1bar = -M6_F0 * dt * b2bar + M6_F1 * dt * b3bar
        a2bar = dt * b1bar - 2 * M6_F1 * dt * b3bar
        a3bar = M6_F0 * dt * b2bar + M6_F1 * dt * b3bar
        return (a1bar, a2bar, a3bar)
    raise ValueError("Unrecognized magnus policy {}.".format(policy))


# ----------------------------------------------------------------------------------
# control plumbing  (qoc/core/common.py)
# ----------------------------------------------------------------------------------

def clip_control_norms(controls, max_control_norms):
    """common.py:8-30 -- in place."""
    for i, max_norm in enumerate(max_control_norms):
        column = controls[:, i]
        norms = np.abs(column)
        bad = np.nonzero(np.less(max_norm, norms))
        column[bad] = (column[bad] / norms[bad]) * max_norm


def strip_controls(complex_controls, controls):
    """common.py:226-246."""
    flat = np.ravel(controls)
    if complex_controls:
        flat = np.hstack((np.real(flat), np.imag(flat)))
    return flat


def slap_controls(complex_controls, controls, controls_shape):
    """common.py:201-223."""
    if complex_controls:
        real, imag = np.split(controls, 2)
        controls = real + 1j * imag
    return np.reshape(controls, controls_shape)


# ----------------------------------------------------------------------------------
# costs  (qoc/standard/costs/*.py) with hand gradients
# ----------------------------------------------------------------------------------

class OracleCost(object):
    name = "parent_cost"
    requires_step_evaluation = False
    uses_states = True

    def __init__(self, cost_multiplier=1.):
        self.cost_multiplier = cost_multiplier

    def cost(self, controls, states, step):
        raise NotImplementedError

    def states_bar(self, controls, states, step):
        """dC/dRe(states) + i dC/dIm(states), same shape as states; None if none."""
        return None

    def controls_bar(self, controls, states, step):
        return None


class TargetStateInfidelity(OracleCost):
    """targetstateinfidelity.py:12-63."""
    name = "target_state_infidelity"

    def __init__(self, target_states, neglect_relative_pahse=False, cost_multiplier=1.):
        super().__init__(cost_multiplier)
        self.state_count = target_states.shape[0]
        self.target_states = np.asarray(target_states, dtype=np.complex128)
        self.target_states_dagger = _h(self.target_states)
        self.neglect_relative_pahse = neglect_relative_pahse
        self.norm = 1.0

    def _inner(self, states):
        return np.matmul(self.target_states_dagger, states)[:, 0, 0]

    def cost(self, controls, states, step):
        ip = self._inner(states)
        if not self.neglect_relative_pahse:
            tot = np.sum(ip)
            fid = np.real(tot * np.conjugate(tot)) / self.state_count ** 2
        else:
            fid = np.sum(np.real(ip * np.conjugate(ip))) / self.state_count
        return (1 - fid) / self.norm * self.cost_multiplier

    def states_bar(self, controls, states, step):
        ip = self._inner(states)
        m = self.cost_multiplier / self.norm
        if not self.neglect_relative_pahse:
            tot = np.sum(ip)
            return -(2 * m / self.state_count ** 2) * tot * self.target_states
        return -(2 * m / self.state_count) * ip[:, None, None] * self.target_states


class TargetStateInfidelityTime(TargetStateInfidelity):
    """targetstateinfidelitytime.py:13-73."""
    name = "target_state_infidelity_time"
    requires_step_evaluation = True

    def __init__(self, system_eval_count, target_states, neglect_relative_pahse=False,
                 cost_eval_step=1, cost_multiplier=1.):
        super().__init__(np.stack(target_states), neglect_relative_pahse, cost_multiplier)
        self.cost_eval_count, _ = np.divmod(system_eval_count - 1, cost_eval_step)
        self.norm = self.cost_eval_count


class ForbidStates(OracleCost):
    """forbidstates.py:12-81."""
    name = "forbid_states"
    requires_step_evaluation = True

    def __init__(self, forbidden_states, system_eval_count, cost_eval_step=1,
                 cost_multiplier=1.):
        super().__init__(cost_multiplier)
        state_count = forbidden_states.shape[0]
        cost_evaluation_count, _ = np.divmod(system_eval_count - 1, cost_eval_step)
        self.cost_normalization_constant = cost_evaluation_count * state_count
        self.forbidden_states = [np.asarray(f, dtype=np.complex128) for f in forbidden_states]
        self.forbidden_states_count = np.array([f.shape[0] for f in forbidden_states])

    def cost(self, controls, states, step):
        cost = 0
        for i, forbidden in enumerate(self.forbidden_states):
            state_cost = 0
            for f in forbidden:
                ip = np.matmul(_h(f), states[i])[0, 0]
                state_cost = state_cost + np.real(ip * np.conjugate(ip))
            cost = cost + state_cost / self.forbidden_states_count[i]
        return cost / self.cost_normalization_constant * self.cost_multiplier

    def states_bar(self, controls, states, step):
        out = np.zeros_like(np.asarray(states, dtype=np.complex128))
        for i, forbidden in enumerate(self.forbidden_states):
            scale = 2 * self.cost_multiplier / (self.cost_normalization_constant
                                                * self.forbidden_states_count[i])
            for f in forbidden:
                ip = np.matmul(_h(f), states[i])[0, 0]
                out[i] = out[i] + scale * ip * f
        return out


class ControlNorm(OracleCost):
    """controlnorm.py:11-73."""
    name = "control_norm"
    uses_states = False

    def __init__(self, control_count, control_eval_count, control_weights=None,
                 cost_multiplier=1., max_control_norms=None):
        super().__init__(cost_multiplier)
        self.control_weights = control_weights
        self.controls_size = control_eval_count * control_count
        self.max_control_norms = max_control_norms

    def _scale(self, controls):
        scale = np.ones(controls.shape[1])
        if self.max_control_norms is not None:
            scale = scale / self.max_control_norms
        if self.control_weights is not None:
            scale = scale * self.control_weights
        return scale

    def cost(self, controls, states, step):
        if self.max_control_norms is not None:
            controls = controls / self.max_control_norms
        if self.control_weights is not None:
            controls = controls[:, ] * self.control_weights
        cost = np.sum(np.real(controls * np.conjugate(controls)))
        return cost / self.controls_size * self.cost_multiplier

    def controls_bar(self, controls, states, step):
        scale = self._scale(controls)
        return 2 * self.cost_multiplier / self.controls_size * controls * scale ** 2


class ControlVariation(OracleCost):
    """controlvariation.py:11-75."""
    name = "control_variation"
    uses_states = False

    def __init__(self, control_count, control_eval_count, cost_multiplier=1.,
                 max_control_norms=None, order=1):
        super().__init__(cost_multiplier)
        self.max_control_norms = max_control_norms
        self.diffs_size = control_count * (control_eval_count - order)
        self.order = order
        self.cost_normalization_constant = self.diffs_size * (2 ** self.order)

    def cost(self, controls, states, step):
        if self.max_control_norms is not None:
            controls = controls / self.max_control_norms
        diffs = np.diff(controls, axis=0, n=self.order)
        cost = np.sum(np.real(diffs * np.conjugate(diffs)))
        return cost / self.cost_normalization_constant * self.cost_multiplier

    def controls_bar(self, controls, states, step):
        scale = np.ones(controls.shape[1])
        if self.max_control_norms is not None:
            scale = scale / self.max_control_norms
        diffs = np.diff(controls * scale, axis=0, n=self.order)
        g = 2 * self.cost_multiplier / self.cost_normalization_constant * diffs
        # transpose of the n-th order forward difference
        for _ in range(self.order):
            padded = np.zeros((g.shape[0] + 1, g.shape[1]), dtype=g.dtype)
            padded[1:] += g
            padded[:-1] -= g
            g = padded
        return g * scale


class ControlArea(OracleCost):
    """controlarea.py:11-67 (raises NameError when max_control_norms is None, :58 vs :64)."""
    name = "control_area"
    uses_states = False

    def __init__(self, control_count, control_eval_count, cost_multiplier=1.,
                 max_control_norms=None):
        super().__init__(cost_multiplier)
        self.control_count = control_count
        self.control_size = control_count * control_eval_count
        self.max_control_norms = max_control_norms

    def cost(self, controls, states, step):
        if self.max_control_norms is None:
            raise NameError("name 'normalized_controls' is not defined")
        normalized = controls / self.max_control_norms
        cost = 0
        for i in range(self.control_count):
            cost = cost + np.abs(np.sum(normalized[:, i]))
        return cost / self.control_size * self.cost_multiplier

    def controls_bar(self, controls, states, step):
        normalized = controls / self.max_control_norms
        sums = np.sum(normalized, axis=0)
        mags = np.abs(sums)
        phase = np.where(mags > 0, sums / np.where(mags > 0, mags, 1), 0)
        g = (self.cost_multiplier / self.control_size) * phase / self.max_control_norms
        return np.broadcast_to(g, controls.shape).astype(controls.dtype)


class ControlBandwidthMax(OracleCost):
    """controlbandwidthmax.py:11-77."""
    name = "control_bandwidth_max"
    uses_states = False

    def __init__(self, control_count, control_eval_count, evolution_time, max_bandwidths,
                 cost_multiplier=1.):
        super().__init__(cost_multiplier)
        self.max_bandwidths = max_bandwidths
        self.control_count = control_count
        dt = evolution_time / (control_eval_count - 1)
        self.freqs = np.fft.fftfreq(control_eval_count, d=dt)

    def cost(self, controls, states, step):
        cost = 0
        for i, max_bandwidth in enumerate(self.max_bandwidths):
            mags = np.abs(np.fft.fft(controls[:, i]))
            idx = np.nonzero(self.freqs >= max_bandwidth)[0]
            pen = mags[idx]
            cost = cost + np.sum(pen) / (idx.shape[0] * np.max(pen))
        return cost / self.control_count * self.cost_multiplier

    def controls_bar(self, controls, states, step):
        out = np.zeros(controls.shape, dtype=np.complex128)
        n = controls.shape[0]
        for i, max_bandwidth in enumerate(self.max_bandwidths):
            spectrum = np.fft.fft(controls[:, i])
            mags = np.abs(spectrum)
            idx = np.nonzero(self.freqs >= max_bandwidth)[0]
            pen = mags[idx]
            top = np.max(pen)
            jmax = idx[int(np.argmax(pen))]
            total = np.sum(pen)
            count = idx.shape[0]
            # d(penalty_normalized)/d|F_f|
            magbar = np.zeros(n)
            magbar[idx] = 1.0 / (count * top)
            magbar[jmax] -= total / (count * top * top)
            # |F| -> F: Fbar = magbar * F/|F| ; F = DFT(u): ubar_t = sum_f Fbar_f conj(e^{-2 pi i f t/n})
            safe = np.where(mags > 0, mags, 1)
            fbar = magbar * spectrum / safe
            ubar = np.fft.ifft(fbar) * n
            out[:, i] = ubar * self.cost_multiplier / self.control_count
        if not np.iscomplexobj(controls):
            return np.real(out)
        return out


# ----------------------------------------------------------------------------------
# the evolve loop and its adjoint  (qoc/core/schroedingerdiscrete.py:356-502)
# ----------------------------------------------------------------------------------

class SchroedingerProblem(object):
    """
    Static data of one evolution, named after the reference's ProgramState fields
    (qoc/models/programstate.py:33-61).
    """

    def __init__(self, evolution_time, hamiltonian, initial_states, system_eval_count,
                 control_eval_count=0, costs=(), cost_eval_step=1, magnus_policy="M2",
                 complex_controls=False, control_count=0):
        self.evolution_time = evolution_time
        self.hamiltonian = hamiltonian
        self.initial_states = np.asarray(initial_states, dtype=np.complex128)
        self.system_eval_count = system_eval_count
        self.control_eval_count = control_eval_count
        self.control_eval_times = np.linspace(0, evolution_time, control_eval_count)
        self.costs = list(costs)
        self.step_costs = [c for c in self.costs if c.requires_step_evaluation]
        self.cost_eval_step = cost_eval_step
        self.dt = evolution_time / (system_eval_count - 1)
        self.final_system_eval_step = system_eval_count - 1
        self.magnus_policy = magnus_policy
        self.complex_controls = complex_controls
        self.control_count = control_count

    # -- Hamiltonian structure: H(u,t) = H0(t) + sum_k Re(u_k) G_k(t) + Im(u_k) G'_k(t)
    def hamiltonian_slopes(self, time):
        """dH/dRe(u_k) and dH/dIm(u_k) at `time`, by probing the (real-linear) callable."""
        k = self.control_count
        dtype = np.complex128 if self.complex_controls else np.float64
        zero = np.zeros(k, dtype=dtype)
        h0 = np.asarray(self.hamiltonian(zero, time), dtype=np.complex128)
        g_re, g_im = [], []
        for i in range(k):
            e = zero.copy()
            e[i] = 1
            g_re.append(np.asarray(self.hamiltonian(e, time), dtype=np.complex128) - h0)
            if self.complex_controls:
                e = zero.copy()
                e[i] = 1j
                g_im.append(np.asarray(self.hamiltonian(e, time), dtype=np.complex128) - h0)
        return g_re, g_im


def _generator(problem, controls, time):
    """get_hamiltonian closure, schroedingerdiscrete.py:483-486."""
    if controls is not None and problem.control_eval_count > 0:
        u = interpolate_linear_set(time, problem.control_eval_times, controls)
    else:
        u = None
    return -1j * problem.hamiltonian(u, time)


def evolve_step(problem, controls, states, time, want_cache=False):
    """_evolve_step_schroedinger_discrete, schroedingerdiscrete.py:441-502."""
    dt = problem.dt
    nodes = MAGNUS_NODES.get(problem.magnus_policy)
    if nodes is None:
        raise ValueError("Unrecognized magnus policy {}.".format(problem.magnus_policy))
    times = [time + dt * c for c in nodes]
    gens = [_generator(problem, controls, t) for t in times]
    magnus, mcache = magnus_combine(problem.magnus_policy, dt, gens)
    unitary, ecache = expm_pade_cached(magnus)
    new_states = np.matmul(unitary, states)
    if want_cache:
        return new_states, dict(times=times, gens=gens, mcache=mcache, ecache=ecache,
                                unitary=unitary, states=states)
    return new_states


def evaluate(problem, controls, want_tape=False, intermediate=None):
    """
    _evaluate_schroedinger_discrete, schroedingerdiscrete.py:356-438.
    Returns (error, final_states) or (error, final_states, tape).
    """
    states = problem.initial_states
    error = 0
    tape = [] if want_tape else None
    cost_hits = [] if want_tape else None
    for step in range(problem.system_eval_count):
        if intermediate is not None:
            intermediate.append(states)
        cost_step, rem = divmod(step, problem.cost_eval_step)
        time = step * problem.dt
        if rem == 0 and step != 0:
            for c in problem.step_costs:
                error = error + c.cost(controls, states, step)
                if want_tape:
                    cost_hits.append((step, c))
        if step != problem.final_system_eval_step:
            if want_tape:
                states, cache = evolve_step(problem, controls, states, time, True)
                tape.append(cache)
            else:
                states = evolve_step(problem, controls, states, time)
    for c in problem.costs:
        if not c.requires_step_evaluation:
            error = error + c.cost(controls, states, problem.final_system_eval_step)
            if want_tape:
                cost_hits.append((problem.final_system_eval_step, c))
    if want_tape:
        return error, states, (tape, cost_hits)
    return error, states


def evaluate_with_grad(problem, controls):
    """
    Value and gradient of `evaluate` w.r.t. `controls`, in qoc's final convention
    (dC/dRe + i dC/dIm for complex controls, real for real controls) -- what
    ans_jacobian + the conjugation at schroedingerdiscrete.py:318-324 deliver.
    Returns (error, grads, final_states).
    """
    controls = np.asarray(controls)
    error, final_states, (tape, cost_hits) = evaluate(problem, controls, want_tape=True)
    nsteps = problem.system_eval_count - 1
    grads = np.zeros(controls.shape, dtype=np.complex128)

    # cotangent injections, by system step
    hits = {}
    for step, c in cost_hits:
        hits.setdefault(step, []).append(c)
        cb = c.controls_bar(controls, None, step)
        if cb is not None:
            grads = grads + cb

    def states_at(step):
        if step == nsteps:
            return final_states
        return tape[step]["states"]

    lam = np.zeros_like(final_states)
    for c in hits.get(nsteps, []):
        sb = c.states_bar(controls, final_states, nsteps)
        if sb is not None:
            lam = lam + sb

    slope_cache = {}
    for step in range(nsteps - 1, -1, -1):
        cache = tape[step]
        psi = cache["states"]
        # psi' = U psi
        ubar = np.sum(np.matmul(lam, _h(psi)), axis=0)
        mbar = expm_pade_vjp(cache["ecache"], ubar)
        genbars = magnus_combine_vjp(problem.magnus_policy, problem.dt, cache["gens"],
                                     cache["mcache"], mbar)
        for t, abar in zip(cache["times"], genbars):
            hbar = 1j * abar  # a = -i H
            g_re, g_im = problem.hamiltonian_slopes(t)
            ubar_mid = np.zeros(problem.control_count, dtype=np.complex128)
            for k in range(problem.control_count):
                ubar_mid[k] = np.real(np.sum(np.conjugate(hbar) * g_re[k]))
                if problem.complex_controls:
                    ubar_mid[k] += 1j * np.real(np.sum(np.conjugate(hbar) * g_im[k]))
            i1, w1, i2, w2 = interpolation_weights(t, problem.control_eval_times)
            grads[i1] += w1 * ubar_mid
            grads[i2] += w2 * ubar_mid
        lam = np.matmul(_h(cache["unitary"]), lam)
        # step costs are evaluated on the states *before* evolving from `step`
        for c in hits.get(step, []):
            sb = c.states_bar(controls, psi, step)
            if sb is not None:
                lam = lam + sb
    if not problem.complex_controls:
        grads = np.real(grads)
    return error, grads, final_states
