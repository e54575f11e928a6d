"""
structure.py - turn the user's Hamiltonian callable into the structured form the device needs.

The reference calls `hamiltonian(controls, time)` (arbitrary autograd-traceable Python) inside
its time loop (qoc/core/schroedingerdiscrete.py:483-486). A GPU kernel cannot call Python, so
the host samples the callable once per problem at every quadrature time and extracts

    H(u, t) = H0(t) + sum_k Re(u_k) G_k(t) + Im(u_k) G'_k(t)

by probing u = 0, e_k, i e_k, then VERIFIES real-linearity at a random control value. Every
Hamiltonian in the reference's examples and tests has this form (examples/0_transmon_pi.py:24-26,
tests/test_core.py:529-531, :582). A callable that fails the check is rejected loudly - there is
no CPU fallback.
"""

import numpy as np


class NonLinearHamiltonianError(ValueError):
    pass


def probe_hamiltonian(hamiltonian, hilbert_size, control_count, complex_controls, times,
                      rtol=1e-9):
    """
    Returns (h0, g): h0 :: (nt, n, n), g :: (nt, Kr, n, n) with Kr = K (real controls) or 2K
    (complex: [Re_0, Im_0, Re_1, Im_1, ...]); nt == 1 when nothing depends on time.
    """
    n, k = hilbert_size, control_count
    kr = k * (2 if complex_controls else 1)
    dtype = np.complex128 if complex_controls else np.float64
    rng = np.random.default_rng(12345)
    h0 = np.empty((len(times), n, n), dtype=np.complex128)
    g = np.empty((len(times), kr, n, n), dtype=np.complex128)
    for ti, t in enumerate(times):
        if k == 0:
            h0[ti] = np.asarray(hamiltonian(None, t), dtype=np.complex128)
            continue
        zero = np.zeros(k, dtype=dtype)
        base = np.asarray(hamiltonian(zero, t), dtype=np.complex128)
        if base.shape != (n, n):
            raise ValueError("hamiltonian returned shape {}, expected {}".format(base.shape, (n, n)))
        h0[ti] = base
        for j in range(k):
            unit = zero.copy()
            unit[j] = 1
            slope = np.asarray(hamiltonian(unit, t), dtype=np.complex128) - base
            if complex_controls:
                g[ti, 2 * j] = slope
                unit = zero.copy()
                unit[j] = 1j
                g[ti, 2 * j + 1] = np.asarray(hamiltonian(unit, t), dtype=np.complex128) - base
            else:
                g[ti, j] = slope
        # real-linearity check at a random point
        u = rng.standard_normal(k)
        if complex_controls:
            u = u + 1j * rng.standard_normal(k)
        direct = np.asarray(hamiltonian(u.astype(dtype), t), dtype=np.complex128)
        model = base.copy()
        for j in range(k):
            if complex_controls:
                model = model + u[j].real * g[ti, 2 * j] + u[j].imag * g[ti, 2 * j + 1]
            else:
                model = model + u[j] * g[ti, j]
        scale = max(1.0, float(np.max(np.abs(direct))))
        if np.max(np.abs(direct - model)) > rtol * scale:
            raise NonLinearHamiltonianError(
                "hamiltonian(controls, time) is not real-linear in the controls at time {}: the "
                "MI355X engine needs H = H0(t) + sum_k Re(u_k) G_k(t) + Im(u_k) G'_k(t) "
                "(deviation {:.3e}). There is no CPU fallback.".format(
                    t, float(np.max(np.abs(direct - model)))))
    if len(times) > 1 and np.all(h0 == h0[0]) and np.all(g == g[0]):
        h0, g = h0[:1], g[:1]
    return h0, g


class TimeDependentSystemError(NotImplementedError):
    pass


# ---- opaque Hamiltonians: the host samples the step generators itself --------------------------

def interpolation_rows(evolution_time, control_eval_count, times):
    """
    Linear interpolation of the controls at `times`, the reference's rule
    (qoc/core/mathmethods.py:36-67: the two lowest / highest knots beyond the ends, else the
    first knot >= t and its predecessor). Returns (i1, i2, x1, x2): index and abscissa arrays.
    """
    xs = np.linspace(0, evolution_time, control_eval_count)
    i1 = np.empty(len(times), dtype=np.int64)
    i2 = np.empty(len(times), dtype=np.int64)
    for q, t in enumerate(times):
        if t <= xs[0]:
            i1[q], i2[q] = 0, 1
        elif t >= xs[-1]:
            i1[q], i2[q] = control_eval_count - 2, control_eval_count - 1
        else:
            index = int(np.argmax(t <= xs))
            i1[q], i2[q] = index - 1, index
    return i1, i2, xs[i1], xs[i2]


def controls_at(controls, rows, times):
    """(len(times), K) controls at `times`: y1 + ((y2 - y1) / (x2 - x1)) * (x3 - x1), the
    reference's operation order (mathmethods.py:33)."""
    i1, i2, x1, x2 = rows
    y1, y2 = controls[i1], controls[i2]
    return y1 + (((y2 - y1) / (x2 - x1)[:, None]) * (np.asarray(times) - x1)[:, None])


def sample_generators(hamiltonian, controls, rows, times, dt, hilbert_size):
    """M_j = dt * (-1j * hamiltonian(u(t_j), t_j)) for one control array (magnus_m2,
    mathmethods.py:72-93 with a(t) = -1j H, schroedingerdiscrete.py:483-486)."""
    u = controls_at(controls, rows, times)
    out = np.empty((len(times), hilbert_size, hilbert_size), dtype=np.complex128)
    for j, t in enumerate(times):
        out[j] = dt * (-1j * np.asarray(hamiltonian(u[j], t), dtype=np.complex128))
    return out, u


_FD_H = 1e-4


def hamiltonian_slopes(hamiltonian, u, t, complex_controls):
    """
    d H / d Re(u_k) (and d H / d Im(u_k) for complex controls) at (u, t) by 4th-order central
    differences of the user's callable (the reference gets them from autograd's trace of it):
    truncation ~h^4 |H^(5)|, round-off ~1e-16 |H| / h = 1e-12 |H| at h = 1e-4 max(1, |u_k|).
    Returns [(k, direction, dH)] with direction 1 or 1j.
    """
    out = []
    u = np.array(u)
    for k in range(len(u)):
        for direction in ((1.0, 1.0j) if complex_controls else (1.0,)):
            h = _FD_H * max(1.0, abs(u[k]))

            def at(step):
                v = u.copy()
                v[k] = v[k] + step * direction
                return np.asarray(hamiltonian(v, t), dtype=np.complex128)
            dh = (8.0 * (at(h) - at(-h)) - (at(2 * h) - at(-2 * h))) / (12.0 * h)
            out.append((k, direction, dh))
    return out


def linearize_hamiltonian(hamiltonian, controls, evolution_time, times, hilbert_size,
                          complex_controls):
    """
    The tangent of a hamiltonian(controls, time) that is NOT linear in the controls, at the
    control array `controls` (Nc x K): at every time t of `times`

        H_lin(v, t) = H0'(t) + sum_k Re(v_k) G'_{re,k}(t) + Im(v_k) G'_{im,k}(t),
        G'(t) = d H / d u at (u(t), t)   (4th-order central differences of the callable),
        H0'(t) = H(u(t), t) - sum_k Re(u_k(t)) G'_{re,k}(t) + Im(u_k(t)) G'_{im,k}(t),

    with u(t) the reference's linear interpolation of `controls`. H_lin(u(t), t) = H(u(t), t) and
    d H_lin / d v = d H / d u there, so a structured, time-dependent problem built from (H0', G')
    and evaluated AT `controls` has the cost and the control gradient of the original one - under
    any Magnus policy and on the Lindblad path (the reference differentiates the callable itself
    with autograd, schroedingerdiscrete.py:483-497, lindbladdiscrete.py:486-489).
    Returns (h0 (nt, n, n), g (nt, Kr, n, n)), Kr = K or 2 K ([Re_0, Im_0, Re_1, ...]).
    """
    controls = np.asarray(controls)
    k = controls.shape[1]
    kr = k * (2 if complex_controls else 1)
    rows = interpolation_rows(evolution_time, controls.shape[0], times)
    u = controls_at(controls, rows, times)
    h0 = np.empty((len(times), hilbert_size, hilbert_size), dtype=np.complex128)
    g = np.empty((len(times), kr, hilbert_size, hilbert_size), dtype=np.complex128)
    for ti, t in enumerate(times):
        base = np.asarray(hamiltonian(u[ti], t), dtype=np.complex128)
        if base.shape != (hilbert_size, hilbert_size):
            raise ValueError("hamiltonian returned shape {}, expected {}".format(
                base.shape, (hilbert_size, hilbert_size)))
        for slot, (j, direction, dh) in enumerate(hamiltonian_slopes(hamiltonian, u[ti], t,
                                                                     complex_controls)):
            g[ti, slot] = dh
            base = base - (u[ti][j].real if direction == 1.0 else u[ti][j].imag) * dh
        h0[ti] = base
    return h0, g


def generator_gradients(hamiltonian, controls, rows, times, dt, gen_bars, complex_controls):
    """
    d cost / d controls from the generator cotangents Mbar_j (qocx_download_generator_cotangents):
    d cost / d u_k(t_j) = Re sum conj(Mbar_j) * d M_j / d u_k with M_j = -1j dt H, carried to the
    control grid by the transpose of the linear interpolation. Complex controls: qoc's
    convention d/dRe + i d/dIm.
    """
    i1, i2, x1, x2 = rows
    u = controls_at(controls, rows, times)
    grads = np.zeros(controls.shape, dtype=np.complex128 if complex_controls else np.float64)
    for j, t in enumerate(times):
        w2 = (t - x1[j]) / (x2[j] - x1[j])
        for k, direction, dh in hamiltonian_slopes(hamiltonian, u[j], t, complex_controls):
            value = float(np.real(np.sum(np.conj(gen_bars[j]) * (-1j * dt * dh))))
            grads[i1[j], k] += (1.0 - w2) * value * direction
            grads[i2[j], k] += w2 * value * direction
    return grads


def decision_times(evolution_time, count=257):
    """Fallback probe times for the time-dependence decision when the caller has no integrator
    grid: `count` points of [0, T] displaced by a golden-ratio sequence, so that no drive whose
    period divides T (or T / (count - 1)) aliases to a constant."""
    phi = 0.6180339887498949
    return [evolution_time * min(1.0, (q + (q * phi) % 1.0) / count) for q in range(count)]


def probe_static_lindblad_system(hamiltonian, lindblad_data, hilbert_size, control_count,
                                 complex_controls, evolution_time, probe_times=None):
    """
    Structure of the Lindblad path's inputs (qoc/core/lindbladdiscrete.py:444-493):
    hamiltonian(controls, time) and lindblad_data(time) -> (dissipators, operators).
    Returns (h0 (n, n), g (Kr, n, n), dissipators (L,) or None, operators (L, n, n) or None,
    time_dependent). Explicit time dependence of the Hamiltonian is decided on `probe_times` -
    the caller passes the integrator's own stage-time grid (every time the device would ever
    read H at for the coarsest sub-division), so a periodic drive cannot alias to a constant - by
    comparing H(0, t) and H(u_random, t) with their values at the first time, bit for bit. A
    time-dependent Hamiltonian is then sampled at the stage times (sample_lindblad_hamiltonian),
    and so is a time-dependent lindblad_data (sample_lindblad_data; the reference calls both at
    every right-hand side, lindbladdiscrete.py:483-492). `time_dependent` covers either; the
    function attribute `lindblad_time_dependent` says whether lindblad_data was the (or a) cause.
    """
    n = hilbert_size
    times = list(decision_times(evolution_time) if probe_times is None else probe_times)
    kr = control_count * (2 if complex_controls else 1)
    time_dependent = False
    if hamiltonian is None:
        h0 = np.zeros((1, n, n), dtype=np.complex128)
        g = np.zeros((1, kr, n, n), dtype=np.complex128)
    else:
        h0, g = probe_hamiltonian(hamiltonian, n, control_count, complex_controls, times[:1])
        if control_count == 0:
            trial = None
        else:
            rng = np.random.default_rng(54321)
            trial = rng.standard_normal(control_count)
            if complex_controls:
                trial = trial + 1j * rng.standard_normal(control_count)
            zero = np.zeros(control_count, dtype=trial.dtype)
            at_trial = np.asarray(hamiltonian(trial, times[0]), dtype=np.complex128)
        for t in times[1:]:
            if control_count == 0:
                same = np.array_equal(np.asarray(hamiltonian(None, t), dtype=np.complex128), h0[0])
            else:
                same = (np.array_equal(np.asarray(hamiltonian(zero, t), dtype=np.complex128), h0[0])
                        and np.array_equal(np.asarray(hamiltonian(trial, t), dtype=np.complex128),
                                           at_trial))
            if not same:
                time_dependent = True
                break
    dissipators, operators = None, None
    lindblad_time_dependent = False
    if lindblad_data is not None:
        first = lindblad_data(times[0])
        if first[0] is not None and first[1] is not None:
            dissipators = np.asarray(first[0], dtype=np.float64)
            operators = np.asarray(first[1], dtype=np.complex128)
            if operators.shape != (dissipators.shape[0], n, n):
                raise ValueError("lindblad_data returned operators of shape {}, expected {}"
                                 "".format(operators.shape, (dissipators.shape[0], n, n)))
            if np.any(np.iscomplex(np.asarray(first[0]))):
                raise ValueError("lindblad_data returned complex dissipators")
        for t in times[1:]:
            d, o = lindblad_data(t)
            same = ((d is None) == (dissipators is None)) and (
                dissipators is None
                or (np.array_equal(np.asarray(d, dtype=np.float64), dissipators)
                    and np.array_equal(np.asarray(o, dtype=np.complex128), operators)))
            if not same:
                if dissipators is None or d is None:
                    raise ValueError("lindblad_data(time) returns data at some times and None at "
                                     "others")
                # explicit time dependence: sampled at the integrator's stage times, like the
                # Hamiltonian (sample_lindblad_data)
                time_dependent = True
                lindblad_time_dependent = True
                break
    probe_static_lindblad_system.lindblad_time_dependent = lindblad_time_dependent
    return h0[0], g[0], dissipators, operators, time_dependent


def sample_lindblad_hamiltonian(hamiltonian, hilbert_size, control_count, complex_controls, times):
    """(h0 (nt, n, n), g (nt, Kr, n, n) or None if constant) at the given stage times."""
    h0, g = probe_hamiltonian(hamiltonian, hilbert_size, control_count, complex_controls,
                              list(times))
    nt = len(times)
    if h0.shape[0] == 1:  # turned out constant on this grid
        h0 = np.repeat(h0, nt, axis=0)
        g = np.repeat(g, nt, axis=0)
    g_constant = bool(np.all(g == g[:1]))
    return h0, (None if g_constant else g)


def sample_lindblad_data(lindblad_data, hilbert_size, times):
    """(dissipators (nt, L), operators (nt, L, n, n)) at the given stage times."""
    diss, ops = [], []
    for t in times:
        d, o = lindblad_data(t)
        diss.append(np.asarray(d, dtype=np.float64))
        ops.append(np.asarray(o, dtype=np.complex128))
    diss, ops = np.stack(diss), np.stack(ops)
    if ops.shape[1:] != (diss.shape[1], hilbert_size, hilbert_size):
        raise ValueError("lindblad_data returned operators of shape {}".format(ops.shape[1:]))
    return diss, ops


def lindblad_subdivision(h0_norm, g_norms, control_bounds, dissipators, operators, dt,
                         margin=1.25, max_phase=0.4):
    """Sub-division count for a time-dependent Hamiltonian: the engine's per-seed rule
    (||Liouvillian||_2 * piece <= max_phase, with ||.||_2 bounded by 2 ||H||_2 + 2 sum gamma
    ||L||_2^2) evaluated for the largest controls, with a margin for the norm of H between the
    probe times. h0_norm / g_norms are spectral norms."""
    bound = h0_norm + sum(b * g for b, g in zip(control_bounds, g_norms))
    diss = 0.0
    if dissipators is not None:
        for gm, op in zip(dissipators, operators):
            diss += abs(gm) * np.linalg.norm(op, 2) ** 2
    return max(1, int(np.ceil(margin * (2 * bound + 2 * diss) * abs(dt) / max_phase)))


def to_real_controls(controls, complex_controls):
    """(..., K) controls -> (..., Kr) float64 in the device ordering."""
    controls = np.asarray(controls)
    if not complex_controls:
        return np.ascontiguousarray(controls, dtype=np.float64)
    out = np.empty(controls.shape[:-1] + (2 * controls.shape[-1],), dtype=np.float64)
    out[..., 0::2] = controls.real
    out[..., 1::2] = controls.imag
    return out


def from_real_gradients(grads, complex_controls):
    """Inverse mapping for gradients: qoc's convention d/dRe + i d/dIm for complex controls."""
    if not complex_controls:
        return grads
    return grads[..., 0::2] + 1j * grads[..., 1::2]
