"""
device_model.py - TEST INFRASTRUCTURE: a NumPy model of the algorithm the HIP kernels run
(DESIGN.md "state-space adjoint"), kernel by kernel, so the math can be checked against the
oracle on CPU before (and independently of) the device code:

  K1 pade_factor : a = -i dt H(u_mid) 2^-s ; P = v - u, Q = v + u ; LU(P) with partial pivoting
                   (Pade order 3 / 5 / 7 / 9 / 13 by ||a||_1, see PADE_THETA)
  K2 sweep       : psi' = (P^-1 Q)^(2^s) psi by 2^s solves; costs; lambda; x = P^-H lambda', lambda = Q^H x
  K3 krylov_grad : abar = sum_i tau_i rho_i^H from Krylov chains of a and a^H; g_k = Re<abar, dA/du_k>
  K4 scatter     : grads = W^T g  (transpose of the linear interpolation)

Nothing here is imported by the product.
"""

import numpy as np

from oracle import qoc_numpy as onp

B = onp.PADE_B

# Order selection. The reference always evaluates the [13/13] approximant (expm.py:230-233: the
# loop over PADE_ORDERS has no break, so the last order - 13 - wins whenever the norm is below
# theta_13). The device takes the order from the table the reference cites (Higham 2005,
# Algorithm 2.3; expm.py:194-209 carries its theta_m): for ||a||_1 <= theta_m the [m/m] approximant
# is exp(a + da) with ||da|| <= 2^-53 ||a||, i.e. the same matrix as the [13/13] one to rounding,
# for 2..5 products instead of 6 and an adjoint chain of m instead of 13 terms.
PADE_THETA = {3: 1.495585217958292e-2, 5: 2.539398330063230e-1, 7: 9.504178996162932e-1,
              9: 2.097847961257068, 13: 5.371920351148152}
PADE_COEFFS = {
    3: (120., 60., 12., 1.),
    5: (30240., 15120., 3360., 420., 30., 1.),
    7: (17297280., 8648640., 1995840., 277200., 25200., 1512., 56., 1.),
    9: (17643225600., 8821612800., 2075673600., 302702400., 30270240., 2162160., 110880., 3960.,
        90., 1.),
    13: tuple(B),
}


def pade_order(norm1, policy=0):
    """policy 0: smallest order whose theta covers the norm; 13: always 13 (as the reference runs)."""
    if policy == 13:
        return 13
    for m in (3, 5, 7, 9):
        if norm1 < PADE_THETA[m]:
            return m
    return 13


def coeff_tables(order):
    b = list(PADE_COEFFS[order]) + [0.0] * (14 - len(PADE_COEFFS[order]))
    cu = [b[m] if m % 2 == 1 else 0.0 for m in range(14)]              # u(a) = sum_m cu[m] a^m
    cv = [b[m] if (m % 2 == 0 and m > 0) else 0.0 for m in range(14)]  # v(a) - b0 I
    return b, cu, cv


CU, CV = coeff_tables(13)[1:]


def lu_partial_pivot(p):
    """LAPACK zgetf2 semantics: pivot = first max of |re|+|im| (izamax), returns (lu, perm)."""
    lu = np.array(p, dtype=np.complex128)
    n = lu.shape[0]
    perm = np.arange(n)
    for k in range(n):
        col = np.abs(lu[k:, k].real) + np.abs(lu[k:, k].imag)
        piv = k + int(np.argmax(col))
        if piv != k:
            lu[[k, piv]] = lu[[piv, k]]
            perm[[k, piv]] = perm[[piv, k]]
        lu[k + 1:, k] = lu[k + 1:, k] * (1.0 / lu[k, k])
        lu[k + 1:, k + 1:] -= np.outer(lu[k + 1:, k], lu[k, k + 1:])
    return lu, perm


def solve_lu(lu, perm, y):
    """x = P^-1 y for (n x S) y."""
    n = lu.shape[0]
    z = np.array(y[perm], dtype=np.complex128)
    for k in range(n):
        z[k + 1:] -= np.outer(lu[k + 1:, k], z[k])
    for k in range(n - 1, -1, -1):
        z[k] = z[k] / lu[k, k]
        z[:k] -= np.outer(lu[:k, k], z[k])
    return z


def solve_lu_adjoint(lu, perm, lam):
    """x = P^-H lam."""
    n = lu.shape[0]
    w = np.array(lam, dtype=np.complex128)
    # U^H w = lam  (U^H lower triangular)
    for k in range(n):
        w[k] = w[k] / np.conj(lu[k, k])
        w[k + 1:] -= np.outer(np.conj(lu[k, k + 1:]), w[k])
    # L^H z = w (unit upper)
    for k in range(n - 1, -1, -1):
        w[:k] -= np.outer(np.conj(lu[k, :k]), w[k])
    x = np.empty_like(w)
    x[perm] = w
    return x


def pade_uv(a_s, order):
    """u, v of the [order/order] approximant, products in the order the kernels form them."""
    n = a_s.shape[0]
    b = PADE_COEFFS[order]
    eye = np.eye(n)
    a2 = a_s @ a_s
    if order == 3:
        return a_s @ (b[3] * a2) + b[1] * a_s, b[2] * a2 + b[0] * eye
    a4 = a2 @ a2
    if order == 5:
        return a_s @ (b[5] * a4 + b[3] * a2) + b[1] * a_s, b[4] * a4 + b[2] * a2 + b[0] * eye
    a6 = a2 @ a4
    if order == 7:
        return (a_s @ (b[7] * a6 + b[5] * a4 + b[3] * a2) + b[1] * a_s,
                b[6] * a6 + b[4] * a4 + b[2] * a2 + b[0] * eye)
    if order == 9:
        a8 = a2 @ a6
        return (a_s @ (b[9] * a8 + b[7] * a6 + b[5] * a4 + b[3] * a2) + b[1] * a_s,
                b[8] * a8 + b[6] * a6 + b[4] * a4 + b[2] * a2 + b[0] * eye)
    w2 = a6 @ (b[13] * a6 + b[11] * a4 + b[9] * a2) + b[7] * a6 + b[5] * a4 + b[3] * a2
    u = a_s @ w2 + b[1] * a_s
    v = a6 @ (b[12] * a6 + b[10] * a4 + b[8] * a2) + b[6] * a6 + b[4] * a4 + b[2] * a2 + b[0] * eye
    return u, v


def pade_factor(a, policy=0, order=None):
    norm1 = onp.one_norm(a)
    s = onp.pade_scale_count(norm1)
    a_s = a if norm1 < onp.THETA_13 else a * (2 ** -s)
    # (the kernels pick the order from an upper bound of the norm, sum |re| + |im|: never a
    # lower order than this, sometimes a higher one - any order whose theta covers the norm gives
    # the same matrix to rounding; `order` forces the one a kernel reported)
    if order is None:
        order = pade_order(norm1, policy)
    u, v = pade_uv(a_s, order)
    lu, perm = lu_partial_pivot(v - u)
    return dict(s=s, a=a_s, q=v + u, lu=lu, perm=perm, order=order)


def krylov_abar(a_s, triples, order=13):
    """abar (for the scaled generator) from sub-step triples (x, psi, psi_next), each n x S."""
    n = a_s.shape[0]
    ah = a_s.conj().T
    _, cu, cv = coeff_tables(order)
    abar = np.zeros((n, n), dtype=np.complex128)
    for x, psi, psi_next in triples:
        sig = [psi + psi_next]
        dlt = [psi - psi_next]
        tau = [x]
        for _ in range(order - 1):
            sig.append(a_s @ sig[-1])
            dlt.append(a_s @ dlt[-1])
            tau.append(ah @ tau[-1])
        for i in range(order):
            rho = np.zeros_like(x)
            for j in range(order - i):
                rho = rho + cu[i + j + 1] * sig[j] + cv[i + j + 1] * dlt[j]
            abar = abar + tau[i] @ rho.conj().T
    return abar


def krylov_abar_horner(a_s, triples, order=13):
    """
    The same abar in the order the device kernel (K3) works: tau chain first, then the rho_i by
    the Horner recurrence rho_12 = b13 sigma, rho_{i-1} = b_i w_i + a rho_i (w_i = sigma for odd i,
    delta for even i), each followed by its rank-1 update - 24 matrix-vector products instead
    of 36.
    """
    n = a_s.shape[0]
    ah = a_s.conj().T
    b = coeff_tables(order)[0]
    abar = np.zeros((n, n), dtype=np.complex128)
    for x, psi, psi_next in triples:
        sig, dlt = psi + psi_next, psi - psi_next
        tau = [x]
        for _ in range(order - 1):
            tau.append(ah @ tau[-1])
        rho = b[order] * sig  # order odd: w_order = sigma
        for i in range(order - 1, -1, -1):
            abar = abar + tau[i] @ rho.conj().T
            if i > 0:
                rho = b[i] * (sig if i % 2 else dlt) + a_s @ rho
    return abar


def evaluate_with_grad(problem, controls, pade_policy=0):
    """Model of the whole device path for MagnusPolicy.M2. Returns (error, grads, final_states)."""
    assert problem.magnus_policy == "M2"
    controls = np.asarray(controls)
    n_steps = problem.system_eval_count - 1
    dt = problem.dt
    xs = problem.control_eval_times
    psi = problem.initial_states[:, :, 0].T.copy()  # n x S
    n, S = psi.shape
    factors, substates = [], []
    error = 0.0
    hits = {}
    for step in range(problem.system_eval_count):
        as_states = psi.T[:, :, None]
        if step % problem.cost_eval_step == 0 and step != 0:
            for c in problem.step_costs:
                error = error + c.cost(controls, as_states, step)
                hits.setdefault(step, []).append(c)
        if step == n_steps:
            break
        t_mid = step * dt + dt * 0.5
        a = dt * (-1j * problem.hamiltonian(onp.interpolate_linear_set(t_mid, xs, controls), t_mid))
        f = pade_factor(a, pade_policy)
        factors.append(f)
        subs = [psi]
        for _ in range(2 ** f["s"]):
            subs.append(solve_lu(f["lu"], f["perm"], f["q"] @ subs[-1]))
        substates.append(subs)
        psi = subs[-1]
    final_states = psi.T[:, :, None]
    grads = np.zeros(controls.shape, dtype=np.complex128)
    for c in problem.costs:
        if not c.requires_step_evaluation:
            error = error + c.cost(controls, final_states, n_steps)
            hits.setdefault(n_steps, []).append(c)
            cb = c.controls_bar(controls, None, n_steps)
            if cb is not None:
                grads = grads + cb
    lam = np.zeros((n, S), dtype=np.complex128)
    for c in hits.get(n_steps, []):
        sb = c.states_bar(controls, final_states, n_steps)
        if sb is not None:
            lam = lam + sb[:, :, 0].T
    for step in range(n_steps - 1, -1, -1):
        f, subs = factors[step], substates[step]
        triples = []
        for m in range(2 ** f["s"] - 1, -1, -1):
            x = solve_lu_adjoint(f["lu"], f["perm"], lam)
            lam = f["q"].conj().T @ x
            triples.append((x, subs[m], subs[m + 1]))
        abar = krylov_abar_horner(f["a"], triples, f["order"]) * (2 ** -f["s"])
        t_mid = step * dt + dt * 0.5
        g_re, g_im = problem.hamiltonian_slopes(t_mid)
        ubar = np.zeros(problem.control_count, dtype=np.complex128)
        for k in range(problem.control_count):
            ubar[k] = np.real(np.sum(np.conj(abar) * (-1j * dt * g_re[k])))
            if problem.complex_controls:
                ubar[k] += 1j * np.real(np.sum(np.conj(abar) * (-1j * dt * g_im[k])))
        i1, w1, i2, w2 = onp.interpolation_weights(t_mid, xs)
        grads[i1] += w1 * ubar
        grads[i2] += w2 * ubar
        for c in hits.get(step, []):
            sb = c.states_bar(controls, subs[0].T[:, :, None], step)
            if sb is not None:
                lam = lam + sb[:, :, 0].T
    if not problem.complex_controls:
        grads = np.real(grads)
    return error, grads, final_states
