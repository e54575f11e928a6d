"""
CPU tests (no GPU) of the algorithms qoc_amd/csrc/qocx_general.hip adds for Hilbert sizes above 64, restated in NumPy
the way the kernels run them:

* the in-place Gauss-Jordan inversion with partial pivoting BLOCKED by KB pivots (`invert_kb_body`): the KB columns of
  a block are eliminated on a panel of their own, the block's row interchanges are applied to the other columns, and
  every other column takes the block's KB elimination steps as ONE rank-KB update X <- Y + (G - E)(E^T Y); the column
  interchanges are undone at the end from the composed permutation;
* the many-state sweep and K3 as products with the states of a seed as the ROWS of a matrix (`gemm_op` forms):
  Psi' = (Psi Q^T) P^-T, X = Lambda conj(P^-1), Lambda' = X conj(Q), abar = T^T conj(R).

The GPU tests (tests/test_gpu_general.py) hold the kernels themselves to the oracle; these tests hold the algebra.
"""

import numpy as np
import pytest


def blocked_gauss_jordan_inverse(m, kb):
    """In place, as invert_kb_body does it (LAPACK's izamax measure |re| + |im|, ties to the smaller row)."""
    m = np.array(m, dtype=np.complex128)
    n = m.shape[0]
    assert n % kb == 0
    piv = np.arange(n)
    for k0 in range(0, n, kb):
        cols = slice(k0, k0 + kb)
        rest = np.r_[0:k0, k0 + kb:n]
        g = m[:, cols].copy()                      # the panel, eliminated on its own
        for j in range(kb):
            k = k0 + j
            mag = np.abs(g[k:, j].real) + np.abs(g[k:, j].imag)
            p = k + int(np.argmax(mag))            # (argmax returns the first maximum: the smaller row)
            piv[k] = p
            if p != k:
                g[[k, p], :] = g[[p, k], :]        # whole panel rows, earlier columns included
            inv = 1.0 / g[k, j]
            f = g[:, j].copy()
            g[k, :] *= inv
            g[k, j] = inv
            others = np.arange(n) != k
            g[others, :] -= np.outer(f[others], g[k, :])
            g[others, j] = -f[others] * inv
        for j in range(kb):                        # the block's row interchanges on the other columns, in order
            k, p = k0 + j, piv[k0 + j]
            if p != k:
                m[np.ix_([k, p], rest)] = m[np.ix_([p, k], rest)]
        r = m[k0:k0 + kb][:, rest].copy()          # the pivot rows R = E^T Y
        y = m[:, rest]
        y[k0:k0 + kb, :] = 0                       # Y - E (E^T Y) ...
        m[:, rest] = y + g @ r                     # ... + G (E^T Y)
        m[:, cols] = g
    idx = np.arange(n)                             # column c of the inverse: the composed interchanges, last to first
    for k in range(n - 1, -1, -1):
        p = piv[k]
        idx[k], idx[p] = idx[p], idx[k]
    return m[:, idx], piv


@pytest.mark.parametrize("n, kb", [(16, 16), (48, 16), (80, 16), (32, 8), (64, 8)])
def test_blocked_gauss_jordan_equals_the_inverse(n, kb):
    rng = np.random.default_rng(n + kb)
    # a diagonally dominant matrix (Pade denominators of small generators: no interchange) ...
    dom = np.eye(n) * 30 + rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
    out, piv = blocked_gauss_jordan_inverse(dom, kb)
    assert np.array_equal(piv, np.arange(n))
    assert np.max(np.abs(out @ dom - np.eye(n))) < 1e-13
    # ... a general one (interchanges in every block) ...
    gen = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
    out, piv = blocked_gauss_jordan_inverse(gen, kb)
    assert np.count_nonzero(piv != np.arange(n)) > n // 4
    assert np.max(np.abs(out @ gen - np.eye(n))) < 1e-10 * np.linalg.cond(gen)
    # ... and the scaled cyclic shift of tests/test_gpu_general.py, whose pivots all sit below the diagonal
    shift = np.roll(np.eye(n), 1, axis=0) * 5.0 + np.eye(n)
    out, piv = blocked_gauss_jordan_inverse(shift, kb)
    assert np.count_nonzero(piv != np.arange(n)) >= n - 1
    assert np.max(np.abs(out @ shift - np.eye(n))) < 1e-12


def test_blocked_update_is_the_sequence_of_single_steps():
    """One block = its KB single Gauss-Jordan steps with their interchanges: the identity the rank-KB update rests on."""
    rng = np.random.default_rng(5)
    n, kb = 32, 8
    m = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
    ref = m.copy()
    for k in range(kb):                            # unblocked in-place steps on the whole matrix
        mag = np.abs(ref[k:, k].real) + np.abs(ref[k:, k].imag)
        p = k + int(np.argmax(mag))
        ref[[k, p], :] = ref[[p, k], :]
        inv = 1.0 / ref[k, k]
        f = ref[:, k].copy()
        ref[k, :] *= inv
        ref[k, k] = inv
        others = np.arange(n) != k
        ref[others, :] -= np.outer(f[others], ref[k, :])
        ref[others, k] = -f[others] * inv
    # the same through the blocked code, stopped after the first block (no column interchange yet)
    blk = m.copy()
    g = blk[:, :kb].copy()
    piv = np.arange(n)
    for j in range(kb):
        mag = np.abs(g[j:, j].real) + np.abs(g[j:, j].imag)
        p = j + int(np.argmax(mag))
        piv[j] = p
        g[[j, p], :] = g[[p, j], :]
        inv = 1.0 / g[j, j]
        f = g[:, j].copy()
        g[j, :] *= inv
        g[j, j] = inv
        others = np.arange(n) != j
        g[others, :] -= np.outer(f[others], g[j, :])
        g[others, j] = -f[others] * inv
    rest = np.arange(kb, n)
    for j in range(kb):
        if piv[j] != j:
            blk[np.ix_([j, piv[j]], rest)] = blk[np.ix_([piv[j], j], rest)]
    r = blk[:kb][:, rest].copy()
    y = blk[:, rest]
    y[:kb, :] = 0
    blk[:, rest] = y + g @ r
    blk[:, :kb] = g
    assert np.max(np.abs(blk - ref)) < 1e-11 * np.max(np.abs(ref))


def test_many_state_products_are_the_vector_recursions():
    """The row forms of the many-state sweep and K3 against the per-state vector formulas."""
    rng = np.random.default_rng(9)
    n, S, M = 12, 5, 4

    def cm(*shape):
        return rng.standard_normal(shape) + 1j * rng.standard_normal(shape)

    q, pinv, a = cm(n, n), cm(n, n), cm(n, n)
    psi, lam = cm(S, n), cm(S, n)                  # one state / cotangent per ROW
    fwd = (psi @ q.T) @ pinv.T
    x = lam @ pinv.conj()
    lam2 = x @ q.conj()
    for s in range(S):
        assert np.allclose(fwd[s], pinv @ (q @ psi[s]))
        assert np.allclose(x[s], pinv.conj().T @ lam[s])
        assert np.allclose(lam2[s], q.conj().T @ x[s])
    # K3: T_j = T_{j-1} conj(a) (rows: tau_j = a^H tau_{j-1}), R_{i-1} = R_i a^T (rows: a rho_i), abar = T^T conj(R)
    t = [cm(S, n)]
    for _ in range(1, M):
        t.append(t[-1] @ a.conj())
    r = [cm(S, n)]
    for _ in range(1, M):
        r.append(r[-1] @ a.T)
    abar = np.concatenate(t).T @ np.concatenate(r).conj()
    ref = np.zeros((n, n), dtype=np.complex128)
    for i in range(M):
        for s in range(S):
            if i > 0:
                assert np.allclose(t[i][s], a.conj().T @ t[i - 1][s])
                assert np.allclose(r[i][s], a @ r[i - 1][s])
            ref += np.outer(t[i][s], r[i][s].conj())
    assert np.allclose(abar, ref)
    # a skew-Hermitian generator (Hermitian H): a rho = -a^H rho, both chains on a^H (knob general_skew)
    h = cm(n, n)
    ask = -1j * (h + h.conj().T)
    v = cm(n)
    assert np.allclose(ask @ v, -(ask.conj().T @ v))
