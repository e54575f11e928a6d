"""
functions.py - small dense helpers with the names of qoc/standard/functions/convenience.py
(:16-104), plain NumPy (nothing here is traced), and `expm`, the public name of the reference's
Pade-13 matrix exponential (qoc/standard/functions/expm.py:210-252), evaluated by the HIP engine.
"""

from functools import reduce

import numpy as np


def commutator(a, b):
    return np.matmul(a, b) - np.matmul(b, a)


def conjugate_transpose(matrix):
    return np.conjugate(np.swapaxes(matrix, -1, -2))


def krons(*matrices):
    return reduce(np.kron, matrices)


def matmuls(*matrices):
    return reduce(np.matmul, matrices)


def rms_norm(array):
    square_norm = np.sum(array * np.conjugate(array))
    return np.sqrt(square_norm / np.prod(np.shape(array)))


def column_vector_list_to_matrix(column_vector_list):
    return np.hstack(column_vector_list)


def matrix_to_column_vector_list(matrix):
    return np.stack([np.vstack(matrix[:, i]) for i in range(matrix.shape[1])])


_EXPM_ENGINE = []


def expm(a):
    """
    Matrix exponential of one square complex matrix (n <= 1024, the engine's size limit) by the
    engine's Pade scaling-and-squaring path - the same kernels the propagation uses: exp(a) is the
    one-step propagator of H = i a over dt = 1 applied to the identity columns (the engine takes the
    Pade order from the norm, [13/13] with squarings for large ones; qoc/standard/functions/expm.py:210-252).
    For 33 <= n <= 64 the engine propagates at most 13 states at a time, above that (its general path)
    at most 64, so the columns go in blocks.
    There is no CPU implementation in this package; without the HIP library / a GPU the call raises.
    """
    from qoc_amd.core import device
    a = np.asarray(a, dtype=np.complex128)
    if a.ndim != 2 or a.shape[0] != a.shape[1]:
        raise ValueError("expm expects one square matrix, got shape {}".format(a.shape))
    n = a.shape[0]
    if n > 1024:
        raise NotImplementedError("expm on the MI355X engine handles n <= 1024 (got {})".format(n))
    if not _EXPM_ENGINE:
        _EXPM_ENGINE.append(device.make_backend())
    engine = _EXPM_ENGINE[0]
    block = n if n <= 32 else (13 if n <= 64 else 64)
    out = np.empty((n, n), dtype=np.complex128)
    eye = np.eye(n, dtype=np.complex128)
    for c0 in range(0, n, block):
        cols = eye[c0:c0 + block]
        engine.set_schroedinger_problem(n, len(cols), 0, 0, 2, 1.0, (1j * a)[None], None, cols, costs=())
        engine.upload_controls(1)
        engine.eval_resident(False)
        _, _, final = engine.download_results(want_grad=False)
        out[:, c0:c0 + block] = final[0].T  # final[s] = U e_s is column s of U
    return out
