"""
functions.py - small dense helpers with the names of qoc/standard/functions/convenience.py
(:16-104). Plain NumPy: nothing here is traced.
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
