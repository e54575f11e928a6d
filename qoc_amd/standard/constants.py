"""
constants.py - Pauli matrices and truncated ladder operators
(same names and values as qoc/standard/constants.py:9-65).
"""

import numpy as np

SIGMA_X = np.array(((0, 1), (1, 0)))
SIGMA_Y = np.array(((0, -1j), (1j, 0)))
SIGMA_Z = np.array(((1, 0), (0, -1)))
SIGMA_PLUS = np.array(((0, 1), (0, 0)))
SIGMA_MINUS = np.array(((0, 0), (1, 0)))


def get_creation_operator(size):
    """a^dagger truncated to `size` levels."""
    return np.diag(np.sqrt(np.arange(1, size)), k=-1)


def get_annihilation_operator(size):
    """a truncated to `size` levels."""
    return np.diag(np.sqrt(np.arange(1, size)), k=1)


def get_eij(i, j, size):
    """The (size x size) matrix unit with a one at row i, column j."""
    eij = np.zeros((size, size))
    eij[i, j] = 1
    return eij
