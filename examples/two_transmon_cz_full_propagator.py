"""
two_transmon_cz_full_propagator.py - a problem beyond Hilbert size 64: two coupled transmons with ten levels each
(n = 100), a controlled-Z gate on the computational subspace optimised as a FULL propagator (every basis state is
propagated: 100 states of dimension 100), with forbidden leakage - the kind of script the reference runs unchanged
for any size (qoc/core/schroedingerdiscrete.py:356-502). On qoc_amd it takes the general path of
qoc_amd/csrc/qocx_general.hip: matrices in HBM, products and the blocked inversion on the matrix cores, the sweep
and the Krylov adjoint as products over the states.

    python examples/two_transmon_cz_full_propagator.py      # needs an MI355X and qoc_amd/libqocx.so

(The ten-level anharmonic terms make ||dt H||_1 large at this step size: every step takes 2^4..2^5 squaring
sub-steps, each of them two products over the 100 states - ~0.3 s per GRAPE iteration)
"""

import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # run from a checkout
from qoc_amd import grape_schroedinger_discrete  # noqa: E402
from qoc_amd.standard import (Adam, TargetStateInfidelity, get_annihilation_operator,  # noqa: E402
                              get_creation_operator)

LEVELS = 10
N = LEVELS * LEVELS
A = get_annihilation_operator(LEVELS)
AD = get_creation_operator(LEVELS)
EYE = np.eye(LEVELS)
A1, A2 = np.kron(A, EYE), np.kron(EYE, A)
A1D, A2D = np.kron(AD, EYE), np.kron(EYE, AD)
ALPHA, COUPLING, DETUNING = -0.25 * 2 * np.pi, 0.01 * 2 * np.pi, 0.1 * 2 * np.pi
H_SYSTEM = (DETUNING * A2D @ A2 + 0.5 * ALPHA * (A1D @ A1D @ A1 @ A1 + A2D @ A2D @ A2 @ A2)
            + COUPLING * (A1D @ A2 + A1 @ A2D))
DRIVES = [A1 + A1D, A2 + A2D, A2D @ A2]   # x drives and a flux-like detuning of the second transmon


def hamiltonian(controls, time_):
    return H_SYSTEM + controls[0] * DRIVES[0] + controls[1] * DRIVES[1] + controls[2] * DRIVES[2]


def index(i, j):
    return i * LEVELS + j


# every basis state is an initial state; the targets of the four computational states carry the CZ phases, the others
# are asked to come back to themselves (a unitary on the whole space that is CZ on the qubit subspace)
INITIAL_STATES = np.eye(N, dtype=np.complex128)[:, :, None]
TARGET_STATES = INITIAL_STATES.copy()
TARGET_STATES[index(1, 1), index(1, 1), 0] = -1
EVOLUTION_TIME = 60.0  # nanoseconds
EVAL_COUNT = 121


def main(iteration_count=20):
    rng = np.random.default_rng(0)
    initial_controls = 0.02 * rng.standard_normal((EVAL_COUNT, 3))
    start = time.perf_counter()
    result = grape_schroedinger_discrete(
        3, EVAL_COUNT, [TargetStateInfidelity(TARGET_STATES)], EVOLUTION_TIME, hamiltonian, INITIAL_STATES,
        EVAL_COUNT, initial_controls=initial_controls, iteration_count=iteration_count, log_iteration_step=5,
        optimizer=Adam(learning_rate=2e-3), max_control_norms=np.full(3, 0.3))
    wall = time.perf_counter() - start
    print("best error {:.6f} at iteration {} - {:.0f} ms per GRAPE iteration ({} states of dimension {}, {} steps)".format(
        result.best_error, result.best_iteration, 1e3 * wall / iteration_count, N, N, EVAL_COUNT - 1))
    return result


if __name__ == "__main__":
    main()
