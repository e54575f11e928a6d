"""
transmon_pi.py - the reference's first example (examples/0_transmon_pi.py: drive a two-level
transmon from |0> to |1> with one complex control) written against qoc_amd: only the import
lines differ from a script written for qoc.

    python examples/transmon_pi.py            # needs an MI355X and qoc_amd/libqocx.so
"""

import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # run from a checkout
from qoc_amd import grape_schroedinger_discrete  # noqa: E402
from qoc_amd.standard import (SIGMA_Z, Adam, TargetStateInfidelity,  # noqa: E402
                              get_annihilation_operator, get_creation_operator)

HILBERT_SIZE = 2
A = get_annihilation_operator(HILBERT_SIZE)
A_DAGGER = get_creation_operator(HILBERT_SIZE)
H_SYSTEM = SIGMA_Z / 2


def hamiltonian(controls, time):
    return H_SYSTEM + controls[0] * A + np.conjugate(controls[0]) * A_DAGGER


INITIAL_STATES = np.stack((np.array([[1], [0]]),))
TARGET_STATES = np.stack((np.array([[0], [1]]),))
EVOLUTION_TIME = 10  # nanoseconds
EVAL_COUNT = EVOLUTION_TIME + 1


def main():
    result = grape_schroedinger_discrete(
        1, EVAL_COUNT, [TargetStateInfidelity(TARGET_STATES)], EVOLUTION_TIME, hamiltonian,
        INITIAL_STATES, EVAL_COUNT, complex_controls=True, iteration_count=300,
        log_iteration_step=25, optimizer=Adam(learning_rate=1e-2))
    print("best error {:.3e} at iteration {}".format(result.best_error, result.best_iteration))
    return result


if __name__ == "__main__":
    main()
