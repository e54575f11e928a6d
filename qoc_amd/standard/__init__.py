"""standard - standard definitions (public names of qoc.standard that the hot path needs)."""

from .constants import (SIGMA_MINUS, SIGMA_PLUS, SIGMA_X, SIGMA_Y, SIGMA_Z,
                        get_annihilation_operator, get_creation_operator, get_eij)
from .costs import (ControlArea, ControlBandwidthMax, ControlNorm, ControlVariation,
                    ForbidDensities, ForbidStates, TargetDensityInfidelity,
                    TargetDensityInfidelityTime, TargetStateInfidelity,
                    TargetStateInfidelityTime)
from .functions import (column_vector_list_to_matrix, commutator, conjugate_transpose, expm,
                        krons, matmuls, matrix_to_column_vector_list, rms_norm)
from .optimizers import LBFGSB, SGD, Adam
from .utils import CustomJSONEncoder, generate_save_file_path

__all__ = [
    "get_annihilation_operator", "get_creation_operator", "get_eij",
    "SIGMA_X", "SIGMA_Y", "SIGMA_Z", "SIGMA_MINUS", "SIGMA_PLUS",
    "ControlArea", "ControlBandwidthMax", "ControlNorm", "ControlVariation",
    "ForbidDensities", "ForbidStates", "TargetDensityInfidelity", "TargetDensityInfidelityTime",
    "TargetStateInfidelity", "TargetStateInfidelityTime",
    "commutator", "conjugate_transpose", "expm", "krons", "rms_norm", "matmuls",
    "column_vector_list_to_matrix", "matrix_to_column_vector_list",
    "Adam", "LBFGSB", "SGD",
    "generate_save_file_path", "CustomJSONEncoder",
]
