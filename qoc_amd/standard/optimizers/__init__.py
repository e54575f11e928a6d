"""optimizers - Optimizer plugins (same names as qoc.standard.optimizers)."""

from .adam import Adam
from .lbfgsb import LBFGSB
from .sgd import SGD

__all__ = ["Adam", "LBFGSB", "SGD"]
