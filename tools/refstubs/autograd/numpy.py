"""autograd.numpy stand-in: plain NumPy (forward evaluation only)."""
from numpy import *  # noqa: F401,F403
from numpy import linalg, fft  # noqa: F401
