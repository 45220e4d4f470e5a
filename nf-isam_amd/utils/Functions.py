"""utils.Functions — the one helper of the reference's `utils/Functions.py` that sits on the hot
path: angle wrapping (reference: src/utils/Functions.py:20-21)."""
import numpy as np

_TWO_PI = 2 * np.pi


def theta_to_pipi(theta):
    """Wrap angles into [-pi, pi).  Works on numpy arrays and torch tensors alike."""
    return (theta + np.pi) % _TWO_PI - np.pi
