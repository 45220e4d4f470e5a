"""geometry.TwoDimension — SE(2) poses (reference: src/geometry/TwoDimension.py:303-541).

Two layers: batched numpy functions on [n, 3] arrays (x, y, theta) — what the clique samplers use,
replacing the reference's per-sample Python loops over SE2Pose objects (e.g.
src/factors/Factors.py:1223-1229) — and a small `SE2Pose` value class with the reference's
constructor / operators for host code that handles single poses.
"""
import math

import numpy as np

_TWO_PI = 2.0 * np.pi


def wrap_pi(theta):
    return (theta + np.pi) % _TWO_PI - np.pi


def se2_exp(v):
    """Exponential map se(2) -> SE(2), batched.  v: [n, 3] = (vx, vy, w).
    t = V(w) [vx, vy],  V = [[sin w, -(1-cos w)], [1-cos w, sin w]] / w   (identity for |w| < 1e-10)."""
    v = np.asarray(v, dtype=np.float64)
    w = v[:, 2]
    small = np.abs(w) < 1e-10
    ws = np.where(small, 1.0, w)
    a = np.where(small, 1.0, np.sin(ws) / ws)
    b = np.where(small, 0.0, (1.0 - np.cos(ws)) / ws)
    out = np.empty_like(v)
    out[:, 0] = a * v[:, 0] - b * v[:, 1]
    out[:, 1] = b * v[:, 0] + a * v[:, 1]
    out[:, 2] = w
    return out


def se2_log(p):
    """Logarithmic map SE(2) -> se(2), batched; inverse of `se2_exp` for theta in [-pi, pi)."""
    p = np.asarray(p, dtype=np.float64)
    w = p[:, 2]
    small = np.abs(w) < 1e-10
    ws = np.where(small, 1.0, w)
    half = 0.5 * ws
    # V^{-1} = [[c, half*... ]]: with A = sin w / w, B = (1-cos w)/w:  V^{-1} = [[A, B], [-B, A]] / (A^2 + B^2)
    a = np.where(small, 1.0, np.sin(ws) / ws)
    b = np.where(small, 0.0, (1.0 - np.cos(ws)) / ws)
    det = a * a + b * b
    out = np.empty_like(p)
    out[:, 0] = (a * p[:, 0] + b * p[:, 1]) / det
    out[:, 1] = (-b * p[:, 0] + a * p[:, 1]) / det
    out[:, 2] = w
    del half
    return out


def se2_compose(a, b):
    """a * b, batched (either operand may be a single pose [3])."""
    a = np.atleast_2d(np.asarray(a, dtype=np.float64))
    b = np.atleast_2d(np.asarray(b, dtype=np.float64))
    c, s = np.cos(a[:, 2]), np.sin(a[:, 2])
    out = np.empty((max(a.shape[0], b.shape[0]), 3))
    out[:, 0] = a[:, 0] + c * b[:, 0] - s * b[:, 1]
    out[:, 1] = a[:, 1] + s * b[:, 0] + c * b[:, 1]
    out[:, 2] = wrap_pi(a[:, 2] + b[:, 2])
    return out


def se2_inverse(a):
    a = np.atleast_2d(np.asarray(a, dtype=np.float64))
    c, s = np.cos(a[:, 2]), np.sin(a[:, 2])
    out = np.empty_like(a)
    out[:, 0] = -(c * a[:, 0] + s * a[:, 1])
    out[:, 1] = -(-s * a[:, 0] + c * a[:, 1])
    out[:, 2] = wrap_pi(-a[:, 2])
    return out


class SE2Pose(object):
    """Single planar pose; theta is kept in [-pi, pi)."""
    dim = 3

    def __init__(self, x: float = 0.0, y: float = 0.0, theta: float = 0.0):
        self._x, self._y, self._theta = float(x), float(y), float(wrap_pi(theta))

    @classmethod
    def by_array(cls, arr):
        return cls(arr[0], arr[1], arr[2])

    @classmethod
    def by_exp_map(cls, vector=None):
        if vector is None:
            return cls()
        return cls(*se2_exp(np.asarray(vector, dtype=np.float64).reshape(1, 3))[0])

    @property
    def x(self):
        return self._x

    @property
    def y(self):
        return self._y

    @property
    def theta(self):
        return self._theta

    @property
    def array(self):
        return np.array([self._x, self._y, self._theta])

    @property
    def matrix(self):
        c, s = math.cos(self._theta), math.sin(self._theta)
        return np.array([[c, -s, self._x], [s, c, self._y], [0.0, 0.0, 1.0]])

    def inverse(self):
        return SE2Pose(*se2_inverse(self.array)[0])

    def log_map(self):
        return se2_log(self.array.reshape(1, 3))[0]

    def __mul__(self, other):
        if isinstance(other, SE2Pose):
            return SE2Pose(*se2_compose(self.array, other.array)[0])
        raise TypeError("SE2Pose can only be composed with SE2Pose")

    def __truediv__(self, other):
        return self * other.inverse()

    def __str__(self):
        return "SE2Pose{x: %s, y: %s, theta: %s}" % (self._x, self._y, self._theta)
