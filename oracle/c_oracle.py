"""ORACLE (test infrastructure, NOT product code) — ctypes front end of the C restatement
(oracle/nsf_oracle.c, built by oracle/Makefile into oracle/_build/libnsf_oracle.so).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
All functions take/return numpy arrays in the torch blob layout documented in nsf_oracle_impl.h.
`dtype` selects the float (reference arithmetic) or double (yardstick) instantiation.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libnsf_oracle.so")
_lib = None


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("nsf_oracle.c", "nsf_oracle_impl.h")]
    if force or not os.path.exists(_LIB_PATH) or any(
            os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in src if os.path.exists(s)):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.nsf_oracle_param_count_f.restype = C.c_size_t
        _lib.nsf_oracle_param_count_d.restype = C.c_size_t
    return _lib


def _sfx(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float32:
        return "_f", C.c_float
    if dtype == np.float64:
        return "_d", C.c_double
    raise ValueError(dtype)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def param_count(D, K, H):
    return int(lib().nsf_oracle_param_count_f(int(D), int(K), int(H)))


def forward(x, blob, K, H, B, L=1, dtype=np.float32):
    s, ct = _sfx(dtype)
    x = np.ascontiguousarray(x, dtype=dtype); blob = np.ascontiguousarray(blob, dtype=dtype)
    n, D = x.shape
    z = np.empty_like(x); ld = np.empty(n, dtype=dtype)
    rc = getattr(lib(), "nsf_oracle_forward" + s)(_p(x), _p(blob), n, D, K, H, ct(B), L, _p(z), _p(ld))
    assert rc == 0, rc
    return z, ld


def nll_grad(x, blob, K, H, B, L=1, dtype=np.float32, want_gx=False):
    """-> (loss, grad[P*L], logprob[n], gx[n,D] or None) for loss = -mean log p(x)."""
    s, ct = _sfx(dtype)
    x = np.ascontiguousarray(x, dtype=dtype); blob = np.ascontiguousarray(blob, dtype=dtype)
    n, D = x.shape
    grad = np.zeros_like(blob); lp = np.empty(n, dtype=dtype)
    gx = np.empty_like(x) if want_gx else None
    loss = C.c_double(0)
    rc = getattr(lib(), "nsf_oracle_backward" + s)(_p(x), _p(blob), n, D, K, H, ct(B), L, None, None, 1,
                                                  _p(grad), _p(gx), C.byref(loss), _p(lp))
    assert rc == 0, rc
    return loss.value, grad, lp, gx


def backward(x, blob, gz, gl, K, H, B, L=1, dtype=np.float32):
    """Vector-Jacobian product: upstream gz[n,D], gl[n] -> (grad_blob, grad_x)."""
    s, ct = _sfx(dtype)
    x = np.ascontiguousarray(x, dtype=dtype); blob = np.ascontiguousarray(blob, dtype=dtype)
    gz = np.ascontiguousarray(gz, dtype=dtype); gl = np.ascontiguousarray(gl, dtype=dtype)
    n, D = x.shape
    grad = np.zeros_like(blob); gx = np.empty_like(x)
    rc = getattr(lib(), "nsf_oracle_backward" + s)(_p(x), _p(blob), n, D, K, H, ct(B), L, _p(gz), _p(gl), 0,
                                                  _p(grad), _p(gx), None, None)
    assert rc == 0, rc
    return grad, gx


def inverse(z, x_sep, blob, K, H, B, L=1, dtype=np.float32):
    s, ct = _sfx(dtype)
    z = np.ascontiguousarray(z, dtype=dtype); blob = np.ascontiguousarray(blob, dtype=dtype)
    n, F = z.shape
    Ds = 0
    if x_sep is not None:
        x_sep = np.ascontiguousarray(x_sep, dtype=dtype); Ds = x_sep.shape[1]
    out = np.empty_like(z); ld = np.empty(n, dtype=dtype)
    rc = getattr(lib(), "nsf_oracle_inverse" + s)(_p(z), _p(x_sep), _p(blob), n, Ds + F, Ds, K, H, ct(B), L,
                                                 _p(out), _p(ld))
    assert rc == 0, rc
    return out, ld


def train(x, blob, K, H, B, L=1, lr=0.015, max_iters=10, average_window=50, loss_delta_tol=1e-2,
          early_stop=True, dtype=np.float32):
    """-> (blob, iter_loss[max_iters], iters_run, adam_m, adam_v)"""
    s, ct = _sfx(dtype)
    x = np.ascontiguousarray(x, dtype=dtype); blob = np.array(blob, dtype=dtype, copy=True)
    n, D = x.shape
    m = np.zeros_like(blob); v = np.zeros_like(blob)
    il = np.zeros(max_iters, dtype=dtype); it = C.c_int(0)
    rc = getattr(lib(), "nsf_oracle_train" + s)(_p(x), _p(blob), _p(m), _p(v), n, D, K, H, ct(B), L, ct(lr),
                                               max_iters, average_window, ct(loss_delta_tol), int(early_stop),
                                               _p(il), C.byref(it))
    assert rc == 0, rc
    return blob, il, it.value, m, v
