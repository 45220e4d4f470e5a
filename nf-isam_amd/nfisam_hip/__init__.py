"""ctypes binding of the C ABI in include/nfisam_hip.h (libnfisam_hip.so, built from
nf-isam_amd/csrc by `make -C nf-isam_amd/csrc` or `__graft_entry__.build()`).

PyTorch is used here only as plumbing: device memory (tensors), streams.  Every compute entry
point goes to the hand-written gfx950 kernels; there is NO CPU or eager-PyTorch fallback — if
the library is missing or the tensors are not on a ROCm device the calls raise.
"""
import ctypes as C
import os
import subprocess
import threading

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libnfisam_hip.so")
CSRC = os.path.join(os.path.dirname(_HERE), "csrc")

OK, ERR_ARG, ERR_LAUNCH, ERR_DOMAIN, ERR_NO_DEVICE, ERR_STALL = 0, 1, 2, 3, 4, 5

EXPORTS = [
    "nfisam_abi_version", "nfisam_last_hip_error", "nfisam_nsf_supported", "nfisam_nsf_param_count",
    "nfisam_nsf_kparam_count", "nfisam_nsf_layout_map", "nfisam_nsf_forward", "nfisam_nsf_inverse",
    "nfisam_nsf_backward", "nfisam_nsf_train_step", "nfisam_nsf_train_loop", "nfisam_nsf_train_plan_create",
    "nfisam_nsf_train_plan_run", "nfisam_nsf_train_plan_destroy", "nfisam_rqs", "nfisam_nsf_posterior_walk", "nfisam_nsf_grad_workspace_count", "nfisam_nsf_train_gradient", "nfisam_nsf_train_chains", "nfisam_nsf_train_gradient_part",
    "nfisam_nsf_train_plan_begin", "nfisam_nsf_train_plan_enqueue", "nfisam_nsf_train_plan_peek", "nfisam_nsf_train_plan_stream",
    "nfisam_nsf_train_plan_end", "nfisam_nsf_train_plan_xcd_span", "nfisam_nsf_train_plan_kernel_ms", "nfisam_nsf_train_plan_create_validated", "nfisam_nsf_train_plan_feed", "nfisam_nsf_train_plan_enqueued", "nfisam_nsf_train_plan_refill",
    "nfisam_normalize_columns", "nfisam_simulate_clique", "nfisam_nsf_train_plan_launch_async",
]


class HipLibraryMissing(ImportError):
    pass


class TrainState(C.Structure):
    _fields_ = [("step", C.c_int32), ("stop", C.c_int32), ("have_avg", C.c_int32), ("loss_avg", C.c_float),
                ("domain_err", C.c_int32), ("reserved", C.c_int32 * 3)]


class AdamCfg(C.Structure):
    _fields_ = [("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
                ("max_iters", C.c_int32), ("average_window", C.c_int32), ("loss_delta_tol", C.c_float),
                ("reserved", C.c_int32)]


class Clique(C.Structure):
    _fields_ = [("x", C.c_void_p), ("kparams", C.c_void_p), ("adam_m", C.c_void_p), ("adam_v", C.c_void_p),
                ("kgrad", C.c_void_p), ("iter_loss", C.c_void_p), ("state", C.c_void_p),
                ("n", C.c_int32), ("D", C.c_int32)]


class Validation(C.Structure):
    _fields_ = [("x_val", C.c_void_p), ("logprob", C.c_void_p), ("val_loss", C.c_void_p), ("n_val", C.c_int32),
                ("reserved", C.c_int32)]


class PostClique(C.Structure):
    _fields_ = [("kparams", C.c_void_p), ("mean", C.c_void_p), ("std", C.c_void_p), ("circular", C.c_void_p),
                ("D_model", C.c_int32), ("n_obs", C.c_int32), ("n_sep", C.c_int32), ("n_frontal", C.c_int32),
                ("obs_off", C.c_int32), ("sep_off", C.c_int32), ("front_off", C.c_int32), ("reserved", C.c_int32)]


assert C.sizeof(TrainState) == 32 and C.sizeof(AdamCfg) == 32 and C.sizeof(Clique) == 64 and C.sizeof(PostClique) == 64

_lib = None


def build(force=False):
    """Compile the shared library for gfx950 with hipcc (works without a GPU)."""
    if force or not os.path.exists(LIB_PATH):
        subprocess.check_call(["make", "-C", CSRC, "-s"] + (["-B"] if force else []))
    else:
        subprocess.check_call(["make", "-C", CSRC, "-s"])
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipLibraryMissing(
                "libnfisam_hip.so not found at %s — build it with `make -C %s` (needs hipcc). "
                "There is no CPU fallback for the NF-iSAM flow hot path." % (LIB_PATH, CSRC))
        _lib = C.CDLL(LIB_PATH)
        _lib.nfisam_nsf_param_count.restype = C.c_size_t
        _lib.nfisam_nsf_kparam_count.restype = C.c_size_t
        _lib.nfisam_nsf_grad_workspace_count.restype = C.c_size_t
        _lib.nfisam_nsf_train_plan_stream.restype = C.c_void_p
        _lib.nfisam_nsf_train_plan_enqueued.restype = C.c_long
        for name in EXPORTS:
            getattr(_lib, name)   # raises AttributeError if the ABI is incomplete
    return _lib


class DomainError(RuntimeError):
    """NFISAM_ERR_DOMAIN: a kernel saw a non-finite loss (the reference raises / asserts at src/flows/utils.py:74-76,133)."""


class PersistentStall(RuntimeError):
    """NFISAM_ERR_STALL: a chunk-persistent training launch gave up waiting for one of its own blocks (somebody else held
    its place on the device).  NOT a numerical failure: the fit is incomplete; re-run it from its initial state -- the
    library keeps to one launch per iteration for the rest of the process."""


def _check(rc, what):
    if rc == OK:
        return
    msg = {ERR_ARG: "invalid argument / unsupported (K,H)", ERR_LAUNCH: "HIP launch failure (hip error %d)" %
           lib().nfisam_last_hip_error(), ERR_DOMAIN: "numerical domain error", ERR_NO_DEVICE: "no gfx950 device",
           ERR_STALL: "a chunk-persistent training launch stalled (a block never became resident)"}
    if rc == ERR_ARG:
        raise ValueError("%s: %s" % (what, msg[rc]))
    if rc == ERR_DOMAIN:
        raise DomainError("%s: %s" % (what, msg[rc]))
    if rc == ERR_STALL:
        raise PersistentStall("%s: %s" % (what, msg[rc]))
    raise RuntimeError("%s: %s" % (what, msg.get(rc, "error %d" % rc)))


def supported(K, H):
    return bool(lib().nfisam_nsf_supported(int(K), int(H)))


def param_count(D, K, H):
    return int(lib().nfisam_nsf_param_count(int(D), int(K), int(H)))


def kparam_count(D, K, H):
    return int(lib().nfisam_nsf_kparam_count(int(D), int(K), int(H)))


_map_cache = {}


def layout_map(D, K, H):
    """np.int32[kparam_count]: index into the reference-order blob, -1 for padding."""
    key = (int(D), int(K), int(H))
    if key not in _map_cache:
        m = np.empty(kparam_count(*key), dtype=np.int32)
        _check(lib().nfisam_nsf_layout_map(key[0], key[1], key[2], m.ctypes.data_as(C.c_void_p)), "layout_map")
        _map_cache[key] = m
    return _map_cache[key]


_tmap_cache = {}


def _torch_maps(D, K, H, device):
    key = (int(D), int(K), int(H), str(device))
    if key not in _tmap_cache:
        m = layout_map(D, K, H)
        valid = torch.from_numpy((m >= 0))
        src = torch.from_numpy(np.where(m >= 0, m, 0).astype(np.int64))
        kidx = torch.from_numpy(np.nonzero(m >= 0)[0].astype(np.int64))
        tidx = torch.from_numpy(m[m >= 0].astype(np.int64))
        _tmap_cache[key] = (valid.to(device).to(torch.float32), src.to(device), kidx.to(device), tidx.to(device))
    return _tmap_cache[key]


class _Staging:
    """Pinned staging ring for the many SMALL host -> device copies of the pipeline (clique tables, observation rows,
    normalisation constants).  A copy from pageable memory blocks the host until the stream has drained in front of it
    (measured: 30-200 us each, 1.6 s of a 17 s eight-replica Plaza1 run); from the ring it is asynchronous.  One ring per
    (thread, stream): a chunk is reused only after the event recorded behind its last copy has fired."""
    CHUNK, CHUNKS = 1 << 16, 8

    def __init__(self):
        self.buf = torch.empty(self.CHUNK * self.CHUNKS, dtype=torch.uint8).pin_memory()
        self.host = self.buf.numpy()
        self.events = [None] * self.CHUNKS
        self.cur, self.off = 0, 0

    def put(self, raw: np.ndarray, device) -> "torch.Tensor":
        n = raw.size
        if self.off + n > self.CHUNK:
            ev = self.events[self.cur] or torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self.events[self.cur] = ev
            self.cur, self.off = (self.cur + 1) % self.CHUNKS, 0
            if self.events[self.cur] is not None:
                self.events[self.cur].synchronize()
        a = self.cur * self.CHUNK + self.off
        self.host[a:a + n] = raw
        self.off += (n + 63) & ~63
        return self.buf[a:a + n].to(device, non_blocking=True)


_staging = threading.local()


def upload(*arrays, device, cached=False):
    """Small numpy arrays -> device tensors of the same dtype and shape, ONE asynchronous copy for all of them on the
    current stream (each array starts 64-byte aligned in the transfer).  The result is ordered behind the copy on THAT
    stream only: a tensor that is kept and later used on other streams (`cached=True`: a model's circular flags, tables that
    outlive the call) is copied the plain way instead -- from pageable memory, host-synchronous, complete when the call
    returns and therefore safe on any stream.  A non-CUDA `device` never goes through the pinned ring (a `.to` of a pinned
    slice onto the CPU would alias the ring)."""
    arrays = [np.ascontiguousarray(a) for a in arrays]
    if not arrays:
        return []
    offs, total = [], 0
    for a in arrays:
        offs.append(total)
        total += (a.nbytes + 63) & ~63
    raw = np.zeros(max(total, 1), dtype=np.uint8)
    for a, o in zip(arrays, offs):
        raw[o:o + a.nbytes] = a.reshape(-1).view(np.uint8)
    if total > _Staging.CHUNK or cached or not torch.cuda.is_available() or torch.device(device).type != "cuda":
        dev = torch.from_numpy(raw).to(device)
    else:
        rings = _staging.__dict__.setdefault("rings", {})
        key = torch.cuda.current_stream().cuda_stream
        ring = rings.get(key)
        if ring is None:
            ring = rings[key] = _Staging()
        dev = ring.put(raw, device)
    out = []
    for a, o in zip(arrays, offs):
        out.append(dev[o:o + a.nbytes].view(_TORCH_OF[a.dtype.type]).reshape(a.shape))
    return out


_TORCH_OF = {np.float32: torch.float32, np.int32: torch.int32, np.uint8: torch.uint8, np.int64: torch.int64,
             np.float64: torch.float64, np.bool_: torch.bool}


def pack(blob, D, K, H, L=1):
    """reference-order parameters [L*P] -> kernel layout [L*Pk] (same device/dtype float32)."""
    P, Pk = param_count(D, K, H), kparam_count(D, K, H)
    blob = blob.reshape(L, P)
    valid, src, _, _ = _torch_maps(D, K, H, blob.device)
    out = blob[:, src] * (valid if blob.dtype == torch.float32 else valid.to(blob.dtype))
    return out.reshape(L * Pk).contiguous()


def unpack(kblob, D, K, H, L=1):
    """kernel layout [L*Pk] -> reference-order parameters [L*P]."""
    P, Pk = param_count(D, K, H), kparam_count(D, K, H)
    kblob = kblob.reshape(L, Pk)
    _, _, kidx, tidx = _torch_maps(D, K, H, kblob.device)
    out = torch.empty(L, P, dtype=kblob.dtype, device=kblob.device)
    out[:, tidx] = kblob[:, kidx]
    return out.reshape(L * P)


def _dev(t, name, dtype=torch.float32):
    if t is None:
        return None
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError("%s must be a tensor on a ROCm device (no CPU path exists)" % name)
    if t.dtype != dtype:
        raise ValueError("%s must be %s" % (name, dtype))
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous" % name)
    return t


def _ptr(t):
    return C.c_void_p(0 if t is None else t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """Raw handle of torch's current stream (what the C ABI takes).  `torch.cuda.current_stream()` builds a Stream object per
    call (2.7 us); the raw-handle lookup is 0.4 us (scripts/exp/stream_call_cost.py) -- it sits inside every launch of this module."""
    if _raw_stream is not None:
        return C.c_void_p(_raw_stream(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _stride(kparams, D, K, H, L, model_D):
    """size_t layer stride for a (possibly truncated) evaluation; validates the blob size."""
    md = D if model_D is None else int(model_D)
    if md < D:
        raise ValueError("model_D=%d is smaller than the evaluated dimension %d" % (md, D))
    if kparams.numel() != L * kparam_count(md, K, H):
        raise ValueError("kparams has %d elements, expected %d" % (kparams.numel(), L * kparam_count(md, K, H)))
    return C.c_size_t(0 if md == D else kparam_count(md, K, H))


def forward(x, kparams, K, H, B, L=1, want_z=True, want_logdet=True, want_logprob=False, model_D=None):
    """x[n,D] -> (z, logdet, logprob) (None for the ones not requested)."""
    _dev(x, "x"); _dev(kparams, "kparams")
    n, D = x.shape
    stride = _stride(kparams, D, K, H, L, model_D)
    z = torch.empty_like(x) if want_z else None
    ld = torch.empty(n, dtype=torch.float32, device=x.device) if want_logdet else None
    lp = torch.empty(n, dtype=torch.float32, device=x.device) if want_logprob else None
    _check(lib().nfisam_nsf_forward(_ptr(x), _ptr(kparams), n, D, int(K), int(H), C.c_float(B), int(L), stride, _ptr(z),
                                    _ptr(ld), _ptr(lp), _stream()), "nfisam_nsf_forward")
    return z, ld, lp


def inverse(z, x_sep, kparams, K, H, B, L=1, mean=None, std=None, circular=None, want_logdet=False,
            model_D=None):
    """z[n,D-Ds], x_sep[n,Ds] raw (or None) -> x_free[n,D-Ds] (and logdet[n])."""
    _dev(z, "z"); _dev(kparams, "kparams"); _dev(x_sep, "x_sep"); _dev(mean, "mean"); _dev(std, "std")
    _dev(circular, "circular", torch.uint8)
    n, F = z.shape
    Ds = 0 if x_sep is None else x_sep.shape[1]
    D = Ds + F
    if x_sep is not None and x_sep.shape[0] != n:
        raise ValueError("x_sep and z disagree on the number of particles")
    stride = _stride(kparams, D, K, H, L, model_D)
    for t, nm in ((mean, "mean"), (std, "std"), (circular, "circular")):
        if t is not None and t.numel() < D:
            raise ValueError("%s must have at least D=%d entries" % (nm, D))
    out = torch.empty_like(z)
    ld = torch.empty(n, dtype=torch.float32, device=z.device) if want_logdet else None
    _check(lib().nfisam_nsf_inverse(_ptr(z), _ptr(x_sep), _ptr(kparams), n, D, Ds, int(K), int(H), C.c_float(B),
                                    int(L), stride, _ptr(mean), _ptr(std), _ptr(circular), _ptr(out), _ptr(ld), _stream()),
           "nfisam_nsf_inverse")
    return (out, ld) if want_logdet else out


def _rqs_call(inputs, w, h, d, inverse, left, right, bottom, top, padded):
    shape = inputs.shape
    inp = _dev(inputs.reshape(-1).contiguous().float(), "inputs")
    K = w.shape[-1]
    M = inp.numel()
    dc = K - 1 if padded else K + 1
    w = _dev(w.reshape(M, K).contiguous().float(), "unnormalized_widths")
    h = _dev(h.reshape(M, K).contiguous().float(), "unnormalized_heights")
    d = _dev(d.reshape(M, dc).contiguous().float(), "unnormalized_derivatives")
    if K * 1e-3 > 1.0:
        raise ValueError("Minimal bin width too large for the number of bins")
    out = torch.empty_like(inp); lad = torch.empty_like(inp)
    _check(lib().nfisam_rqs(_ptr(inp), _ptr(w), _ptr(h), _ptr(d), M, K, int(inverse), C.c_float(left),
                            C.c_float(right), C.c_float(bottom), C.c_float(top), int(padded), _ptr(out), _ptr(lad),
                            _stream()), "nfisam_rqs")
    return out.reshape(shape), lad.reshape(shape)


def rqs(inputs, widths, heights, derivs, inverse, tail_bound):
    """unconstrained_RQS: derivs has K-1 columns, linear tails outside [-tail_bound, tail_bound]."""
    return _rqs_call(inputs, widths, heights, derivs, inverse, -tail_bound, tail_bound, -tail_bound, tail_bound, True)


def rqs_box(inputs, widths, heights, derivs, inverse, left, right, bottom, top):
    """bounded RQS: derivs has K+1 columns."""
    return _rqs_call(inputs, widths, heights, derivs, inverse, left, right, bottom, top, False)


def backward(x, kparams, K, H, B, L=1, gz=None, gl=None, nll_mode=False, want_gx=False, model_D=None):
    """VJP of the flow.  -> (kgrad[L*Pk], gx[n,D] or None, loss_sum tensor[1] or None)."""
    _dev(x, "x"); _dev(kparams, "kparams"); _dev(gz, "gz"); _dev(gl, "gl")
    n, D = x.shape
    stride = _stride(kparams, D, K, H, L, model_D)
    kgrad = torch.zeros_like(kparams)
    gx = torch.empty_like(x) if want_gx else None
    loss = torch.zeros(1, dtype=torch.float32, device=x.device) if nll_mode else None
    _check(lib().nfisam_nsf_backward(_ptr(x), _ptr(kparams), n, D, int(K), int(H), C.c_float(B), int(L), stride, _ptr(gz),
                                     _ptr(gl), int(bool(nll_mode)), _ptr(kgrad), _ptr(gx), _ptr(loss), _stream()),
           "nfisam_nsf_backward")
    return kgrad, gx, loss


class TrainBatch:
    """Device-resident training state of a batch of independent cliques (one flow each).

    Mirrors what NFiSAM.fit_clique_density_model keeps per clique (model parameters, Adam
    moments, per-iteration loss vector; src/slam/NFiSAM.py:418-447) but for many cliques at
    once, so that one launch covers grid.y = clique."""

    def __init__(self, xs, kparams, K, H, B, L, lr, max_iters, average_window=50, loss_delta_tol=1e-2,
                 beta1=0.9, beta2=0.999, eps=1e-8, early_stop=True, x_val=None, validation_interval=10, slower_stop_rate=2.0):
        """x_val (one held-out batch per clique): the plan stops by the reference's hold-out rule (NFiSAM.py:452-468,
        nfisam_nsf_train_plan_create_validated) instead of the window rule; `val_loss[c]` then records every evaluation."""
        if len(xs) != len(kparams) or len(xs) == 0:
            raise ValueError("need one parameter blob per clique batch")
        self.x_val = None
        if x_val is not None:
            if len(x_val) != len(xs) or any(v.shape[1] != x.shape[1] or v.shape[0] < 1 for v, x in zip(x_val, xs)):
                raise ValueError("need one held-out batch [n_val, D] per clique")
            self.x_val = [_dev(v, "x_val") for v in x_val]
            self.validation_interval, self.slower_stop_rate = int(validation_interval), float(slower_stop_rate)
            early_stop = False
        self.K, self.H, self.B, self.L = int(K), int(H), float(B), int(L)
        self.device = xs[0].device
        self.xs = [_dev(x, "x") for x in xs]
        self.kparams = [_dev(p, "kparams") for p in kparams]
        for x, p in zip(self.xs, self.kparams):
            if p.numel() != L * kparam_count(x.shape[1], K, H):
                raise ValueError("kparams size does not match (D,K,H,L)")
        self.m = [torch.zeros_like(p) for p in self.kparams]
        self.v = [torch.zeros_like(p) for p in self.kparams]
        self.max_n = max(x.shape[0] for x in xs)
        self.g = [torch.zeros(int(lib().nfisam_nsf_grad_workspace_count(self.max_n, x.shape[1], int(K), int(H), int(L))),
                              dtype=torch.float32, device=self.device) for x in self.xs]
        self.iter_loss = [torch.zeros(max(int(max_iters), 1), dtype=torch.float32, device=self.device) for _ in xs]
        self.states = torch.zeros(len(xs), C.sizeof(TrainState) // 4, dtype=torch.int32, device=self.device)
        self.cfg = AdamCfg(lr, beta1, beta2, eps, int(max_iters), int(average_window) if early_stop else 0,
                           loss_delta_tol, 0)
        self.nc = len(xs)
        self.max_n = max(x.shape[0] for x in xs)
        self.max_D = max(x.shape[1] for x in xs)
        self.host_desc = (Clique * self.nc)()
        for c in range(self.nc):
            d = self.host_desc[c]
            d.x = self.xs[c].data_ptr(); d.kparams = self.kparams[c].data_ptr()
            d.adam_m = self.m[c].data_ptr(); d.adam_v = self.v[c].data_ptr(); d.kgrad = self.g[c].data_ptr()
            d.iter_loss = self.iter_loss[c].data_ptr()
            d.state = self.states.data_ptr() + C.sizeof(TrainState) * c
            d.n, d.D = self.xs[c].shape
        raw = np.frombuffer(bytes(self.host_desc), dtype=np.uint8).copy()
        self.dev_desc = torch.from_numpy(raw).to(self.device)
        if self.x_val is not None:
            n_eval = max(1, int(max_iters) // max(self.validation_interval, 1))
            self.val_scratch = [torch.empty(v.shape[0], dtype=torch.float32, device=self.device) for v in self.x_val]
            self.val_loss = [torch.zeros(n_eval, dtype=torch.float32, device=self.device) for _ in self.x_val]
            self.val_desc = (Validation * self.nc)()
            for c in range(self.nc):
                q = self.val_desc[c]
                q.x_val, q.logprob, q.val_loss = self.x_val[c].data_ptr(), self.val_scratch[c].data_ptr(), self.val_loss[c].data_ptr()
                q.n_val = self.x_val[c].shape[0]

    def step(self):
        """Enqueue ONE training iteration for all cliques (no host sync)."""
        if self.nc == 1:
            rc = lib().nfisam_nsf_train_step(C.byref(self.host_desc[0]), 1, 1, self.max_n, self.max_D, self.K, self.H,
                                             C.c_float(self.B), self.L, C.byref(self.cfg), _stream())
        else:
            rc = lib().nfisam_nsf_train_step(C.c_void_p(self.dev_desc.data_ptr()), self.nc, 0, self.max_n, self.max_D,
                                             self.K, self.H, C.c_float(self.B), self.L, C.byref(self.cfg), _stream())
        _check(rc, "nfisam_nsf_train_step")

    def chains(self):
        """Launches a training plan issues per iteration for this batch (parallel graph branches; 1 = not split)."""
        return int(lib().nfisam_nsf_train_chains(self.nc, self.max_n, self.max_D, self.K, self.H, self.L))

    def gradient_part(self, chain, n_chains, stream=None):
        """Launch `chain` of `n_chains` of the gradient half of an iteration, on `stream` (default: the current one)."""
        st = C.c_void_p(stream.cuda_stream) if stream is not None else _stream()
        if self.nc == 1:
            rc = lib().nfisam_nsf_train_gradient_part(C.byref(self.host_desc[0]), 1, 1, self.max_n, self.max_D, self.K, self.H,
                                                      C.c_float(self.B), self.L, int(chain), int(n_chains), st)
        else:
            rc = lib().nfisam_nsf_train_gradient_part(C.c_void_p(self.dev_desc.data_ptr()), self.nc, 0, self.max_n, self.max_D,
                                                      self.K, self.H, C.c_float(self.B), self.L, int(chain), int(n_chains), st)
        _check(rc, "nfisam_nsf_train_gradient_part")

    def gradient_only(self):
        """Enqueue only the gradient kernel of an iteration (no Adam, no bookkeeping)."""
        if self.nc == 1:
            rc = lib().nfisam_nsf_train_gradient(C.byref(self.host_desc[0]), 1, 1, self.max_n, self.max_D, self.K,
                                                 self.H, C.c_float(self.B), self.L, _stream())
        else:
            rc = lib().nfisam_nsf_train_gradient(C.c_void_p(self.dev_desc.data_ptr()), self.nc, 0, self.max_n,
                                                 self.max_D, self.K, self.H, C.c_float(self.B), self.L, _stream())
        _check(rc, "nfisam_nsf_train_gradient")

    def prepare(self, use_graph=True, timing=False, span=False):
        """Validate descriptors and (optionally) capture + instantiate the hipGraph of one chunk of
        iterations.  One-time set-up; `run` calls it on first use.  `timing`: the graph carries two timing events around
        the chunk's training launches (`kernel_ms`; measurement only).  `span`: the plan also gets the window-spanning graph
        that `launch_async` needs (single-clique plans)."""
        if getattr(self, "_plan", None) is not None and self._plan_graph == bool(use_graph) and (not timing or getattr(self, "_plan_timing", False)) \
                and (not span or getattr(self, "_plan_span", False)):
            return
        self._plan_timing = bool(timing) and bool(use_graph)
        self._plan_span = (bool(span) or getattr(self, "_plan_span", False)) and bool(use_graph) and not self._plan_timing
        self.close()
        plan = C.c_void_p(0)
        dev_desc = C.c_void_p(self.dev_desc.data_ptr()) if self.nc > 1 else None
        if self.x_val is not None:
            rc = lib().nfisam_nsf_train_plan_create_validated(self.host_desc, dev_desc, self.nc, self.K, self.H, C.c_float(self.B), self.L,
                                                              C.byref(self.cfg), self.val_desc, self.validation_interval,
                                                              C.c_float(self.slower_stop_rate), int(bool(use_graph)) | (2 if self._plan_timing else 0) | (4 if self._plan_span else 0), C.byref(plan))
        else:
            rc = lib().nfisam_nsf_train_plan_create(self.host_desc, dev_desc, self.nc, self.K, self.H, C.c_float(self.B), self.L,
                                                    C.byref(self.cfg), int(bool(use_graph)) | (2 if self._plan_timing else 0) | (4 if self._plan_span else 0), C.byref(plan))
        _check(rc, "nfisam_nsf_train_plan_create")
        self._plan, self._plan_graph = plan, bool(use_graph)

    def run(self, use_graph=True):
        """Run until every clique stopped early or reached max_iters (the reference's
        `for i in range(flow_iterations)` loop).  Host-synchronising once per chunk.
        -> list of iterations run per clique."""
        self.prepare(use_graph)
        iters = (C.c_int32 * self.nc)()
        rc = lib().nfisam_nsf_train_plan_run(self._plan, iters, _stream())
        self.last_iters = [int(v) for v in iters]          # valid also when a clique hit a domain error
        _check(rc, "nfisam_nsf_train_plan_run")
        return self.last_iters

    def launch_async(self):
        """Enqueue the WHOLE run of a single-clique plan on the current stream as one window-spanning launch that evaluates the
        early-stop rule itself, and return at once (nfisam_nsf_train_plan_launch_async).  -> True: enqueued -- the outcome is in
        `states[0]` (step = iterations run, stop, domain_err) / `iter_loss[0]` / `kparams[0]` when the stream has drained;
        False: this plan or this moment does not allow it (call `run`)."""
        if self.nc != 1 or self.x_val is not None:
            return False
        self.prepare(True, span=True)
        rc = lib().nfisam_nsf_train_plan_launch_async(self._plan, _stream())
        if rc == ERR_ARG:
            return False
        _check(rc, "nfisam_nsf_train_plan_launch_async")
        return True

    def kernel_ms(self):
        """GPU milliseconds of the training launches of the most recent chunk replay (plans prepared with `timing=True`;
        synchronise first)."""
        ms = C.c_float(0.0)
        _check(lib().nfisam_nsf_train_plan_kernel_ms(self._plan, C.byref(ms)), "nfisam_nsf_train_plan_kernel_ms")
        return float(ms.value)

    def xcd_span(self):
        """Most XCDs one (clique, dim) group of the plan's chunk-persistent launches ran on (0: none ran; diagnostic)."""
        return int(lib().nfisam_nsf_train_plan_xcd_span(self._plan)) if getattr(self, "_plan", None) is not None else 0

    # ---- stepping the plan by hand (nfisam_nsf_train_plan_begin / enqueue / peek / stream / end) ----------------------
    def begin(self):
        """Start a hand-stepped run of the (graph) plan: chunks are enqueued with `enqueue`, looked at with `peek`."""
        self.prepare(True)
        _check(lib().nfisam_nsf_train_plan_begin(self._plan, _stream()), "nfisam_nsf_train_plan_begin")
        if getattr(self, "_plan_stream", None) is None:
            self._plan_stream = torch.cuda.ExternalStream(int(lib().nfisam_nsf_train_plan_stream(self._plan)), device=self.device)
            self._peek = (TrainState * self.nc)()

    def enqueue(self):
        """Append one chunk of iterations to the plan's stream (non-blocking)."""
        _check(lib().nfisam_nsf_train_plan_enqueue(self._plan), "nfisam_nsf_train_plan_enqueue")

    def peek(self):
        """-> (chunks closed since `begin`, [(step, stop, domain_err) per clique]) as of the last closed chunk; chunks = -1
        when a chunk closed while the mirror was being copied (look again)."""
        _check(lib().nfisam_nsf_train_plan_peek(self._plan, self._peek), "nfisam_nsf_train_plan_peek")
        seq = min(int(s.reserved[0]) for s in self._peek)
        return seq, [(int(s.step), int(s.stop), int(s.domain_err)) for s in self._peek]

    def feed(self, depth):
        """The library's feeder thread keeps `depth` chunks enqueued ahead of the last closed one (0: pause)."""
        _check(lib().nfisam_nsf_train_plan_feed(self._plan, int(depth)), "nfisam_nsf_train_plan_feed")

    def enqueued(self):
        return int(lib().nfisam_nsf_train_plan_enqueued(self._plan))

    def refill(self, c, x, kparams):
        """Slot c gets a NEW problem of the same shape: batch, fresh parameters, zeroed moments / workspace / loss record and
        -- last -- state, enqueued on the plan's stream behind the chunks already there (the slot's old clique must have
        stopped: nothing writes its buffers any more).  `x` / `kparams` (float32, contiguous, produced on the current stream)
        must stay alive until the copies have run -- keep a reference as long as the slot trains them."""
        if x.shape != self.xs[c].shape or kparams.numel() != self.kparams[c].numel() or x.dtype != torch.float32 or \
                kparams.dtype != torch.float32 or not x.is_contiguous() or not kparams.is_contiguous() or x.device != self.device:
            raise ValueError("refill needs a contiguous float32 batch and parameters of the slot's shape on the plan's device")
        _check(lib().nfisam_nsf_train_plan_refill(self._plan, int(c), C.c_void_p(x.data_ptr()), C.c_void_p(kparams.data_ptr()),
                                                  _stream()), "nfisam_nsf_train_plan_refill")

    def end(self):
        """The current stream continues behind everything enqueued on the plan."""
        _check(lib().nfisam_nsf_train_plan_end(self._plan, _stream()), "nfisam_nsf_train_plan_end")

    def reset(self, kparams=None):
        """Re-initialise Adam moments / state / loss record in place (pointers stay valid, so a
        prepared plan can be re-run)."""
        for c in range(self.nc):
            if kparams is not None:
                self.kparams[c].copy_(kparams[c])
            self.m[c].zero_(); self.v[c].zero_(); self.g[c].zero_(); self.iter_loss[c].zero_()
            if self.x_val is not None:
                self.val_loss[c].zero_()
        self.states.zero_()

    def close(self):
        if getattr(self, "_plan", None) is not None:
            lib().nfisam_nsf_train_plan_destroy(self._plan)
            self._plan = None
            self._plan_stream = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def state(self, c=0):
        s = self.states[c].cpu().numpy()
        return {"step": int(s[0]), "stop": int(s[1]), "have_avg": int(s[2]),
                "loss_avg": float(s[3:4].view(np.float32)[0]), "domain_err": int(s[4]), "slower_stop_iter": int(s[7])}


def posterior_walk(entries, total_dim, n, K, H, B, L, device, generator=None, Zt=None):
    """Sample a whole Bayes tree root -> leaves in one launch.

    entries: list (parents before children) of dicts with keys
        kparams, mean, std, circular (device tensors), D_model, obs (1-D numpy), sep_cols, front_cols (lists of
        column indices into the [n, total_dim] sample matrix).
    -> device tensor [n, total_dim] (float32) of posterior samples."""
    nc = len(entries)
    table = (PostClique * nc)()
    cols, obs = [], []
    max_D = 1
    for e, q in zip(entries, table):
        q.kparams = e["kparams"].data_ptr(); q.mean = e["mean"].data_ptr(); q.std = e["std"].data_ptr()
        q.circular = e["circular"].data_ptr()
        q.D_model = int(e["D_model"]); q.n_obs = len(e["obs"]); q.n_sep = len(e["sep_cols"])
        q.n_frontal = len(e["front_cols"])
        if q.n_obs + q.n_sep + q.n_frontal > q.D_model:
            raise ValueError("clique columns exceed its model dimension")
        q.obs_off = len(obs); obs.extend(float(v) for v in e["obs"])
        q.sep_off = len(cols); cols.extend(int(v) for v in e["sep_cols"])
        q.front_off = len(cols); cols.extend(int(v) for v in e["front_cols"])
        max_D = max(max_D, q.D_model)
    if sum(int(q.n_frontal) for q in table) > int(total_dim) or (cols and not 0 <= min(cols) <= max(cols) < int(total_dim)):
        raise ValueError("the cliques' frontal columns exceed the %d columns of the sample matrix (one latent row of Zt per "
                         "frontal column, consumed in walk order), or a column index is out of range" % total_dim)
    tbl = torch.from_numpy(np.frombuffer(bytes(table), dtype=np.uint8).copy()).to(device)
    cols_t = torch.tensor(cols if cols else [0], dtype=torch.int32, device=device)
    obs_t = torch.tensor(obs if obs else [0.0], dtype=torch.float32, device=device)
    if Zt is None:
        Zt = torch.randn(total_dim, n, dtype=torch.float32, device=device, generator=generator)
    Zt = _dev(Zt, "Zt")
    if tuple(Zt.shape) != (total_dim, n):
        raise ValueError("Zt must be [total_dim, n] (column-major sample layout)")
    St = torch.zeros(total_dim, n, dtype=torch.float32, device=device)
    _check(lib().nfisam_nsf_posterior_walk(C.c_void_p(tbl.data_ptr()), nc, _ptr(cols_t), _ptr(obs_t), max_D, int(K),
                                           int(H), C.c_float(B), int(L), int(n), _ptr(Zt), _ptr(St), _stream()),
           "nfisam_nsf_posterior_walk")
    return St.t().contiguous()


class SimOp(C.Structure):
    """`nfisam_sim_op` (include/nfisam_hip.h)."""
    _fields_ = [("code", C.c_int32), ("a", C.c_int32), ("b", C.c_int32), ("c", C.c_int32), ("cand", C.c_int32 * 4),
                ("k", C.c_int32), ("p", C.c_float * 9), ("src", C.c_uint64)]


SIM_MAX_OPS = 40
SIM_COPY, SIM_PRIOR_SE2, SIM_REL_FWD, SIM_REL_BWD, SIM_REL_OBS, SIM_RING, SIM_RANGE_OBS, SIM_ADA_OBS, SIM_NH_RING, \
    SIM_NH_OBS, SIM_PRIOR_R2, SIM_PRIOR_R2_RING, SIM_REL_R2_FWD, SIM_REL_R2_BWD, SIM_REL_R2_OBS = range(1, 16)
assert C.sizeof(SimOp) == 80


def simulate_clique(ops, n, D_out, D_total, seed, device):
    """Run a compiled simulation schedule (list of SimOp) -> device tensor [n, D_out] float32."""
    arr = (SimOp * len(ops))(*ops)
    out = torch.empty(int(n), int(D_out), dtype=torch.float32, device=device)
    _check(lib().nfisam_simulate_clique(arr, len(ops), int(n), int(D_out), int(D_total), C.c_uint64(int(seed)), _ptr(out),
                                        _stream()), "nfisam_simulate_clique")
    return out


def normalize_columns(x, circular=None):
    """NFiSAM.normalize_training_samples on the device: x [n, D] float32 (cuda) -> (x_normalised, mean[D], std[D])."""
    x = _dev(x, "x")
    n, D = x.shape
    circ = None
    if circular is not None:
        circ, = upload(np.asarray(circular, dtype=np.uint8), device=x.device)
    out = torch.empty_like(x)
    mean = torch.empty(D, dtype=torch.float32, device=x.device)
    std = torch.empty(D, dtype=torch.float32, device=x.device)
    _check(lib().nfisam_normalize_columns(_ptr(x), int(n), int(D), _ptr(circ) if circ is not None else None, _ptr(out),
                                          _ptr(mean), _ptr(std), _stream()), "nfisam_normalize_columns")
    return out, mean, std


POST_DTYPE = np.dtype([("kparams", np.uint64), ("mean", np.uint64), ("std", np.uint64), ("circular", np.uint64),
                       ("D_model", np.int32), ("n_obs", np.int32), ("n_sep", np.int32), ("n_frontal", np.int32),
                       ("obs_off", np.int32), ("sep_off", np.int32), ("front_off", np.int32), ("reserved", np.int32)])
assert POST_DTYPE.itemsize == C.sizeof(PostClique)


def posterior_walk_raw(table: np.ndarray, cols: np.ndarray, obs: np.ndarray, total_dim, n, max_D, K, H, B, L, device,
                       Zt=None):
    """`posterior_walk` with the clique table already assembled as a numpy array of POST_DTYPE
    (callers that walk large trees every update cache the per-clique pointers)."""
    if int(table["n_frontal"].sum()) > int(total_dim) or (cols.size and not 0 <= int(cols.min()) <= int(cols.max()) < int(total_dim)):
        raise ValueError("the cliques' frontal columns exceed the %d columns of the sample matrix, or a column index is out of range"
                         % total_dim)
    if Zt is not None and tuple(Zt.shape) != (int(total_dim), int(n)):
        raise ValueError("Zt must be [total_dim, n] (column-major sample layout)")
    tbl, cols_t, obs_t = upload(table.view(np.uint8).reshape(-1), np.asarray(cols if cols.size else np.zeros(1), dtype=np.int32),
                                np.asarray(obs if obs.size else np.zeros(1), dtype=np.float32), device=device)
    if Zt is None:
        Zt = torch.randn(total_dim, n, dtype=torch.float32, device=device)
    St = torch.zeros(total_dim, n, dtype=torch.float32, device=device)
    _check(lib().nfisam_nsf_posterior_walk(C.c_void_p(tbl.data_ptr()), int(table.shape[0]), _ptr(cols_t), _ptr(obs_t),
                                           int(max_D), int(K), int(H), C.c_float(B), int(L), int(n), _ptr(Zt), _ptr(St),
                                           _stream()), "nfisam_nsf_posterior_walk")
    return St.t().contiguous()
