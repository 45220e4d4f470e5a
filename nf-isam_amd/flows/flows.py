"""flows.flows — FCNN and NSF_AR with the reference's constructor/forward/inverse signatures
(reference: src/flows/flows.py:26-137), evaluated by the gfx950 kernels behind
include/nfisam_hip.h.

Parameters are ordinary nn.Parameters laid out exactly as in the reference (`init_param`,
`layers.{j}.network.{0,2,4}.{weight,bias}`), so `state_dict()`, `.parameters()`, `.to()` and
`torch.optim.Adam(model.parameters())` (src/slam/NFiSAM.py:425) keep working.  For a kernel call
they are packed once into the kernel layout (cached until a parameter changes).

Deliberate difference: `forward` returns z and log_det in the mathematically correct (n, d)
layout; the reference returns a scrambled one (flows.py:88-93).  Pass
`reference_scramble=True` to reproduce the reference's return value bit-for-layout.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.init as init

import nfisam_hip as _nh
from flows.utils import unconstrained_RQS  # noqa: F401  (same import as the reference module)


class FCNN(nn.Module):
    """Linear-tanh-Linear-tanh-Linear conditioner (reference: flows.py:26-41).  Holds the parameters;
    inside NSF_AR the network is evaluated by the fused kernels, `forward` here is only the
    stand-alone module contract and uses torch ops."""

    def __init__(self, in_dim, out_dim, hidden_dim):
        super().__init__()
        self.network = nn.Sequential(
            nn.Linear(in_dim, hidden_dim),
            nn.Tanh(),
            nn.Linear(hidden_dim, hidden_dim),
            nn.Tanh(),
            nn.Linear(hidden_dim, out_dim),
        )

    def forward(self, x):
        return self.network(x)


class _NSFFunction(torch.autograd.Function):
    """(x, reference-order parameter blob) -> (z, log_det) through the HIP kernels, with the
    analytic VJP kernel as backward (replaces autograd through ~2.2k eager ops, NFiSAM.py:470-474)."""

    @staticmethod
    def forward(ctx, x, blob, D, K, H, B):
        kp = _nh.pack(blob, D, K, H, 1)
        z, ld, _ = _nh.forward(x, kp, K, H, B, 1)
        ctx.save_for_backward(x, kp)
        ctx.cfg = (D, K, H, B, x.requires_grad)
        return z, ld

    @staticmethod
    def backward(ctx, gz, gl):
        x, kp = ctx.saved_tensors
        D, K, H, B, need_gx = ctx.cfg
        gz = torch.zeros_like(x) if gz is None else gz.contiguous()
        gl = torch.zeros(x.shape[0], device=x.device) if gl is None else gl.contiguous()
        kg, gx, _ = _nh.backward(x, kp, K, H, B, 1, gz=gz, gl=gl, want_gx=need_gx)
        return gx, _nh.unpack(kg, D, K, H, 1), None, None, None, None


def _init_bounds(dim, K, H):
    """Per-parameter half-width of the reference initialisation, in the reference's parameter order:
    init_param ~ U(-1/2, 1/2) (flows.py:62-63); nn.Linear default = U(+-1/sqrt(fan_in)) for weight and bias."""
    Po = 3 * K - 1
    parts = [np.full(Po, 0.5, dtype=np.float32)]
    for i in range(1, dim):
        for fan_in, cnt in ((i, H * i), (i, H), (H, H * H), (H, H), (H, Po * H), (H, Po)):
            parts.append(np.full(cnt, 1.0 / np.sqrt(fan_in), dtype=np.float32))
    return np.concatenate(parts)


_bounds_cache = {}


def init_reference_blob(dim, K, H, device, generator=None):
    """Freshly initialised parameters of one NSF_AR layer as ONE flat device tensor (reference order)."""
    key = (dim, K, H, str(device))
    if key not in _bounds_cache:
        _bounds_cache[key] = torch.from_numpy(_init_bounds(dim, K, H)).to(device)
    b = _bounds_cache[key]
    return (torch.rand(b.shape, device=device, generator=generator) * 2.0 - 1.0) * b


class NSF_AR(nn.Module):
    """Neural spline flow, auto-regressive [Durkan et al. 2019] (reference: flows.py:43-137).
    K is the number of spline bins, B the tail bound.

    Two construction modes.  The reference's: `NSF_AR(dim, K, B, hidden_dim)` builds `init_param` and
    the `layers` ModuleList of FCNNs right away.  The solver's: `NSF_AR.from_kernel_params(...)` wraps a
    kernel-layout parameter blob that already lives on the device and builds the nn.Module tree only if
    somebody asks for it (`.layers`, `.init_param`, `.parameters()`, `state_dict()`): constructing ~90
    module objects and moving ~90 tiny tensors per clique costs more host time than training the clique."""

    def __init__(self, dim, K=5, B=5.0, hidden_dim=8, base_network=FCNN, reference_scramble=False):
        super().__init__()
        if base_network is not FCNN:
            raise NotImplementedError("the gfx950 kernels implement the FCNN conditioner only")
        self.dim = dim
        self.K = K
        self.B = B
        self.hidden_dim = hidden_dim
        self.reference_scramble = reference_scramble
        self._kcache = None
        self._lazy_kparams = None
        self._build_modules()
        self.reset_parameters()

    def _build_modules(self):
        self.layers = nn.ModuleList()
        self.init_param = nn.Parameter(torch.Tensor(3 * self.K - 1))
        for i in range(1, self.dim):
            self.layers += [FCNN(i, 3 * self.K - 1, self.hidden_dim)]

    @classmethod
    def from_kernel_params(cls, dim, K, B, hidden_dim, kparams):
        """Wrap a kernel-layout blob [kparam_count(dim, K, H)] (device tensor) without building modules."""
        self = cls.__new__(cls)
        nn.Module.__init__(self)
        self.dim, self.K, self.B, self.hidden_dim = dim, K, B, hidden_dim
        self.reference_scramble = False
        self._kcache = None
        self._lazy_kparams = kparams
        return self

    def _materialize(self):
        """Build the nn.Module tree from the wrapped kernel blob (rare: only for module-level access)."""
        kp = self._lazy_kparams
        self._lazy_kparams = None
        self._build_modules()
        self.to(kp.device)
        self.load_kernel_params(kp)

    def __getattr__(self, name):
        if name in ("layers", "init_param") and self.__dict__.get("_lazy_kparams") is not None:
            self._materialize()
        return super().__getattr__(name)

    def parameters(self, recurse: bool = True):
        if self.__dict__.get("_lazy_kparams") is not None:
            self._materialize()
        return super().parameters(recurse)

    def named_parameters(self, *args, **kwargs):
        if self.__dict__.get("_lazy_kparams") is not None:
            self._materialize()
        return super().named_parameters(*args, **kwargs)

    def state_dict(self, *args, **kwargs):
        if self.__dict__.get("_lazy_kparams") is not None:
            self._materialize()
        return super().state_dict(*args, **kwargs)

    def _apply(self, fn, recurse=True):
        if self.__dict__.get("_lazy_kparams") is not None:
            self._lazy_kparams = fn(self._lazy_kparams)
            return self
        return super()._apply(fn, recurse)

    @property
    def device(self):
        kp = self.__dict__.get("_lazy_kparams")
        return kp.device if kp is not None else self.init_param.device

    def reset_parameters(self):
        init.uniform_(self.init_param, -1 / 2, 1 / 2)

    # ---- parameter plumbing ------------------------------------------------------------------
    def reference_blob(self):
        """All parameters flattened in the reference's `.parameters()` order (differentiable)."""
        return torch.cat([p.reshape(-1) for p in self.parameters()])

    def kernel_params(self):
        """Kernel-layout blob of the current parameters (cached until a parameter is modified)."""
        kp = self.__dict__.get("_lazy_kparams")
        if kp is not None:
            return kp
        key = tuple((p.data_ptr(), p._version) for p in self.parameters())
        if self._kcache is None or self._kcache[0] != key:
            with torch.no_grad():
                kp = _nh.pack(self.reference_blob().detach().float(), self.dim, self.K, self.hidden_dim, 1)
            self._kcache = (key, kp)
        return self._kcache[1]

    def load_kernel_params(self, kparams):
        """Write a kernel-layout blob (e.g. the result of the fused training loop) back into the
        nn.Parameters."""
        if self.__dict__.get("_lazy_kparams") is not None:
            self._lazy_kparams = kparams
            return
        blob = _nh.unpack(kparams, self.dim, self.K, self.hidden_dim, 1)
        off = 0
        with torch.no_grad():
            for p in self.parameters():
                p.copy_(blob[off:off + p.numel()].reshape(p.shape))
                off += p.numel()

    def _check(self, t, cols):
        if not t.is_cuda:
            raise RuntimeError("NSF_AR runs on a ROCm device only (no CPU path); got a %s tensor" % t.device)
        if t.dim() != 2 or t.shape[1] != cols:
            raise ValueError("expected a [n, %d] tensor, got %s" % (cols, tuple(t.shape)))
        return t.contiguous().float()

    # ---- reference API -----------------------------------------------------------------------
    def forward(self, x: torch.Tensor):
        x = self._check(x, self.dim)
        lazy = self.__dict__.get("_lazy_kparams") is not None
        if torch.is_grad_enabled() and (x.requires_grad or (not lazy and any(p.requires_grad for p in self.parameters()))):
            z, ld = _NSFFunction.apply(x, self.reference_blob(), self.dim, self.K, self.hidden_dim, self.B)
        else:
            z, ld, _ = _nh.forward(x, self.kernel_params(), self.K, self.hidden_dim, self.B, 1)
        if self.reference_scramble:   # flows.py:88-93 evaluates dim-major and reshapes as (n, d)
            n, d = x.shape
            z = z.t().reshape(-1).reshape(n, d)
        return z, ld

    def inverse(self, z):
        z = self._check(z, self.dim)
        x, ld = _nh.inverse(z, None, self.kernel_params(), self.K, self.hidden_dim, self.B, 1, want_logdet=True)
        return x, ld

    def inverse_given_separator(self, z, x_s):
        sep = 0 if x_s is None else x_s.shape[1]
        if z.shape[1] + sep > self.dim:
            raise ValueError("separator + latent columns exceed the flow dimension")
        z = self._check(z, z.shape[1])
        if x_s is not None:
            x_s = self._check(x_s, sep)
        # sep + z.shape[1] may be smaller than self.dim: the leading dims of an autoregressive flow
        # are its marginal flow, evaluated in place through the layer stride of the full model
        return _nh.inverse(z, x_s, self.kernel_params(), self.K, self.hidden_dim, self.B, 1, model_D=self.dim)
