"""flows.models — NormalizingFlowModel (reference: src/flows/models.py:4-40).

`forward` chains the flows and evaluates the prior log-probability.  When every flow is an
NSF_AR of the same shape and no autograd graph is needed, all layers run in ONE fused kernel
launch (layers chained in LDS); otherwise the flows are applied one after the other (each one a
kernel launch with an analytic-VJP backward)."""
import torch
import torch.nn as nn

import nfisam_hip as _nh
from flows.flows import NSF_AR


class NormalizingFlowModel(nn.Module):
    def __init__(self, prior, flows):
        super().__init__()
        self.prior = prior
        self.flows = nn.ModuleList(flows)
        self._prior_device_check = False

    # ---- fused-path helpers ------------------------------------------------------------------
    def _homogeneous(self):
        f0 = self.flows[0] if len(self.flows) else None
        return isinstance(f0, NSF_AR) and all(
            isinstance(f, NSF_AR) and (f.dim, f.K, f.hidden_dim, f.B) == (f0.dim, f0.K, f0.hidden_dim, f0.B)
            and not f.reference_scramble for f in self.flows)

    def kernel_params(self):
        """Kernel-layout blob of all layers, [L * kparam_count]."""
        return torch.cat([f.kernel_params() for f in self.flows])

    def load_kernel_params(self, kparams):
        Pk = kparams.numel() // len(self.flows)
        for l, f in enumerate(self.flows):
            f.load_kernel_params(kparams[l * Pk:(l + 1) * Pk])

    def _needs_grad(self, x):
        if not torch.is_grad_enabled():
            return False
        if x.requires_grad:
            return True
        for f in self.flows:
            if isinstance(f, NSF_AR) and f.__dict__.get("_lazy_kparams") is not None:
                continue          # frozen, solver-trained layer: nothing to differentiate
            if any(p.requires_grad for p in f.parameters()):
                return True
        return False

    # ---- layer-by-layer path (heterogeneous flows, or an autograd graph is wanted) ---------------
    def _chain(self, value, method_name, order):
        total = None
        for flow in order:
            value, ld = getattr(flow, method_name)(value)
            total = ld if total is None else total + ld
        if total is None:
            total = value.new_zeros(value.shape[0])
        return value, total

    def _bind_prior(self, like):
        # the prior follows the first batch's device (the reference's one-shot check, models.py:12-15)
        if self._prior_device_check:
            return
        dev = str(like.device)
        if self.prior._device != dev:
            self.prior = self.prior.to(dev)
        self._prior_device_check = True

    # ---- reference API: forward -> (z, prior_logprob, log_det); inverse -> (x, log_det) ---------
    def forward(self, x):
        self._bind_prior(x)
        if self._homogeneous() and not self._needs_grad(x):
            f0 = self.flows[0]
            z, log_det, lp = _nh.forward(f0._check(x, f0.dim), self.kernel_params(), f0.K, f0.hidden_dim, f0.B,
                                         len(self.flows), want_logprob=True)
            return z, lp - log_det, log_det           # the kernel's log-prob includes the log-det
        z, log_det = self._chain(x, "forward", self.flows)
        return z, self.prior.log_prob(z), log_det

    def inverse(self, z):
        if self._homogeneous():
            f0 = self.flows[0]
            return _nh.inverse(f0._check(z, f0.dim), None, self.kernel_params(), f0.K, f0.hidden_dim, f0.B,
                               len(self.flows), want_logdet=True)
        return self._chain(z, "inverse", list(self.flows)[::-1])

    def sample(self, n_samples):
        return self.inverse(self.prior.sample((int(n_samples),)))[0]
