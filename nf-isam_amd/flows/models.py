"""flows.models — NormalizingFlowModel (reference: src/flows/models.py:4-40).

`forward` chains the flows and evaluates the prior log-probability.  When every flow is an
NSF_AR of the same shape and no autograd graph is needed, all layers run in ONE fused kernel
launch (layers chained in LDS); otherwise the flows are applied one after the other (each one a
kernel launch with an analytic-VJP backward)."""
import math

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

    # ---- reference API -----------------------------------------------------------------------
    def forward(self, x):
        if not self._prior_device_check:
            if self.prior._device != x.device.__str__():
                self.prior = self.prior.to(x.device.__str__())
            self._prior_device_check = True
        if self._homogeneous() and not self._needs_grad(x):
            f0 = self.flows[0]
            xc = f0._check(x, f0.dim)
            z, log_det, lp = _nh.forward(xc, self.kernel_params(), f0.K, f0.hidden_dim, f0.B, len(self.flows),
                                         want_logprob=True)
            return z, lp - log_det, log_det
        m, _ = x.shape
        log_det = torch.zeros(m, device=x.device)
        for flow in self.flows:
            x, ld = flow.forward(x)
            log_det = log_det + ld
        z, prior_logprob = x, self.prior.log_prob(x)
        return z, prior_logprob, log_det

    def inverse(self, z):
        if self._homogeneous():
            f0 = self.flows[0]
            zc = f0._check(z, f0.dim)
            return _nh.inverse(zc, None, self.kernel_params(), f0.K, f0.hidden_dim, f0.B, len(self.flows),
                               want_logdet=True)
        m, _ = z.shape
        log_det = torch.zeros(m, device=z.device)
        for flow in self.flows[::-1]:
            z, ld = flow.inverse(z)
            log_det = log_det + ld
        return z, log_det

    def sample(self, n_samples):
        z = self.prior.sample((n_samples,), )
        x, _ = self.inverse(z)
        return x
