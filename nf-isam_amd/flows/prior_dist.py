"""flows.prior_dist — base densities of the flows (API of the reference's src/flows/prior_dist.py:5-70:
`CustomMultivariateNormal(dim, device)` and `MultivariateNormalVonmises(circular_dim_list, device)` with
`log_prob`, `sample`, `to`, `cpu`, `is_cpu`, `dim`, `_device`).

The base density of NF-iSAM's flows is the standard normal, for which neither a covariance
factorisation nor torch.distributions bookkeeping is needed: log N(z; 0, I) = -|z|^2/2 - D/2 log 2pi and
sampling is `randn`.  (The fused kernels evaluate this term themselves; these classes serve host code
and the module surface.)  Constructing one costs nothing, which matters because the solver creates
two per trained clique.
"""
import math

import torch

_LOG_2PI = math.log(2.0 * math.pi)


class CustomMultivariateNormal(object):
    """N(0, I_dim) bound to a device string."""

    def __init__(self, dim: int, device: str = "cpu") -> None:
        self._dim = int(dim)
        self._device = str(device)

    # -- density / sampling ----------------------------------------------------------------------
    def log_prob(self, value: torch.Tensor) -> torch.Tensor:
        if value.shape[-1] != self._dim:
            raise ValueError("expected last dimension %d, got %d" % (self._dim, value.shape[-1]))
        return -0.5 * (value * value).sum(-1) - 0.5 * self._dim * _LOG_2PI

    def sample(self, sample_shape=torch.Size()) -> torch.Tensor:
        shape = tuple(sample_shape) + (self._dim,)
        return torch.randn(shape, device=self._device)

    def rsample(self, sample_shape=torch.Size()) -> torch.Tensor:
        return self.sample(sample_shape)

    # -- the attributes the flow container and the solver look at -------------------------------------
    @property
    def dim(self) -> int:
        return self._dim

    @property
    def mean(self) -> torch.Tensor:
        return torch.zeros(self._dim, device=self._device)

    @property
    def covariance_matrix(self) -> torch.Tensor:
        return torch.eye(self._dim, device=self._device)

    def is_cpu(self) -> bool:
        return self._device == "cpu"

    def to(self, device) -> "CustomMultivariateNormal":
        return CustomMultivariateNormal(self._dim, str(device))

    def cpu(self) -> "CustomMultivariateNormal":
        return self.to("cpu")

    def __repr__(self) -> str:
        return "CustomMultivariateNormal(dim=%d, device=%r)" % (self._dim, self._device)


class MultivariateNormalVonmises(object):
    """Independent columns: N(0, 1) for Euclidean ones, VonMises(0, 1) for circular ones.  In the
    reference this prior is only reachable through the undefined `NSF_AR_CS` flow type;
    it is kept because it is part of the module surface."""

    def __init__(self, circular_dim_list, device="cpu") -> None:
        self._circular = [bool(c) for c in circular_dim_list]
        self._device = str(device)
        self._vm = None

    def _vonmises(self):
        if self._vm is None:
            self._vm = torch.distributions.VonMises(torch.zeros(1, device=self._device),
                                                    torch.ones(1, device=self._device))
        return self._vm

    def sample(self, sample_shape: tuple) -> torch.Tensor:
        n = int(sample_shape[0])
        out = torch.randn(n, len(self._circular), device=self._device)
        for c, circ in enumerate(self._circular):
            if circ:
                out[:, c] = self._vonmises().sample((n,))[:, 0]
        return out

    def log_prob(self, x: torch.Tensor) -> torch.Tensor:
        if x.shape[1] != len(self._circular):
            raise ValueError("expected %d columns" % len(self._circular))
        total = torch.zeros(x.shape[0], device=x.device)
        for c, circ in enumerate(self._circular):
            col = x[:, c]
            total = total + (self._vonmises().log_prob(col) if circ else -0.5 * col * col - 0.5 * _LOG_2PI)
        return total

    @property
    def dim(self) -> int:
        return len(self._circular)

    def is_cpu(self) -> bool:
        return self._device == "cpu"

    def to(self, device) -> "MultivariateNormalVonmises":
        # (the reference returns a CustomMultivariateNormal with a list as `dim` here, prior_dist.py:69-70)
        return MultivariateNormalVonmises(self._circular, str(device))

    def cpu(self) -> "MultivariateNormalVonmises":
        return self.to("cpu")
