"""flows.prior_dist — base densities (reference: src/flows/prior_dist.py:5-70).

N(0, I) only needs `log_prob` / `sample`; the fused kernels evaluate the prior log-probability
themselves, these classes exist for the module surface and for host-side code."""
import torch
from torch.distributions import MultivariateNormal, Normal, VonMises


class CustomMultivariateNormal(MultivariateNormal):
    """Standard normal of dimension `dim` that remembers its device string."""

    def __init__(self, dim: int, device: str = "cpu") -> None:
        self._dim = dim
        self._device = device
        self._loc = torch.zeros(dim).to(device)
        self._scale_tril = torch.eye(dim).to(device)
        super().__init__(self._loc, scale_tril=self._scale_tril)

    def cpu(self):
        return CustomMultivariateNormal(dim=self._dim, device="cpu")

    def is_cpu(self):
        return self._device == "cpu"

    @property
    def dim(self) -> int:
        return self._dim

    def to(self, device: str):
        return CustomMultivariateNormal(dim=self._dim, device=str(device))


class MultivariateNormalVonmises(object):
    """Product of N(0,1) (Euclidean dims) and VonMises(0,1) (circular dims).  Only reachable from the
    reference's undefined `NSF_AR_CS` flow type; kept for the module surface."""

    def __init__(self, circular_dim_list, device="cpu") -> None:
        self._device = device
        self._circular_dim_list = list(circular_dim_list)
        self._dist = []
        for circular in self._circular_dim_list:
            if circular:
                self._dist.append(VonMises(loc=torch.tensor([0.0]).to(device),
                                           concentration=torch.tensor([1.0]).to(device)))
            else:
                self._dist.append(Normal(loc=torch.tensor([0.0]).to(device), scale=torch.tensor([1.0]).to(device)))

    def sample(self, sample_shape: tuple):
        cols = [d.sample((sample_shape[0],)) for d in self._dist]
        return torch.cat(cols, 1).to(self._device)

    def log_prob(self, x):
        assert len(self._dist) == x.shape[1]
        res = self._dist[0].log_prob(x[:, 0])
        for i in range(1, x.shape[1]):
            res = res + self._dist[i].log_prob(x[:, i])
        return res

    def cpu(self):
        return MultivariateNormalVonmises(self._circular_dim_list, device="cpu")

    def is_cpu(self):
        return self._device == "cpu"

    @property
    def dim(self) -> int:
        return len(self._circular_dim_list)

    def to(self, device: str):
        # the reference returns a CustomMultivariateNormal with a list as `dim` here (prior_dist.py:69-70,
        # a bug in dead code); the sensible behaviour is kept instead
        return MultivariateNormalVonmises(self._circular_dim_list, device=str(device))
