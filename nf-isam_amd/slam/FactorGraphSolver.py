"""slam.FactorGraphSolver — the solver-side interface of the hot path (reference:
src/slam/FactorGraphSolver.py:27-107,360-400): argument object, the `ConditionalSampler` /
`CliqueSeparatorFactor` contracts and the three hooks a density back end implements.

Scope note: the Bayes-tree bookkeeping of the reference's FactorGraphSolver
(graph update, elimination, tree traversal) is host-side Python that *calls* the hot path; it is a
"next" row and not rebuilt here.  This module carries exactly the part the density back end
(slam.NFiSAM) plugs into, so a reference solver can use the back end unchanged."""
import json
from typing import List

import numpy as np


class SolverArgs:
    def __init__(self, elimination_method: str = "natural", posterior_sample_num: int = 500,
                 local_sample_num: int = 500, store_clique_samples: bool = False, local_sampling_method="direct",
                 adaptive_posterior_sampling=None, *args, **kwargs):
        self.elimination_method = elimination_method
        self.posterior_sample_num = posterior_sample_num
        self.store_clique_samples = store_clique_samples
        self.local_sampling_method = local_sampling_method
        self.local_sample_num = local_sample_num
        self.adaptive_posterior_sampling = adaptive_posterior_sampling

    def jsonStr(self):
        return json.dumps(self.__dict__)


class ConditionalSampler:
    def conditional_sample_given_observation(self, conditional_dim, obs_samples=None, sample_number=None):
        """Samples of the first `conditional_dim` columns after the columns fixed by `obs_samples`
        ([n, dim]); with `sample_number` instead, unconditional samples of the first columns."""
        raise NotImplementedError("Implementation depends on density estimation method.")


class CliqueSeparatorFactor:
    """Prior over a clique's separator variables induced by its trained density model."""

    def sample(self, num_samples: int, **kwargs):
        raise NotImplementedError("implementation depends on density models")

    @property
    def vars(self) -> List:
        raise NotImplementedError

    @property
    def dim(self) -> int:
        return sum(v.dim for v in self.vars)


class FactorGraphSolver:
    """Holds the per-clique dictionaries the density back end reads/writes
    (reference: FactorGraphSolver.py:79-104) and declares the hooks (…:360-400)."""

    def __init__(self, args: SolverArgs):
        self._args = args
        self._samples = {}
        self._clique_samples = {}
        self._clique_true_obs = {}
        self._clique_density_model = {}
        self._clique_variable_pattern = {}
        self._implicit_factors = {}
        self._temp_training_loss = {}

    def fit_clique_density_model(self, clique, samples, var_ordering, timer, *args, **kwargs) -> ConditionalSampler:
        raise NotImplementedError("Implementation depends on probabilistic modeling.")

    def root_clique_density_model_to_leaf(self, old_clique, new_clique, device) -> ConditionalSampler:
        raise NotImplementedError("Implementation depends on probabilistic modeling")

    def clique_density_to_separator_factor(self, separator_var_list, density_model,
                                           true_obs: np.ndarray) -> CliqueSeparatorFactor:
        raise NotImplementedError("Implementation depends on probabilistic modeling")
