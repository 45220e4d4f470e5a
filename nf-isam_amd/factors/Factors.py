"""factors.Factors — the factor types that feed the flow hot path in the range-only SLAM
configurations (reference: src/factors/Factors.py, 3489 lines).

Only what the clique training-batch simulator needs is restated: the class hierarchy used for
dispatch (`PriorFactor`, `BinaryFactor`, ...), `.fg` text (de)serialisation, and **batched**
`sample` methods for
    UnarySE2ApproximateGaussianPriorFactor   (reference :682-849)
    SE2RelativeGaussianLikelihoodFactor       (reference :1095-1478)
    SE2R2RangeGaussianLikelihoodFactor        (reference :2510-2751)
    AmbiguousDataAssociationFactor            (reference :3043-3298; k-way mixture over candidate landmarks)
    BinaryFactorWithNullHypo                  (reference :3300-3462; possibly-outlier measurement)
    UnaryR2GaussianPriorFactor, UnaryR2RangeGaussianPriorFactor, R2RelativeGaussianLikelihoodFactor,
    R2RangeGaussianLikelihoodFactor           (reference :362, :451, :912, :2026; the toy range-only examples)
The reference draws noise with TransportMaps' `GaussianDistribution.rvs` and then loops over
samples building `SE2Pose` objects; here the noise comes from numpy's global RNG (seeded by the
example scripts exactly like the reference's) and the pose algebra is vectorised
(geometry.TwoDimension).  Density evaluation (`log_pdf`, gradients) of these factors is only used
by the nested-sampling baselines (out of scope) and is not provided.
"""
from typing import Iterable, List, Union

import numpy as np

from geometry.TwoDimension import SE2Pose, se2_compose, se2_exp, se2_inverse
from slam.Variables import R1Variable, R2Variable, SE2Variable, Variable, VariableType


# ---- class hierarchy used for dispatch ---------------------------------------------------------
class Factor(object):
    @property
    def vars(self) -> List[Variable]:
        raise NotImplementedError

    @property
    def dim(self) -> int:
        return sum(v.dim for v in self.vars)

    @property
    def is_gaussian(self) -> bool:
        return False

    def __str__(self) -> str:
        return "Factor " + self.__class__.__name__ + " " + " ".join(str(v.name) for v in self.vars)

    @classmethod
    def construct_from_text(cls, line: str, variables: Iterable[Variable]) -> "Factor":
        """`Factor <ClassName> <args...>` -> instance (reference :42-52)."""
        tok = line.strip().split()
        if tok[0] == "Factor":
            tok = tok[1:]
        klass = _FACTOR_CLASSES.get(tok[0])
        if klass is None:
            raise NotImplementedError("factor type %s is outside the rebuilt scope (SURVEY.md §8 f-2)" % tok[0])
        return klass.construct_from_text(" ".join(tok), variables)


class UnaryFactor(Factor):
    @property
    def var(self) -> Variable:
        return self.vars[0]


class BinaryFactor(Factor):
    @property
    def var1(self) -> Variable:
        return self.vars[0]

    @property
    def var2(self) -> Variable:
        return self.vars[1]


class PriorFactor(Factor):
    def sample(self, num_samples: int, **kwargs) -> np.ndarray:
        raise NotImplementedError


class LikelihoodFactor(Factor):
    @property
    def observation(self) -> np.ndarray:
        raise NotImplementedError


class ExplicitPriorFactor(PriorFactor):
    pass


class ImplicitPriorFactor(PriorFactor):
    """A prior that is only available through sampling, e.g. a trained clique density."""


class OdomFactor(object):
    """Marker of relative-motion factors between consecutive poses (reference :908)."""


class KWayFactor(Factor):
    @property
    def root_var(self) -> Variable:
        raise NotImplementedError

    @property
    def child_vars(self) -> List[Variable]:
        raise NotImplementedError


class BinaryFactorWithNullHypo(BinaryFactor):
    """Defined below, after the mixture base (reference :3300-3462); declared here for the dispatch order."""


def _gaussian_noise(cov_chol: np.ndarray, n: int) -> np.ndarray:
    return np.random.standard_normal((n, cov_chol.shape[0])) @ cov_chol.T


def _mat3(tok, start):
    return np.array([[float(tok[start + 3 * r + c]) for c in range(3)] for r in range(3)])


# ---- SE(2) prior -----------------------------------------------------------------------------
class UnarySE2ApproximateGaussianPriorFactor(ExplicitPriorFactor, UnaryFactor):
    """x = prior_pose * Exp(eps), eps ~ N(0, covariance) in the tangent space."""

    def __init__(self, var: Variable, prior_pose: Union[SE2Pose, np.ndarray], covariance: np.ndarray,
                 correlated_R_t: bool = True):
        if not isinstance(prior_pose, SE2Pose):
            prior_pose = SE2Pose(*prior_pose)
        assert var.dim == 3 and np.shape(covariance) == (3, 3)
        self._vars = [var]
        self._prior_pose = prior_pose
        self._covariance = np.array(covariance, dtype=np.float64)
        self._chol = np.linalg.cholesky(self._covariance)
        self._correlated_R_t = correlated_R_t

    @property
    def vars(self):
        return self._vars

    @property
    def observation(self):
        return self._prior_pose.array

    @property
    def mu(self):
        return self.observation

    @property
    def covariance(self):
        return self._covariance

    @property
    def is_gaussian(self):
        return True

    def sample(self, num_samples: int, **kwargs) -> np.ndarray:
        noise = _gaussian_noise(self._chol, num_samples)
        if self._correlated_R_t:
            return se2_compose(self._prior_pose.array, se2_exp(noise))
        theta = np.random.vonmises(mu=0.0, kappa=1.0 / self._covariance[2, 2], size=num_samples)
        out = np.empty((num_samples, 3))
        out[:, :2] = self._prior_pose.array[:2] + noise[:, :2]
        out[:, 2] = (self._prior_pose.theta + theta + np.pi) % (2 * np.pi) - np.pi
        return out

    @classmethod
    def construct_from_text(cls, line: str, variables):
        tok = line.strip().split()
        if tok[0] != cls.__name__:
            raise ValueError("The factor name is incorrect")
        var = {v.name: v for v in variables}[tok[1]]
        mat = _mat3(tok, 6)
        if tok[5] == "covariance":
            cov = mat
        elif tok[5] == "information":
            cov = np.linalg.inv(mat)
        else:
            raise ValueError("Either covariance or information should be specified")
        return cls(var=var, prior_pose=SE2Pose(float(tok[2]), float(tok[3]), float(tok[4])), covariance=cov)

    def __str__(self):
        c = self._covariance
        return " ".join(["Factor", self.__class__.__name__, str(self.var.name)] + [str(v) for v in self.mu] +
                        ["covariance"] + [str(c[r, k]) for r in range(3) for k in range(3)])


# ---- SE(2) relative pose (odometry / loop closure) -------------------------------------------------
class SE2RelativeGaussianLikelihoodFactor(LikelihoodFactor, BinaryFactor, OdomFactor):
    """T_j = T_i * (observation * Exp(eps)), eps ~ N(0, covariance)."""
    measurement_dim = 3
    measurement_type = SE2Variable

    def __init__(self, var1: Variable, var2: Variable, observation: Union[SE2Pose, np.ndarray],
                 covariance: np.ndarray = None, correlated_R_t: bool = True, information: np.ndarray = None):
        if isinstance(observation, (np.ndarray, list, tuple)):
            observation = SE2Pose(*observation)
        if covariance is None:
            covariance = np.linalg.inv(information)
        if not (var1.dim == var2.dim == 3):
            raise ValueError("Dimensionality of poses, relative pose and observation must be 3")
        self._vars = [var1, var2]
        self._observation = observation
        self._covariance = np.array(covariance, dtype=np.float64)
        self._chol = np.linalg.cholesky(self._covariance)
        self._correlated_Rt = correlated_R_t
        self._observation_var = SE2Variable(name="O" + str(var1.name) + str(var2.name),
                                            variable_type=VariableType.Measurement)

    @property
    def vars(self):
        return self._vars

    @property
    def observation_var(self):
        return self._observation_var

    @property
    def circular_dim_list(self):
        return self._observation_var.circular_dim_list

    @property
    def covariance(self):
        return self._covariance

    @property
    def noise_cov(self):
        return self._covariance

    @property
    def observation(self) -> np.ndarray:
        return self._observation.array

    def _noisy_relative(self, base: np.ndarray, n: int) -> np.ndarray:
        if not self._correlated_Rt:
            raise NotImplementedError("correlated_R_t=False is not used by the shipped configurations")
        return se2_compose(base, se2_exp(_gaussian_noise(self._chol, n)))

    def sample(self, var1: Union[np.ndarray, None] = None, var2: Union[np.ndarray, None] = None) -> np.ndarray:
        """var2 given -> var1 samples; var1 given -> var2 samples; both -> simulated observations."""
        if var1 is None:
            if var2 is None:
                raise ValueError("Samples of at least one variable must be specified")
            rel = self._noisy_relative(self.observation, var2.shape[0])
            return se2_compose(var2, se2_inverse(rel))
        if var2 is None:
            rel = self._noisy_relative(self.observation, var1.shape[0])
            return se2_compose(var1, rel)
        if var1.shape != var2.shape or var1.shape[1] != 3:
            raise ValueError("Dimensionality of variable 1 or variable 2 is wrong")
        return self._noisy_relative(se2_compose(se2_inverse(var1), var2), var1.shape[0])

    @classmethod
    def construct_from_text(cls, line: str, variables):
        tok = line.strip().split()
        if tok[0] != cls.__name__:
            raise ValueError("The factor name is incorrect")
        name_to_var = {v.name: v for v in variables}
        obs = SE2Pose(float(tok[3]), float(tok[4]), float(tok[5]))
        return cls(var1=name_to_var[tok[1]], var2=name_to_var[tok[2]], observation=obs, **{tok[6]: _mat3(tok, 7)})

    def __str__(self):
        c = self._covariance
        return " ".join(["Factor", self.__class__.__name__] + [str(v.name) for v in self.vars] +
                        [str(v) for v in self.observation] + ["covariance"] +
                        [str(c[r, k]) for r in range(3) for k in range(3)])


# ---- range between an SE(2)/R2 variable and an R2/SE(2) variable ---------------------------------
class SE2R2RangeGaussianLikelihoodFactor(LikelihoodFactor, BinaryFactor):
    """|t_2 - t_1| = observation + N(0, sigma^2); sampling one end from the other puts it on a
    ring of uniformly random bearing (the source of the non-Gaussian posteriors)."""
    measurement_dim = 1
    measurement_type = R1Variable

    def __init__(self, var1: Variable, var2: Variable, observation: Union[np.ndarray, float], sigma: float = 1.0):
        self._vars = [var1, var2]
        self._observation = observation if isinstance(observation, np.ndarray) else np.array([observation])
        self._sigma = float(sigma)
        self._observation_var = R1Variable(name="O" + str(var1.name) + str(var2.name),
                                           variable_type=VariableType.Measurement)

    @property
    def vars(self):
        return self._vars

    @property
    def observation_var(self):
        return self._observation_var

    @property
    def circular_dim_list(self):
        return self._observation_var.circular_dim_list

    @property
    def observation(self) -> np.ndarray:
        return self._observation

    @property
    def sigma(self) -> float:
        return self._sigma

    def _ring(self, centers: np.ndarray) -> np.ndarray:
        n = centers.shape[0]
        r = self._observation[0] + self._sigma * np.random.standard_normal(n)
        phi = np.random.uniform(-np.pi, np.pi, n)
        return centers + np.stack([r * np.cos(phi), r * np.sin(phi)], 1)

    def sample_var2_from_var1(self, var1_samples: np.ndarray) -> np.ndarray:
        if var1_samples.ndim != 2 or var1_samples.shape[1] != self.var1.dim:
            raise ValueError("The dimensionality of variable 1 is wrong")
        return self._ring(var1_samples[:, self.var1.t_dim_indices])

    def sample_var1_from_var2(self, var2_samples: np.ndarray) -> np.ndarray:
        if var2_samples.ndim != 2 or var2_samples.shape[1] != self.var2.dim:
            raise ValueError("The dimensionality of variable 2 is wrong")
        return self._ring(var2_samples[:, self.var2.t_dim_indices])

    def sample_observations(self, var1_samples: np.ndarray, var2_samples: np.ndarray) -> np.ndarray:
        d = var2_samples[:, self.var2.t_dim_indices] - var1_samples[:, self.var1.t_dim_indices]
        n = var1_samples.shape[0]
        return (np.sqrt((d ** 2).sum(1)) + self._sigma * np.random.standard_normal(n)).reshape(n, 1)

    def pdf(self, x: np.ndarray) -> np.ndarray:
        """Likelihood of the stored observation given joint samples x = [var1 | var2] (reference :2680-2700)."""
        d1 = self.var1.dim
        t1 = x[:, :d1][:, self.var1.t_dim_indices]
        t2 = x[:, d1:][:, self.var2.t_dim_indices]
        r = np.sqrt(((t2 - t1) ** 2).sum(1))
        return np.exp(-0.5 * ((r - self._observation[0]) / self._sigma) ** 2) / (np.sqrt(2 * np.pi) * self._sigma)

    def sample(self, var1=None, var2=None) -> np.ndarray:
        if var1 is None:
            if var2 is None:
                raise ValueError("Samples of at least one variable must be specified")
            return self.sample_var1_from_var2(var2)
        if var2 is None:
            return self.sample_var2_from_var1(var1)
        return self.sample_observations(var1, var2)

    @classmethod
    def construct_from_text(cls, line: str, variables):
        tok = line.strip().split()
        if tok[0] != cls.__name__:
            raise ValueError("The factor name is incorrect")
        name_to_var = {v.name: v for v in variables}
        return cls(var1=name_to_var[tok[1]], var2=name_to_var[tok[2]], observation=float(tok[3]),
                   sigma=float(tok[4]))

    def __str__(self):
        return " ".join(["Factor", self.__class__.__name__, str(self.var1.name), str(self.var2.name),
                         str(self.observation[0]), str(self.sigma)])


# ---- ambiguous data association: one measurement, k candidate partners ----------------------------
class BinaryFactorMixture(LikelihoodFactor):
    """Mixture over `observed_vars` of binary factors of one class between the observer and each
    candidate (reference :3043-3181).  Simulation splits the sample batch among the hypotheses with a
    multinomial draw (rows of a sample batch are exchangeable)."""

    def __init__(self, observer_var: Variable, observed_vars: List[Variable], weights: np.ndarray,
                 binary_factor_class, obs_arr: List, sigma_arr: List):
        weights = np.asarray(weights, dtype=np.float64)
        assert np.all(weights > 0) and len(weights) == len(obs_arr) == len(sigma_arr) == len(observed_vars)
        self.observer_var = observer_var
        seen, uniq = set(), []
        for v in observed_vars:
            if v not in seen:
                seen.add(v)
                uniq.append(v)
        self.observed_vars = uniq
        self._vars = [observer_var] + self.observed_vars
        self.weights = weights / weights.sum()
        self.observations = obs_arr
        self.sigmas = sigma_arr
        self.components = [binary_factor_class(observer_var, v, obs_arr[i], sigma_arr[i])
                           for i, v in enumerate(observed_vars)]
        self.var2idx, off = {}, 0
        for v in self._vars:
            self.var2idx[v] = np.arange(off, off + v.dim)
            off += v.dim
        self.comp2idx = {c: np.concatenate((self.var2idx[c.var1], self.var2idx[c.var2])) for c in self.components}

    @property
    def vars(self):
        return self._vars

    @property
    def observation_var(self):
        return self.components[0].observation_var

    @property
    def measurement_dim(self):
        return self.observation_var.dim

    def _split(self, n):
        counts = np.random.multinomial(n, self.weights)
        bounds = np.concatenate(([0], np.cumsum(counts)))
        return [(int(bounds[i]), int(bounds[i + 1])) for i in range(len(self.components))]

    def sample_observations(self, var_samples) -> np.ndarray:
        """Simulated measurements given samples of every variable of the factor."""
        n = var_samples[self.observer_var].shape[0]
        out = np.zeros((n, self.measurement_dim))
        for (lo, hi), c in zip(self._split(n), self.components):
            if hi > lo:
                out[lo:hi] = c.sample(var1=var_samples[c.var1][lo:hi], var2=var_samples[c.var2][lo:hi])
        return out

    def pdf(self, x: np.ndarray) -> np.ndarray:
        return sum(c.pdf(x[:, self.comp2idx[c]]) * w for c, w in zip(self.components, self.weights))

    def posterior_weights(self, var2x) -> np.ndarray:
        """Re-weight the association hypotheses with posterior samples (reference :3158-3181)."""
        x = np.concatenate([var2x[v] for v in self.vars], axis=1)
        lik = np.array([c.pdf(x[:, self.comp2idx[c]]) * w for c, w in zip(self.components, self.weights)])
        tot = lik.sum(0)
        hw = np.full_like(lik, 0.5)
        ok = tot > 0
        hw[:, ok] = lik[:, ok] / tot[ok]
        return hw.sum(1) / hw.sum()


class AmbiguousDataAssociationFactor(BinaryFactorMixture, KWayFactor):
    """One measurement taken by `observer_var` of ONE of `observed_vars` (reference :3192-3298)."""

    def __init__(self, observer_var: Variable, observed_vars: List[Variable], weights: np.ndarray,
                 binary_factor_class, observation, sigma):
        k = len(observed_vars)
        assert k == len(weights)
        super().__init__(observer_var, observed_vars, weights, binary_factor_class, [observation] * k, [sigma] * k)

    @property
    def observation(self) -> np.ndarray:
        return self.components[0].observation

    @property
    def root_var(self) -> Variable:
        return self.observer_var

    @property
    def child_vars(self) -> List[Variable]:
        return self.observed_vars

    def sample_observer(self, var2sample) -> np.ndarray:
        """Samples of the observer given samples of all candidates."""
        n = var2sample[self.observed_vars[0]].shape[0]
        out = np.zeros((n, self.observer_var.dim))
        for (lo, hi), c in zip(self._split(n), self.components):
            if hi <= lo:
                continue
            if c.var1 == self.observer_var:
                out[lo:hi] = c.sample(var1=None, var2=var2sample[c.var2][lo:hi])
            elif c.var2 == self.observer_var:
                out[lo:hi] = c.sample(var1=var2sample[c.var1][lo:hi], var2=None)
            else:
                raise ValueError("None of the vars of component matches the observer var.")
        return out

    @classmethod
    def construct_from_text(cls, line: str, variables):
        tok = line.strip().split()
        if tok[0] != cls.__name__:
            raise ValueError("The factor name is incorrect")
        name_to_var = {v.name: v for v in variables}
        i_obs, i_seen, i_w = tok.index("Observer") + 1, tok.index("Observed") + 1, tok.index("Weights") + 1
        i_cls, i_meas, i_sig = tok.index("Binary") + 1, tok.index("Observation") + 1, tok.index("Sigma") + 1
        observed = [name_to_var[t] for t in tok[i_seen:i_w - 1]]
        weights = np.array(tok[i_w:i_cls - 1], dtype=float)
        klass = _FACTOR_CLASSES[tok[i_cls]]
        if i_sig - i_meas - 1 != 1:
            raise NotImplementedError("vector-valued ambiguous measurements are not used by the shipped graphs")
        return cls(name_to_var[tok[i_obs]], observed, weights, klass, float(tok[i_meas]), float(tok[i_sig]))

    def __str__(self):
        return " ".join(["Factor", self.__class__.__name__, "Observer", str(self.observer_var.name), "Observed"] +
                        [str(v.name) for v in self.observed_vars] + ["Weights"] + [str(w) for w in self.weights] +
                        ["Binary", self.components[0].__class__.__name__, "Observation",
                         str(float(np.ravel(self.observation)[0])), "Sigma", str(self.components[0].sigma)])


_FACTOR_CLASSES = {c.__name__: c for c in (UnarySE2ApproximateGaussianPriorFactor,
                                           SE2RelativeGaussianLikelihoodFactor,
                                           SE2R2RangeGaussianLikelihoodFactor,
                                           AmbiguousDataAssociationFactor)}


# ---- the R2 family of the toy range-only examples (BASELINE config[2]; reference :362, :451/:2226, :912, :2026) ----
def _cov_of(covariance, precision):
    if covariance is not None:
        return np.array(covariance, dtype=np.float64)
    if precision is not None:
        return np.linalg.inv(np.array(precision, dtype=np.float64))
    raise ValueError("None of cov and info. were defined.")


def _mat2(tok, start):
    return np.array([[float(tok[start]), float(tok[start + 1])], [float(tok[start + 2]), float(tok[start + 3])]])


class UnaryR2GaussianPriorFactor(ExplicitPriorFactor, UnaryFactor):
    """x ~ N(mu, covariance) on the plane (reference :362-448)."""

    def __init__(self, var: Variable, mu: np.ndarray, covariance: np.ndarray = None, precision: np.ndarray = None):
        assert var.dim == 2
        self._vars = [var]
        self._mu = np.array(mu, dtype=np.float64).ravel()
        self._covariance = _cov_of(covariance, precision)
        self._chol = np.linalg.cholesky(self._covariance)

    @property
    def vars(self):
        return self._vars

    @property
    def mu(self):
        return self._mu

    @property
    def observation(self):
        return self._mu

    @property
    def covariance(self):
        return self._covariance

    @property
    def is_gaussian(self):
        return True

    def sample(self, num_samples: int, **kwargs) -> np.ndarray:
        return self._mu + _gaussian_noise(self._chol, num_samples)

    @classmethod
    def construct_from_text(cls, line: str, variables):
        tok = line.strip().split()
        if tok[0] != cls.__name__:
            raise ValueError("The factor name is incorrect")
        if tok[4] not in ("covariance", "precision"):
            raise ValueError("Must specify either covariance or precision")
        var = {v.name: v for v in variables}[tok[1]]
        return cls(var=var, mu=np.array([float(tok[2]), float(tok[3])]), **{tok[4]: _mat2(tok, 5)})

    def __str__(self):
        c = self._covariance
        return " ".join(["Factor", self.__class__.__name__, str(self.var.name), str(self._mu[0]), str(self._mu[1]),
                         "covariance", str(c[0, 0]), str(c[0, 1]), str(c[1, 0]), str(c[1, 1])])


class UnaryR2RangeGaussianPriorFactor(ExplicitPriorFactor, UnaryFactor):
    """x on a ring around `center`: radius ~ N(mu, sigma^2), uniform bearing (reference :451-533, :2226-2308;
    draws: src/stats/Distributions.py:125-130)."""

    def __init__(self, var: Variable, center: np.ndarray, mu: float, sigma: float):
        assert var.dim == 2
        self._vars = [var]
        self._center = np.array(center, dtype=np.float64).ravel()
        self._mu, self._sigma = float(mu), float(sigma)

    @property
    def vars(self):
        return self._vars

    @property
    def center(self):
        return self._center

    @property
    def mu(self):
        return self._mu

    @property
    def observation(self):
        return self._mu

    @property
    def sigma(self):
        return self._sigma

    def sample(self, num_samples: int, **kwargs) -> np.ndarray:
        r = self._mu + self._sigma * np.random.standard_normal(num_samples)
        phi = np.random.uniform(-np.pi, np.pi, num_samples)
        return self._center + np.stack([r * np.cos(phi), r * np.sin(phi)], 1)

    @classmethod
    def construct_from_text(cls, line: str, variables):
        tok = line.strip().split()
        if tok[0] != cls.__name__:
            raise ValueError("The factor name is incorrect")
        var = {v.name: v for v in variables}[tok[1]]
        if tok[2] == "center":                      # the form `__str__` writes (reference :486-491)
            return cls(var, np.array([float(tok[3]), float(tok[4])]), float(tok[6]), float(tok[8]))
        return cls(var, np.array([float(tok[2]), float(tok[3])]), float(tok[4]), float(tok[5]))   # reference :500-505

    def __str__(self):
        return " ".join(["Factor", self.__class__.__name__, str(self.var.name), "center", str(self._center[0]),
                         str(self._center[1]), "mu", str(self._mu), "sigma", str(self._sigma)])


class R2RelativeGaussianLikelihoodFactor(LikelihoodFactor, BinaryFactor, OdomFactor):
    """var2 - var1 = observation + N(0, covariance) (reference :912-1092)."""
    measurement_dim = 2
    measurement_type = R2Variable

    def __init__(self, var1: Variable, var2: Variable, observation: np.ndarray, covariance: np.ndarray = None,
                 precision: np.ndarray = None):
        if var1.dim != var2.dim:
            raise ValueError("The two variables must have the same dimensionality")
        if len(observation) != var1.dim:
            raise ValueError("The observation must have the same dimensionality as the two variables")
        self._vars = [var1, var2]
        self._observation = np.array(observation, dtype=np.float64).ravel()
        self._covariance = _cov_of(covariance, precision)
        self._chol = np.linalg.cholesky(self._covariance)
        self._observation_var = R2Variable(name="O" + str(var1.name) + str(var2.name),
                                           variable_type=VariableType.Measurement)

    @property
    def vars(self):
        return self._vars

    @property
    def observation(self):
        return self._observation

    @property
    def observation_var(self):
        return self._observation_var

    @property
    def circular_dim_list(self):
        return self._observation_var.circular_dim_list

    @property
    def covariance(self):
        return self._covariance

    @property
    def is_gaussian(self):
        return True

    def sample(self, var1: np.ndarray = None, var2: np.ndarray = None) -> np.ndarray:
        """var2 given -> var1 samples; var1 given -> var2 samples; both -> simulated observations (reference :998-1030)."""
        if var1 is None:
            if var2 is None:
                raise ValueError("Samples of at least one variable must be specified")
            return var2 - _gaussian_noise(self._chol, var2.shape[0]) - self._observation
        if var2 is None:
            return var1 + _gaussian_noise(self._chol, var1.shape[0]) + self._observation
        if var1.shape != var2.shape or var1.shape[1] != 2:
            raise ValueError("Dimensionality of variable 1 or variable 2 is wrong")
        return var2 - var1 + _gaussian_noise(self._chol, var1.shape[0])

    @classmethod
    def construct_from_text(cls, line: str, variables):
        tok = line.strip().split()
        if tok[0] != cls.__name__:
            raise ValueError("The factor name is incorrect")
        name_to_var = {v.name: v for v in variables}
        return cls(var1=name_to_var[tok[1]], var2=name_to_var[tok[2]], observation=np.array([float(tok[3]), float(tok[4])]),
                   **{tok[5]: _mat2(tok, 6)})

    def __str__(self):
        c = self._covariance
        return " ".join(["Factor", self.__class__.__name__, str(self.var1.name), str(self.var2.name),
                         str(self._observation[0]), str(self._observation[1]), "covariance",
                         str(c[0, 0]), str(c[0, 1]), str(c[1, 0]), str(c[1, 1])])


class R2RangeGaussianLikelihoodFactor(SE2R2RangeGaussianLikelihoodFactor):
    """|var2 - var1| = observation + N(0, sigma^2) between two points of the plane (reference :2026-2223): the same
    ring / simulated-range draws as the pose-landmark range factor (both variables' translation indices are 0, 1)."""

    def __init__(self, var1: Variable, var2: Variable, observation: Union[np.ndarray, float], sigma: float = 1.0):
        if var1.dim != 2 or var2.dim != 2:
            raise ValueError("R2RangeGaussianLikelihoodFactor connects two R2 variables")
        super().__init__(var1, var2, observation, sigma)


_FACTOR_CLASSES.update({c.__name__: c for c in (UnaryR2GaussianPriorFactor, UnaryR2RangeGaussianPriorFactor,
                                                R2RelativeGaussianLikelihoodFactor, R2RangeGaussianLikelihoodFactor)})


# ---- a binary factor that may be an outlier (reference :3300-3462) -------------------------------------
class _BinaryFactorWithNullHypo(BinaryFactorMixture, BinaryFactorWithNullHypo):
    """Two hypotheses about ONE measurement between var1 and var2: the regular factor (sigma) and a "null" one that
    keeps the measurement but inflates its noise by `null_sigma_scale`; `weights` = (regular, null)."""

    def __init__(self, var1: Variable, var2: Variable, weights: np.ndarray, binary_factor_class, observation, sigma,
                 null_sigma_scale: float = 10.0):
        assert len(weights) == 2
        self.null_sigma_scale = float(null_sigma_scale)
        super().__init__(var1, [var2, var2], weights, binary_factor_class, [observation] * 2,
                         [sigma, sigma * self.null_sigma_scale])

    @property
    def var1(self) -> Variable:
        return self.observer_var

    @property
    def var2(self) -> Variable:
        return self.observed_vars[0]

    @property
    def observation(self) -> np.ndarray:
        return self.components[0].observation

    def _mix(self, n, draw):
        out = None
        for (lo, hi), c in zip(self._split(n), self.components):
            if hi > lo:
                part = draw(c, lo, hi)
                if out is None:
                    out = np.zeros((n, part.shape[1]))
                out[lo:hi] = part
        return out

    def sample(self, var1=None, var2=None) -> np.ndarray:
        """var2 given -> var1 samples; var1 given -> var2 samples; both -> simulated observations."""
        if var1 is None:
            if var2 is None:
                raise ValueError("Samples of at least one variable must be specified")
            return self._mix(var2.shape[0], lambda c, lo, hi: c.sample(var1=None, var2=var2[lo:hi]))
        if var2 is None:
            return self._mix(var1.shape[0], lambda c, lo, hi: c.sample(var1=var1[lo:hi], var2=None))
        return self._mix(var1.shape[0], lambda c, lo, hi: c.sample(var1=var1[lo:hi], var2=var2[lo:hi]))

    @classmethod
    def construct_from_text(cls, line: str, variables):
        tok = line.strip().split()
        if tok[0] != "BinaryFactorWithNullHypo":
            raise ValueError("The factor name is incorrect")
        name_to_var = {v.name: v for v in variables}
        i_obsr, i_obsd, i_w = tok.index("Observer") + 1, tok.index("Observed") + 1, tok.index("Weights") + 1
        i_cls, i_obs, i_sig = tok.index("Binary") + 1, tok.index("Observation") + 1, tok.index("Sigma") + 1
        i_null = tok.index("NullSigmaScale") + 1
        klass = _FACTOR_CLASSES[tok[i_cls]]
        if klass.measurement_dim != 1:
            raise NotImplementedError("null-hypothesis factors are rebuilt for scalar measurements (range factors)")
        weights = np.array(tok[i_w:i_cls - 1], dtype=float)
        return cls(name_to_var[tok[i_obsr]], name_to_var[tok[i_obsd]], weights, klass, float(tok[i_obs]),
                   float(tok[i_sig]), float(tok[i_null]))

    def __str__(self):
        return " ".join(["Factor", "BinaryFactorWithNullHypo", "Observer", str(self.var1.name), "Observed",
                         str(self.var2.name), str(self.var2.name), "Weights"] + [str(w) for w in self.weights] +
                        ["Binary", self.components[0].__class__.__name__, "Observation",
                         str(float(np.ravel(self.observation)[0])), "Sigma", str(self.components[0].sigma),
                         "NullSigmaScale", str(self.null_sigma_scale)])


_BinaryFactorWithNullHypo.__name__ = "BinaryFactorWithNullHypo"
_BinaryFactorWithNullHypo.__qualname__ = "BinaryFactorWithNullHypo"
BinaryFactorWithNullHypo = _BinaryFactorWithNullHypo           # the name the reference exports (isinstance dispatch keeps working)
_FACTOR_CLASSES["BinaryFactorWithNullHypo"] = BinaryFactorWithNullHypo
