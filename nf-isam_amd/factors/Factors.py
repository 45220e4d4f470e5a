"""factors.Factors — the factor types that feed the flow hot path in the range-only SLAM
configurations (reference: src/factors/Factors.py, 3489 lines).

Only what the clique training-batch simulator needs is restated: the class hierarchy used for
dispatch (`PriorFactor`, `BinaryFactor`, ...), `.fg` text (de)serialisation, and **batched**
`sample` methods for
    UnarySE2ApproximateGaussianPriorFactor   (reference :682-849)
    SE2RelativeGaussianLikelihoodFactor       (reference :1095-1478)
    SE2R2RangeGaussianLikelihoodFactor        (reference :2510-2751)
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


class KWayFactor(Factor):
    pass


class AmbiguousDataAssociationFactor(KWayFactor):
    """Placeholder for dispatch (k-way ambiguous association, reference :3192-3298): next row f-2."""


class BinaryFactorWithNullHypo(BinaryFactor):
    """Placeholder for dispatch (reference :3300-3462): next row f-2."""


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
class SE2RelativeGaussianLikelihoodFactor(LikelihoodFactor, BinaryFactor):
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


_FACTOR_CLASSES = {c.__name__: c for c in (UnarySE2ApproximateGaussianPriorFactor,
                                           SE2RelativeGaussianLikelihoodFactor,
                                           SE2R2RangeGaussianLikelihoodFactor)}
