"""sampler.DeviceSimulation — the clique training-batch simulator on the GPU.

`SimulationBasedSampler` decides WHAT to simulate (priors first, then binary factors from a sampled end to the
other, simulated measurements for loop-closing factors, k-way association factors); the draws themselves go through
a backend.  The host backend calls the factors' numpy `sample` methods (factors/Factors.py); this backend keeps
every column as a float32 tensor on the device, so that a clique's batch is simulated, normalised
(`nfisam_normalize_columns`) and trained without touching the host:

  children's flows (FlowsPriorFactor)   conditional sampling kernel, samples stay on the device
  UnarySE2ApproximateGaussianPriorFactor    x = prior * Exp(eps)                          Factors.py (reference :725-743)
  SE2RelativeGaussianLikelihoodFactor       T_j = T_i * (obs * Exp(eps)), either direction (reference :1196-1317)
  SE2R2RangeGaussianLikelihoodFactor        ring around the sampled end / simulated range  (reference :2575-2649)
  AmbiguousDataAssociationFactor            mixture over candidate landmarks              (reference :3146-3157,3260-3276)

Batched SE(2) algebra as in geometry/TwoDimension.py.  Random numbers come from torch's device generator
(`torch.manual_seed`).  A factor type this backend does not know raises `DeviceSimulationUnsupported`; the solver
then simulates that clique on the host.
"""
import math
from typing import Dict, List

import numpy as np
import torch

from factors.Factors import (AmbiguousDataAssociationFactor, BinaryFactorWithNullHypo,
                             R2RelativeGaussianLikelihoodFactor, SE2R2RangeGaussianLikelihoodFactor,
                             SE2RelativeGaussianLikelihoodFactor, UnaryR2GaussianPriorFactor,
                             UnaryR2RangeGaussianPriorFactor, UnarySE2ApproximateGaussianPriorFactor)


class DeviceSimulationUnsupported(NotImplementedError):
    pass


# ---- batched SE(2) on [n, 3] tensors (x, y, theta) ---------------------------------------------------
def _wrap(t):
    return torch.remainder(t + math.pi, 2.0 * math.pi) - math.pi


def se2_exp_t(v):
    w = v[:, 2]
    small = w.abs() < 1e-7
    ws = torch.where(small, torch.ones_like(w), w)
    a = torch.where(small, torch.ones_like(w), torch.sin(ws) / ws)
    b = torch.where(small, 0.5 * w, (1.0 - torch.cos(ws)) / ws)
    return torch.stack([a * v[:, 0] - b * v[:, 1], b * v[:, 0] + a * v[:, 1], _wrap(w)], 1)


def se2_compose_t(a, b):
    c, s = torch.cos(a[:, 2]), torch.sin(a[:, 2])
    return torch.stack([a[:, 0] + c * b[:, 0] - s * b[:, 1], a[:, 1] + s * b[:, 0] + c * b[:, 1],
                        _wrap(a[:, 2] + b[:, 2])], 1)


def se2_inverse_t(a):
    c, s = torch.cos(a[:, 2]), torch.sin(a[:, 2])
    return torch.stack([-(c * a[:, 0] + s * a[:, 1]), -(-s * a[:, 0] + c * a[:, 1]), _wrap(-a[:, 2])], 1)


class TorchSimulationBackend(object):
    """Draws for `SimulationBasedSampler.sample(..., backend=...)`; all arrays are float32 device tensors."""
    is_device = True

    def __init__(self, device):
        self.device = torch.device(device)

    # small host constants (Cholesky factors, poses) are uploaded once and kept on the factor object
    def _c(self, f, name, value):
        cache = f.__dict__.setdefault("_device_constants", {})
        key = (name, str(self.device))
        t = cache.get(key)
        if t is None:
            t = cache[key] = torch.as_tensor(np.asarray(value, dtype=np.float32)).to(self.device)
        return t

    def _noise(self, f, n):
        chol = self._c(f, "chol", f._chol)
        return torch.randn(n, chol.shape[0], device=self.device) @ chol.t()

    # ---- priors ---------------------------------------------------------------------------------
    def prior(self, f, n: int):
        if hasattr(f, "sample_on_device"):                      # a child clique's trained flow
            return f.sample_on_device(n)
        if isinstance(f, UnarySE2ApproximateGaussianPriorFactor) and f._correlated_R_t:
            pose = self._c(f, "pose", f._prior_pose.array).reshape(1, 3).expand(n, 3)
            return se2_compose_t(pose, se2_exp_t(self._noise(f, n)))
        raise DeviceSimulationUnsupported(type(f).__name__)

    # ---- binary likelihood factors -----------------------------------------------------------------
    def _relative(self, f, base, n):
        if not f._correlated_Rt:
            raise DeviceSimulationUnsupported("correlated_R_t=False")
        return se2_compose_t(base, se2_exp_t(self._noise(f, n)))

    def _ring(self, f, centers):
        n = centers.shape[0]
        r = float(f._observation[0]) + f._sigma * torch.randn(n, device=self.device)
        phi = (2.0 * torch.rand(n, device=self.device) - 1.0) * math.pi
        return centers + torch.stack([r * torch.cos(phi), r * torch.sin(phi)], 1)

    def binary(self, f, var1=None, var2=None):
        """var2 given -> var1 samples; var1 given -> var2 samples; both -> simulated observations."""
        if isinstance(f, SE2RelativeGaussianLikelihoodFactor):
            obs = self._c(f, "obs", f.observation).reshape(1, 3)
            if var1 is None:
                n = var2.shape[0]
                return se2_compose_t(var2, se2_inverse_t(self._relative(f, obs.expand(n, 3), n)))
            if var2 is None:
                n = var1.shape[0]
                return se2_compose_t(var1, self._relative(f, obs.expand(n, 3), n))
            return self._relative(f, se2_compose_t(se2_inverse_t(var1), var2), var1.shape[0])
        if isinstance(f, SE2R2RangeGaussianLikelihoodFactor):
            i1, i2 = list(f.var1.t_dim_indices), list(f.var2.t_dim_indices)
            if var1 is None:
                return self._ring(f, var2[:, i2])
            if var2 is None:
                return self._ring(f, var1[:, i1])
            d = var2[:, i2] - var1[:, i1]
            n = d.shape[0]
            return (torch.sqrt((d * d).sum(1)) + f._sigma * torch.randn(n, device=self.device)).reshape(n, 1)
        if isinstance(f, BinaryFactorWithNullHypo):
            ref = var1 if var1 is not None else var2
            n = ref.shape[0]
            out = None
            for (lo, hi), c in zip(self._split(f, n), f.components):
                if hi > lo:
                    part = self.binary(c, var1=None if var1 is None else var1[lo:hi], var2=None if var2 is None else var2[lo:hi])
                    if out is None:
                        out = torch.zeros(n, part.shape[1], device=self.device)
                    out[lo:hi] = part
            return out
        raise DeviceSimulationUnsupported(type(f).__name__)

    # ---- k-way association factors -------------------------------------------------------------------
    @staticmethod
    def _split(f, n):
        counts = np.random.multinomial(n, f.weights)          # host RNG: a handful of integers
        bounds = np.concatenate(([0], np.cumsum(counts)))
        return [(int(bounds[i]), int(bounds[i + 1])) for i in range(len(f.components))]

    def assoc_observations(self, f, drawn: Dict):
        if not isinstance(f, AmbiguousDataAssociationFactor):
            raise DeviceSimulationUnsupported(type(f).__name__)
        n = drawn[f.observer_var].shape[0]
        out = torch.zeros(n, f.measurement_dim, device=self.device)
        for (lo, hi), c in zip(self._split(f, n), f.components):
            if hi > lo:
                out[lo:hi] = self.binary(c, var1=drawn[c.var1][lo:hi], var2=drawn[c.var2][lo:hi])
        return out

    def assoc_observer(self, f, drawn: Dict):
        if not isinstance(f, AmbiguousDataAssociationFactor):
            raise DeviceSimulationUnsupported(type(f).__name__)
        n = drawn[f.observed_vars[0]].shape[0]
        out = torch.zeros(n, f.observer_var.dim, device=self.device)
        for (lo, hi), c in zip(self._split(f, n), f.components):
            if hi <= lo:
                continue
            if c.var1 == f.observer_var:
                out[lo:hi] = self.binary(c, var1=None, var2=drawn[c.var2][lo:hi])
            elif c.var2 == f.observer_var:
                out[lo:hi] = self.binary(c, var1=drawn[c.var1][lo:hi], var2=None)
            else:
                raise ValueError("None of the vars of component matches the observer var.")
        return out

    @staticmethod
    def hstack(cols: List, n: int):
        return torch.cat(cols, 1) if cols else torch.empty(n, 0)


class FusedSimulationBackend(object):
    """The whole schedule of a clique as ONE kernel (csrc/clique_sim.hip): the steps of
    `SimulationBasedSampler.plan()` are compiled into `nfisam_sim_op`s; only the children's flow messages are separate
    launches (conditional-sampling kernel).  Random numbers: Philox keyed by a 64-bit seed drawn from numpy's global
    RNG per clique (so `np.random.seed` makes runs repeatable)."""
    is_device = True

    def __init__(self, device):
        self.device = torch.device(device)

    @staticmethod
    def _nh_range(f):
        return isinstance(f, BinaryFactorWithNullHypo) and len(f.components) == 2 and \
            all(isinstance(c, SE2R2RangeGaussianLikelihoodFactor) for c in f.components)

    @staticmethod
    def _chol3(f):
        L = np.asarray(f._chol, dtype=np.float64)
        return [L[0, 0], L[1, 0], L[1, 1]]

    @staticmethod
    def _chol6(f):
        L = np.asarray(f._chol, dtype=np.float64)
        return [L[0, 0], L[1, 0], L[1, 1], L[2, 0], L[2, 1], L[2, 2]]

    def run_plan(self, steps, pattern_vars, n):
        import nfisam_hip as nh
        # column plan: [simulated observations in schedule order | clique variables in pattern order | scratch]
        obs_dim = {"observe": lambda f: f.observation_var.dim, "assoc_obs": lambda f: f.observation_var.dim}
        n_obs = sum(obs_dim[s[0]](s[1]) for s in steps if s[0] in obs_dim)
        col, off = {}, n_obs
        for v in pattern_vars:
            col[v] = off
            off += v.dim
        D_out = off

        def column(v):
            nonlocal off
            if v not in col:                  # drawn but not part of the batch: scratch column
                col[v] = off
                off += v.dim
            return col[v]

        ops, keep, obs_vars, true_obs, ocol = [], [], [], [], 0

        def emit(code, a=0, b=0, c=0, p=(), cand=(), k=0, src=0):
            op = nh.SimOp()
            op.code, op.a, op.b, op.c, op.k, op.src = code, int(a), int(b), int(c), int(k), int(src)
            for i, v in enumerate(p):
                op.p[i] = float(v)
            for i, v in enumerate(cand):
                op.cand[i] = int(v)
            ops.append(op)

        for step in steps:
            kind, f = step[0], step[1]
            if kind == "prior":
                if hasattr(f, "sample_on_device"):                  # a child clique's trained flow
                    t = f.sample_on_device(n).contiguous()
                    keep.append(t)
                    src_col = 0
                    for v in f.vars:
                        emit(nh.SIM_COPY, a=t.shape[1], b=src_col, c=column(v), k=v.dim, src=t.data_ptr())
                        src_col += v.dim
                elif isinstance(f, UnarySE2ApproximateGaussianPriorFactor) and f._correlated_R_t:
                    emit(nh.SIM_PRIOR_SE2, c=column(f.vars[0]), p=list(f._prior_pose.array) + self._chol6(f))
                elif isinstance(f, UnaryR2GaussianPriorFactor):
                    emit(nh.SIM_PRIOR_R2, c=column(f.vars[0]), p=list(f.mu) + self._chol3(f))
                elif isinstance(f, UnaryR2RangeGaussianPriorFactor):
                    emit(nh.SIM_PRIOR_R2_RING, c=column(f.vars[0]), p=list(f.center) + [f.mu, f.sigma])
                else:
                    raise DeviceSimulationUnsupported(type(f).__name__)
            elif kind == "draw":
                dst = step[2]
                src = f.var2 if dst == f.var1 else f.var1
                if isinstance(f, SE2RelativeGaussianLikelihoodFactor) and f._correlated_Rt:
                    emit(nh.SIM_REL_BWD if dst == f.var1 else nh.SIM_REL_FWD, a=column(src), c=column(dst),
                         p=list(f.observation) + self._chol6(f))
                elif isinstance(f, R2RelativeGaussianLikelihoodFactor):
                    emit(nh.SIM_REL_R2_BWD if dst == f.var1 else nh.SIM_REL_R2_FWD, a=column(src), c=column(dst),
                         p=list(f.observation) + self._chol3(f))
                elif isinstance(f, SE2R2RangeGaussianLikelihoodFactor) and dst.dim == 2:
                    emit(nh.SIM_RING, a=column(src), c=column(dst), p=[float(f._observation[0]), f._sigma])
                elif self._nh_range(f) and dst.dim == 2:
                    c0, c1 = f.components
                    emit(nh.SIM_NH_RING, a=column(src), c=column(dst),
                         p=[float(np.ravel(f.observation)[0]), c0._sigma, c1._sigma, float(f.weights[0])])
                else:
                    raise DeviceSimulationUnsupported(type(f).__name__)
            elif kind == "observe":
                true_obs.append(np.asarray(f.observation, dtype=np.float64).ravel())
                obs_vars.append(f.observation_var)
                if isinstance(f, SE2RelativeGaussianLikelihoodFactor) and f._correlated_Rt:
                    emit(nh.SIM_REL_OBS, a=column(f.var1), b=column(f.var2), c=ocol, p=[0, 0, 0] + self._chol6(f))
                elif isinstance(f, R2RelativeGaussianLikelihoodFactor):
                    emit(nh.SIM_REL_R2_OBS, a=column(f.var1), b=column(f.var2), c=ocol, p=[0, 0] + self._chol3(f))
                elif isinstance(f, SE2R2RangeGaussianLikelihoodFactor):
                    emit(nh.SIM_RANGE_OBS, a=column(f.var1), b=column(f.var2), c=ocol, p=[f._sigma])
                elif self._nh_range(f):
                    c0, c1 = f.components
                    emit(nh.SIM_NH_OBS, a=column(f.var1), b=column(f.var2), c=ocol,
                         p=[c0._sigma, c1._sigma, float(f.weights[0])])
                else:
                    raise DeviceSimulationUnsupported(type(f).__name__)
                ocol += f.observation_var.dim
            elif kind == "assoc_obs":
                comps = f.components
                if not (isinstance(f, AmbiguousDataAssociationFactor) and len(comps) <= 4 and
                        all(isinstance(c, SE2R2RangeGaussianLikelihoodFactor) for c in comps)):
                    raise DeviceSimulationUnsupported(type(f).__name__)
                true_obs.append(np.asarray(f.observation, dtype=np.float64).ravel())
                obs_vars.append(f.observation_var)
                cands = [column(c.var2 if c.var1 == f.observer_var else c.var1) for c in comps]
                cum = np.cumsum(np.asarray(f.weights, dtype=np.float64) / np.sum(f.weights))
                p = list(cum) + [1.0] * (4 - len(comps)) + [comps[0]._sigma]
                emit(nh.SIM_ADA_OBS, a=column(f.observer_var), c=ocol, cand=cands, k=len(comps), p=p)
                ocol += f.observation_var.dim
            else:
                raise DeviceSimulationUnsupported("association factor drawing its observer")
        if len(ops) > nh.SIM_MAX_OPS or not ops:
            raise DeviceSimulationUnsupported("schedule of %d ops" % len(ops))
        seed = int(np.random.randint(0, 2 ** 62))
        x = nh.simulate_clique(ops, n, D_out, off, seed, self.device)
        self._keep = keep                      # the messages are read by the enqueued kernel (same stream)
        unused_obs = np.concatenate(true_obs) if true_obs else np.array([])
        return x, obs_vars + list(pattern_vars), unused_obs


class HostSimulationBackend(object):
    """The factors' own numpy `sample` methods (float64 on the host)."""
    is_device = False

    @staticmethod
    def prior(f, n):
        return f.sample(n)

    @staticmethod
    def binary(f, var1=None, var2=None):
        return f.sample(var1=var1, var2=var2)

    @staticmethod
    def assoc_observations(f, drawn):
        return f.sample_observations({v: drawn[v] for v in f.vars})

    @staticmethod
    def assoc_observer(f, drawn):
        return f.sample_observer(drawn)

    @staticmethod
    def hstack(cols, n):
        return np.hstack(cols) if cols else np.empty((n, 0))
