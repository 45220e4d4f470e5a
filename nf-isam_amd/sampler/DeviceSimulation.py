"""sampler.DeviceSimulation — the clique training-batch simulator on the GPU.

`SimulationBasedSampler` decides WHAT to simulate (priors first, then binary factors from a sampled end to the
other, simulated measurements for loop-closing factors, k-way association factors); the draws themselves go through
a backend.  The host backend calls the factors' numpy `sample` methods (factors/Factors.py); the device backend
compiles the schedule into the op list of ONE kernel (`nfisam_simulate_clique`, csrc/clique_sim.hip), so that a clique's
batch is simulated, normalised (`nfisam_normalize_columns`) and trained without touching the host:

  children's flows (FlowsPriorFactor)   conditional sampling kernel, samples stay on the device
  UnarySE2ApproximateGaussianPriorFactor    x = prior * Exp(eps)                          Factors.py (reference :725-743)
  SE2RelativeGaussianLikelihoodFactor       T_j = T_i * (obs * Exp(eps)), either direction (reference :1196-1317)
  SE2R2RangeGaussianLikelihoodFactor        ring around the sampled end / simulated range  (reference :2575-2649)
  AmbiguousDataAssociationFactor            mixture over candidate landmarks              (reference :3146-3157,3260-3276)
  BinaryFactorWithNullHypo (range)          regular / inflated-sigma mixture              (reference :3300-3380)
  UnaryR2GaussianPriorFactor, UnaryR2RangeGaussianPriorFactor, R2RelativeGaussianLikelihoodFactor,
  R2RangeGaussianLikelihoodFactor           the R2 family of the toy examples             (reference :362, :451, :912, :2026)

Random numbers: Philox4x32-10 keyed per clique from numpy's global RNG.  A factor type the backend does not know
raises `DeviceSimulationUnsupported`; the solver then simulates that clique on the host.
"""
from typing import List

import numpy as np
import torch

from factors.Factors import (AmbiguousDataAssociationFactor, BinaryFactorWithNullHypo,
                             R2RelativeGaussianLikelihoodFactor, SE2R2RangeGaussianLikelihoodFactor,
                             SE2RelativeGaussianLikelihoodFactor, UnaryR2GaussianPriorFactor,
                             UnaryR2RangeGaussianPriorFactor, UnarySE2ApproximateGaussianPriorFactor)


class DeviceSimulationUnsupported(NotImplementedError):
    pass


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
