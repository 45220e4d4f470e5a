"""sampler.SimulationBasedSampler — ancestral simulation of a clique's training batch
(reference: src/sampler/SimulationBasedSampler.py:10-133).

Given the factors inside a clique, draw n joint samples of its variables by forward simulation
(priors first — including the children's trained flows wrapped as `FlowsPriorFactor` — then
binary factors from an already-sampled end to the other), and for every factor whose two ends are
both sampled already (a "loop-closing" factor) draw a *simulated measurement* column instead.
The batch is laid out [simulated observations | clique variables in pattern order], the true
measurement values are returned separately: conditioning the trained flow on them gives the
posterior.
"""
from typing import List

import numpy as np

from factors.utils import unpack_prior_binary_nh_da_factors
from sampler.DeviceSimulation import HostSimulationBackend


class SimulationBasedSampler:
    def __init__(self, factors: List, vars: List):
        self.factors = factors
        self.vars = vars

    def sample(self, num_samples: int, backend=None):
        """-> (batch [n, D], variable ordering incl. observation variables, true observations).  `backend` draws the
        columns: the factors' numpy methods by default, `sampler.DeviceSimulation.TorchSimulationBackend` keeps
        them on the GPU (the batch is then a device tensor)."""
        be = backend if backend is not None else HostSimulationBackend
        priors, binaries, null_hypo, assoc = unpack_prior_binary_nh_da_factors(self.factors)
        if null_hypo:
            raise NotImplementedError("null-hypothesis factors are not rebuilt (SURVEY.md §8 f-2)")
        drawn = {}
        for f in priors:                              # assumes priors do not overlap
            s = be.prior(f, num_samples)
            col = 0
            for v in f.vars:
                drawn[v] = s[:, col:col + v.dim]
                col += v.dim
        obs_cols, obs_vars, true_obs = [], [], []

        def observe(f):
            true_obs.append(np.asarray(f.observation, dtype=np.float64).ravel())
            obs_cols.append(be.binary(f, var1=drawn[f.var1], var2=drawn[f.var2]))
            obs_vars.append(f.observation_var)

        queue = list(binaries)
        deferred = []          # factors that could only be simulated "small -> large" (landmark -> pose)
        stalled = 0
        while queue:
            f = queue.pop(0)
            have1, have2 = f.var1 in drawn, f.var2 in drawn
            if have1 and have2:
                observe(f)
                stalled = 0
            elif have1 or have2:
                src, dst = (f.var1, f.var2) if have1 else (f.var2, f.var1)
                if src.dim < dst.dim:
                    # never simulate a pose from a landmark; retry later, give up if nothing else is left
                    if not queue:
                        deferred.append(f)
                    else:
                        queue.append(f)
                        stalled += 1
                        if stalled > len(queue):
                            deferred.extend(queue)
                            queue = []
                    continue
                drawn[dst] = be.binary(f, var1=drawn[f.var1], var2=None) if have1 else be.binary(f, var1=None,
                                                                                                var2=drawn[f.var2])
                stalled = 0
            else:
                queue.append(f)
                stalled += 1
                if stalled > len(queue):
                    raise ValueError("Some factors connect variables that cannot be simulated: " +
                                     " ".join(str(q) for q in queue))
        # data-association factors (k-way): a simulated measurement if every end is sampled, otherwise
        # they may only be used to draw their observer (reference :99-113)
        for f in assoc:
            if all(v in drawn for v in f.vars):
                true_obs.append(np.asarray(f.observation, dtype=np.float64).ravel())
                obs_cols.append(be.assoc_observations(f, drawn))
                obs_vars.append(f.observation_var)
            else:
                missing = [v for v in f.vars if v not in drawn]
                if missing == [f.observer_var]:
                    drawn[f.observer_var] = be.assoc_observer(f, drawn)
                else:
                    raise ValueError("Some variables of the data association have not been sampled: " +
                                     " ".join(str(v.name) for v in missing))
        for f in deferred:
            if f.var1 in drawn and f.var2 in drawn:
                observe(f)
            else:
                missing = [str(v.name) for v in f.vars if v not in drawn]
                raise ValueError("Some variables have not been sampled: " + " ".join(missing) +
                                 ". Consider using a different variable elimination ordering.")
        cols = obs_cols + [drawn[v] for v in self.vars]
        local_samples = be.hstack(cols, num_samples)
        unused_obs = np.concatenate(true_obs) if true_obs else np.array([])
        return local_samples, obs_vars + list(self.vars), unused_obs
