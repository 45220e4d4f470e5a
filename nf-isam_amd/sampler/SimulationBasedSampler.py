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

    def plan(self):
        """The simulation schedule of the clique, independent of the sample values: a list of steps
            ("prior", f)            draw all variables of a prior factor
            ("draw", f, dst)        draw variable `dst` of binary factor f from its other (already drawn) end
            ("observe", f)          simulated measurement of a binary factor whose two ends are drawn
                                    (binary factors include the two-hypothesis BinaryFactorWithNullHypo)
            ("assoc_obs", f)        simulated measurement of a k-way association factor
            ("assoc_observer", f)   draw the observer of an association factor from its candidates
        (reference: src/sampler/SimulationBasedSampler.py:14-133)."""
        priors, binaries, null_hypo, assoc = unpack_prior_binary_nh_da_factors(self.factors)
        steps, have = [], set()
        for f in priors:                              # assumes priors do not overlap
            steps.append(("prior", f))
            have.update(f.vars)
        queue = list(binaries)
        pending_nh = list(null_hypo)       # possibly-outlier factors are scheduled once the plain binaries are used up
        deferred = []          # factors that could only be simulated "small -> large" (landmark -> pose)
        stalled = 0
        while queue or pending_nh:
            if not queue:
                queue, pending_nh = pending_nh, []
            f = queue.pop(0)
            have1, have2 = f.var1 in have, f.var2 in have
            if have1 and have2:
                steps.append(("observe", f))
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
                steps.append(("draw", f, dst))
                have.add(dst)
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
            if all(v in have for v in f.vars):
                steps.append(("assoc_obs", f))
            else:
                missing = [v for v in f.vars if v not in have]
                if missing == [f.observer_var]:
                    steps.append(("assoc_observer", f))
                    have.add(f.observer_var)
                else:
                    raise ValueError("Some variables of the data association have not been sampled: " +
                                     " ".join(str(v.name) for v in missing))
        for f in deferred:
            if f.var1 in have and f.var2 in have:
                steps.append(("observe", f))
            else:
                missing = [str(v.name) for v in f.vars if v not in have]
                raise ValueError("Some variables have not been sampled: " + " ".join(missing) +
                                 ". Consider using a different variable elimination ordering.")
        return steps

    def sample(self, num_samples: int, backend=None):
        """-> (batch [n, D], variable ordering incl. observation variables, true observations).  `backend` draws the
        columns: the factors' numpy methods by default (`HostSimulationBackend`); a backend whose arrays live on the device keeps
        them on the GPU (the batch is then a device tensor); a backend with `run_plan` executes the whole schedule
        itself (`FusedSimulationBackend`: one kernel per clique)."""
        be = backend if backend is not None else HostSimulationBackend
        steps = self.plan()
        if hasattr(be, "run_plan"):
            return be.run_plan(steps, self.vars, num_samples)
        drawn, obs_cols, obs_vars, true_obs = {}, [], [], []
        for step in steps:
            kind, f = step[0], step[1]
            if kind == "prior":
                s = be.prior(f, num_samples)
                col = 0
                for v in f.vars:
                    drawn[v] = s[:, col:col + v.dim]
                    col += v.dim
            elif kind == "draw":
                dst = step[2]
                drawn[dst] = be.binary(f, var1=None, var2=drawn[f.var2]) if dst == f.var1 else \
                    be.binary(f, var1=drawn[f.var1], var2=None)
            elif kind == "observe":
                true_obs.append(np.asarray(f.observation, dtype=np.float64).ravel())
                obs_cols.append(be.binary(f, var1=drawn[f.var1], var2=drawn[f.var2]))
                obs_vars.append(f.observation_var)
            elif kind == "assoc_obs":
                true_obs.append(np.asarray(f.observation, dtype=np.float64).ravel())
                obs_cols.append(be.assoc_observations(f, drawn))
                obs_vars.append(f.observation_var)
            else:
                drawn[f.observer_var] = be.assoc_observer(f, drawn)
        cols = obs_cols + [drawn[v] for v in self.vars]
        local_samples = be.hstack(cols, num_samples)
        unused_obs = np.concatenate(true_obs) if true_obs else np.array([])
        return local_samples, obs_vars + list(self.vars), unused_obs
