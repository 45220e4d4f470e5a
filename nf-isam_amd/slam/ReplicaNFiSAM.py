"""slam.ReplicaNFiSAM — R independent NF-iSAM runs advanced in lock-step on ONE GPU.

The shipped large examples use the `pose_first` ordering, whose working tree per update is a chain of ~5 cliques
(SURVEY.md §0.4): a parent needs its child's samples, so the cliques of one run are trained strictly one after the other
and a single clique (n = 2000, D = 15: 945 waves) cannot fill 256 CUs — the launch is latency-bound.  The reference
itself loops over eight dataset variants one after the other (example/slam/plaza_dataset/run_nfisam.py:11-21).
Independent runs (seeds, noise / data-association variants, datasets) have no such dependency: here R solvers are
stepped together, each one runs its own host bookkeeping and clique simulation up to its next
`fit_clique_density_model`, and the R pending cliques are trained by ONE batched launch sequence (grid.y = replica,
`NFiSAM.train_prepared`).  That moves the training from the latency regime (one clique per launch) to the throughput
regime without changing any replica's result: every replica owns its random streams (numpy, python,
torch host + device generator states are swapped in and out around its turns), its batch, parameters, Adam state and
early-stop decision, so replica r reproduces the sequential run with seed r — bit for bit when both use the same
kernel family (`NFISAM_TRAIN`), to kernel rounding (loss 5e-4) otherwise.
Replicas may hold different graphs: a replica that has fewer cliques to train in an update simply drops out of the
later batches of that update.
"""
import contextlib
import random
import time
from typing import Dict, List

import numpy as np
import torch

from slam.NFiSAM import NFiSAM, NFiSAMArgs


class _ReplicaSolver(NFiSAM):
    """NFiSAM whose upward pass is a generator: it yields the prepared fit instead of training it."""

    def fit_tree_steps(self, timer: List[float] = None, clique_dim_timer: List[List[float]] = None):
        """`FactorGraphSolver.fit_tree_density_models` (reference :409-477) with the training handed to the caller:
        yields `prep` (see `NFiSAM.prepare_fit`); the caller fills prep["trained"/"iters"/"iter_loss"] and resumes."""
        self._temp_training_loss = {}
        clique_ordering = self._working_bayes_tree.clique_ordering()
        t_begin = time.time()
        while clique_ordering:
            clique = clique_ordering.pop()
            if clique in self._clique_density_model:
                if clique_dim_timer is not None:
                    clique_dim_timer.append([clique.dim, time.time() - t_begin])
                continue
            t0 = time.time()
            local_samples, sample_var_ordering, true_obs = self.clique_training_sampler(
                clique, num_samples=self._args.local_sample_num, method=self._args.local_sampling_method)
            if timer is not None:
                timer.append(time.time() - t0)
            self._clique_true_obs[clique] = true_obs
            if self._args.store_clique_samples:
                self._clique_samples[clique] = local_samples if isinstance(local_samples, np.ndarray) \
                    else local_samples.cpu().numpy()
            prep = self.prepare_fit(clique, local_samples, sample_var_ordering)
            if prep["testing_data"] is not None:
                raise NotImplementedError("replica batching trains on the full batch (training_set_frac = 1)")
            yield prep
            model = self.finish_fit(prep)
            self._clique_density_model[clique] = model
            new_separator_factor = None
            if clique.separator:
                separator_list = sorted(clique.separator, key=lambda x: self._reverse_ordering_map[x])
                new_separator_factor = self.clique_density_to_separator_factor(separator_list, model, true_obs)
                self._implicit_factors[clique] = new_separator_factor
            self._working_graph = self._working_graph.eliminate_clique_variables(clique=clique,
                                                                                 new_factor=new_separator_factor)
            if clique_dim_timer is not None:
                clique_dim_timer.append([clique.dim, time.time() - t_begin])


class ReplicaNFiSAM:
    def __init__(self, args: NFiSAMArgs, seeds: List[int]):
        if not seeds:
            raise ValueError("need at least one replica seed")
        self.solvers = [_ReplicaSolver(args) for _ in seeds]
        self.seeds = list(seeds)
        self._dev = torch.cuda.current_device()
        self._states = []
        outer = self._grab()
        for s in seeds:                      # the state a sequential run starts from after seeding with s
            random.seed(s); np.random.seed(s); torch.manual_seed(s)
            self._states.append(self._grab())
        self._put(outer)
        self.last_batches: List[int] = []    # cliques per batched launch sequence of the last update
        self._streams: List[torch.cuda.Stream] = []

    def __len__(self):
        return len(self.solvers)

    # ---- per-replica random streams -----------------------------------------------------------------------------
    def _grab(self):
        return (random.getstate(), np.random.get_state(), torch.get_rng_state(), torch.cuda.get_rng_state(self._dev))

    def _put(self, st):
        random.setstate(st[0]); np.random.set_state(st[1]); torch.set_rng_state(st[2])
        torch.cuda.set_rng_state(st[3], self._dev)

    @contextlib.contextmanager
    def turn(self, r: int):
        """Everything replica r does that may draw random numbers runs inside its turn."""
        outer = self._grab()
        self._put(self._states[r])
        try:
            yield self.solvers[r]
        finally:
            self._states[r] = self._grab()
            self._put(outer)

    # ---- staging ----------------------------------------------------------------------------------------------
    def add_node(self, var, replica: int = None):
        for s in (self.solvers if replica is None else [self.solvers[replica]]):
            s.add_node(var)
        return self

    def add_factor(self, factor, replica: int = None):
        """Factors are immutable measurements: the replicas of one graph may share the objects."""
        for s in (self.solvers if replica is None else [self.solvers[replica]]):
            s.add_factor(factor)
        return self

    # ---- one incremental update of every replica -----------------------------------------------------------------
    def update(self, timers: List[List[float]] = None) -> List[Dict]:
        """`update_physical_and_working_graphs` + `incremental_inference` of all replicas
        (reference per replica: FactorGraphSolver.py:803-808).  -> posterior samples per replica."""
        R = len(self.solvers)
        timers = timers if timers is not None else [[] for _ in range(R)]
        gens = []
        prof = self.__dict__.setdefault("profile", {"graphs": 0.0, "simulate+prepare": 0.0, "train": 0.0, "posterior": 0.0})
        t_ph = time.time()
        for r in range(R):
            with self.turn(r) as s:
                s.update_physical_and_working_graphs(timer=timers[r])
                gens.append(s.fit_tree_steps(timer=timers[r]))
        live = list(range(R))
        self.last_batches = []
        prof["graphs"] += time.time() - t_ph
        while live:
            t_ph = time.time()
            preps, still = [], []
            for r in live:
                with self.turn(r):
                    try:
                        preps.append(next(gens[r]))     # replica r: finish its previous clique, simulate the next one
                        still.append(r)
                    except StopIteration:
                        pass
            live = still
            torch.cuda.synchronize()
            prof["simulate+prepare"] += time.time() - t_ph
            if preps:
                t0 = time.time()
                self.solvers[0].train_prepared(preps)   # ONE batched launch sequence for the pending cliques
                torch.cuda.synchronize()
                dt = time.time() - t0
                for r in live:
                    timers[r].append(dt / len(preps))   # the reference's per-clique training timer: this replica's share
                self.last_batches.append(len(preps))
                prof["train"] += dt
        t_ph = time.time()
        # posterior walks: every replica's tree walk (a few waves, strictly sequential inside) is enqueued on its own stream,
        # so the R walks run side by side; then one D2H copy each
        if len(self._streams) < R:
            self._streams = [torch.cuda.Stream() for _ in range(R)]
        handles = []
        for r in range(R):
            with self.turn(r) as s:
                self._streams[r].wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(self._streams[r]):
                    handles.append(s.posterior_launch())
        out = []
        for r in range(R):
            s = self.solvers[r]
            s._samples = s.posterior_collect(handles[r], timer=timers[r])
            out.append(s._samples)
        prof["posterior"] += time.time() - t_ph
        return out
