"""slam.ReplicaNFiSAM — R independent NF-iSAM runs advanced in lock-step on ONE GPU.

The shipped large examples use the `pose_first` ordering, whose working tree per update is a chain of ~5 cliques
(SURVEY.md §0.4): a parent needs its child's samples, so the cliques of one run are trained strictly one after the other
and a single clique (n = 2000, D = 15: 945 waves) cannot fill 256 CUs — the launch is latency-bound.  The reference
itself loops over eight dataset variants one after the other (example/slam/plaza_dataset/run_nfisam.py:11-21).
Independent runs (seeds, noise / data-association variants, datasets) have no such dependency: here R solvers share
ONE batched training plan (grid.y = replica): each one runs its own host bookkeeping and clique simulation up to its next
`fit_clique_density_model`, its clique trains in the replica's slot of the plan while the other replicas' cliques train
next to it or their host steps run (`_update_in_slots`; in lock-step batches with `NFISAM_REPLICA_SLOTS=0`).  That moves
the training from the latency regime (one clique per launch) to the throughput regime without changing
any replica's result: every replica owns its random streams (numpy, python,
torch host + device generator states are swapped in and out around its turns), its batch, parameters, Adam state and
early-stop decision, so replica r reproduces the sequential run with seed r — bit for bit when both use the same
kernel family (`NFISAM_TRAIN`), to kernel rounding (loss 5e-4) otherwise.
Replicas may hold different graphs: a replica that has fewer cliques to train in an update simply drops out of the
later batches of that update.
"""
import contextlib
import os
import random
import threading
import time
from typing import Dict, List, Optional

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
            model.posterior_static()      # device constants of the tree walk: now, while other replicas train, not in the serial tail
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

    # ---- training without lock-step: one slot per replica in ONE batched plan ----------------------------------------
    def _update_in_slots(self, timers, prof, steps=None, on_update=None) -> list:
        """One incremental update of all replicas (`steps` None) -- or a whole sequence of them, every replica at its own pace
        -- with their cliques trained in SLOTS of one batched training plan.
        -> `steps` None: the handles of the replicas' posterior walks (launched as each replica finishes its upward pass);
           else: per replica the list of posterior samples per step.

        In lock-step (`_update_in_lock_step`) a batch trains until its SLOWEST clique stops -- 1263 iterations on Plaza1 where
        the average clique needs 681 -- and the host work of all replicas sits between two batches with the GPU idle.  Here
        the plan of R same-shaped cliques runs as a conveyor of chunks (`TrainBatch.begin / feed / peek`: the library's
        feeder thread keeps two chunks enqueued ahead): replica r owns slot r; when the mirror shows its clique stopped, its
        trained parameters are taken out, the replica does its host step (wrap the model, eliminate, simulate and prepare
        its next clique) while the other slots keep training, and the new clique goes into the slot between two chunks
        (`TrainBatch.refill`).  Graph updates at the start and posterior walks at the end are interleaved the same way.
        Every clique runs exactly its own number of iterations and is trained by the same kernels on the same launch shape
        as in the batch, so the results are bit-identical to lock-step (and to the sequential runs).  Cliques of another
        shape than the plan's wait for the next idle moment and are trained together, blocking (`train_odd`).
        With `steps` (`run_incrementally`) a replica that has its posterior goes straight on to its next update: the
        replicas drift apart and nobody waits for the slowest one of an update."""
        R = len(self.solvers)
        depth = max(1, int(os.environ.get("NFISAM_SLOT_DEPTH", "2")))
        odd: List[tuple] = []                            # (replica, prepared fit) of another shape than the plan's
        free_running = steps is not None
        n_steps = len(steps) if free_running else 1
        done, fits = [False] * R, [0] * R                # done: upward pass of the current step finished (posterior launched)
        finished = [False] * R                           # all steps of the replica finished
        step_idx, t_step = [0] * R, [0.0] * R
        results: List[list] = [[] for _ in range(R)]
        owner: List[Optional[dict]] = [None] * R         # the prepared fit in slot r
        loaded_at, refill_seq = [0.0] * R, [0] * R
        gens: List[object] = [None] * R
        handles: List[object] = [None] * R
        pending: List[list] = [[] for _ in range(R)]     # (free running) posterior walks in flight: (step, handle, step start)
        want_next = [False] * R                          # (free running) the replica's next step is to be staged
        state = {"trainer": None, "key": None}
        if len(self._streams) < R:
            self._streams = [torch.cuda.Stream() for _ in range(R)]
        if free_running and len(self.__dict__.setdefault("_copy_streams", [])) < R:
            self._copy_streams = [torch.cuda.Stream() for _ in range(R)]
        self.fit_iterations = self.__dict__.get("fit_iterations") or [0] * R

        def shape_key(prep):
            return (prep["n"], prep["D"], prep["cfg"], str(prep["device"]))

        def graph_step(r):
            t0 = time.time()
            with self.turn(r) as s:
                if free_running:
                    vs, fs = steps[step_idx[r]]
                    for v in vs:
                        s.add_node(v)
                    for f in fs:
                        s.add_factor(f)
                s.update_physical_and_working_graphs(timer=timers[r])
                gens[r] = s.fit_tree_steps(timer=timers[r])
            prof["graphs"] += time.time() - t0

        def host_step(r):
            """replica r: finish its previous clique, simulate / prepare the next one -> prep or None (upward pass done)"""
            t0 = time.time()
            with self.turn(r):
                try:
                    prep = next(gens[r])
                except StopIteration:
                    prep = None
            prof["simulate+prepare"] += time.time() - t0
            return prep

        def open_plan(prep):
            state["key"] = shape_key(prep)
            state["trainer"] = self._slot_plan(prep, R)
            state["trainer"].begin()
            state["trainer"].feed(depth)

        def place(r, prep):
            while prep is not None:
                if state["trainer"] is None:
                    open_plan(prep)
                if shape_key(prep) == state["key"]:
                    tb = state["trainer"]
                    tb.refill(r, prep["training_data"], prep["kp0"])
                    # (read AFTER the refill is enqueued: every chunk launched from now on is behind it)
                    owner[r], refill_seq[r], loaded_at[r] = prep, tb.enqueued(), time.time()
                    return
                odd.append((r, prep))                      # another shape: waits for the next idle moment (see train_odd)
                return
            done[r] = True
            t0 = time.time()                               # upward pass finished: this replica's posterior walk, on its own stream
            with self.turn(r) as s:
                self._streams[r].wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(self._streams[r]):
                    handles[r] = s.posterior_launch()
            prof["posterior"] += time.time() - t0
            if not free_running:
                finished[r] = True
            else:
                # the walk is in flight: the replica goes straight on to its next step (the samples are taken out when the walk's
                # event has fired; the random draws of the walk are made, in the replica's stream, before anything of the next step)
                pending[r].append((step_idx[r], handles[r], t_step[r]))
                handles[r] = None
                step_idx[r] += 1
                want_next[r] = step_idx[r] < n_steps

        odd_job: List[object] = []                       # at most one: (thread, batch, error holder, start time)

        def train_odd():
            """Cliques of another shape than the plan's, collected since the last idle moment: ONE batched (ragged) training
            of all of them -- runs whose cliques keep changing shape (Manhattan) end up in lock-step batches this way, runs
            with one dominant shape (Plaza) train the odd clique every few dozen updates on its own.  A batch that holds a
            minority of the replicas is trained by a worker thread (the call releases the interpreter lock while it waits
            for the device; its plan has its own stream), so the slots keep being served and `finish_odd` picks the results
            up; a batch that holds most of them blocks (NFISAM_ODD_THREAD=0 / 1: always block / always the thread)."""
            batch, odd[:] = list(odd), []
            err: List[BaseException] = []
            mode = os.environ.get("NFISAM_ODD_THREAD", "auto")
            busy = sum(1 for o in owner if o is not None)
            if mode == "0" or (mode != "1" and (2 * len(batch) >= sum(1 for f in finished if not f) or busy < len(batch))):
                # most replicas wait in this batch (runs whose cliques keep changing shape): blocking, so that the others
                # pile up behind it and the next batch holds all of them again -- small staggered batches each pay the full
                # latency-bound training time (Manhattan-136, eight replicas: 4.0-4.2 s either way, nothing to win there)
                odd_job.append((None, batch, err, time.time()))
                self.solvers[0].train_prepared([p for _, p in batch])
                prof["train"] += time.time() - odd_job[0][3]
                finish_odd()
                return

            def work():
                try:
                    self.solvers[0].train_prepared([p for _, p in batch])
                except BaseException as e:              # re-raised by finish_odd
                    err.append(e)
            th = threading.Thread(target=work, daemon=True)
            odd_job.append((th, batch, err, time.time()))
            th.start()

        def finish_odd():
            """-> True if a finished odd batch was picked up (its replicas go on)"""
            if not odd_job or (odd_job[0][0] is not None and odd_job[0][0].is_alive()):
                return False
            th, batch, err, t0 = odd_job.pop()
            if th is not None:
                th.join()
            if err:
                raise err[0]
            dt = time.time() - t0
            for r, p in batch:
                timers[r].append(dt / len(batch))
                fits[r] += 1
                self.fit_iterations[r] += int(p["iters"])
            for r, _ in batch:
                place(r, host_step(r))
            return True

        def start_step(r):
            done[r], t_step[r] = False, time.time()
            graph_step(r)
            place(r, host_step(r))

        def harvest_ready():
            """Replicas whose clique has stopped: take the result out, do the host step, refill.  -> any progress"""
            tb = state["trainer"]
            busy = [r for r in range(R) if owner[r] is not None]
            if tb is None or not busy:
                return False
            seq, states = tb.peek()
            if seq < 0:
                return False
            ready = [r for r in busy if seq > refill_seq[r] and (states[r][1] != 0 or states[r][0] >= tb.cfg.max_iters)]
            for r in ready:
                prep, owner[r] = owner[r], None
                step, _, err = states[r]
                if err:                                    # non-finite loss: the batch path's handling (one retry from fresh parameters),
                    self.solvers[r].train_prepared([prep])  # on the replica's OWN solver: `last_fit_retried` and the plan cache are its
                else:
                    prep["trained"], prep["iters"], prep["iter_loss"] = tb.kparams[r].clone(), step, tb.iter_loss[r].clone()
                self.fit_iterations[r] += int(prep["iters"])
                timers[r].append((time.time() - loaded_at[r]) / max(1, len(busy)))
                fits[r] += 1
                place(r, host_step(r))
            return bool(ready)

        def collect_ready():
            """(free running) stage the next step of replicas that have launched their posterior; take out the samples of walks
            that have finished (in step order), call `on_update`.  -> any progress"""
            progress = False
            for r in range(R):
                if want_next[r]:
                    want_next[r] = False
                    start_step(r)
                    progress = True
                while pending[r] and pending[r][0][1]["done"].query():
                    k, handle, t_start = pending[r].pop(0)
                    t0 = time.time()
                    s = self.solvers[r]
                    s._samples = s.posterior_collect(handle, timer=timers[r], copy_stream=self._copy_streams[r])
                    prof["posterior"] += time.time() - t0
                    results[r].append(s._samples)
                    if on_update is not None:
                        on_update(r, k, s._samples, time.time() - t_start)
                    progress = True
                if not finished[r] and step_idx[r] >= n_steps and done[r] and not pending[r] and not want_next[r]:
                    finished[r] = True
                    progress = True
            return progress

        # the shape of the plan: the one of the previous update if there is one (then the replicas start training one by one
        # while the others' graphs are still being updated), else the shape most replicas start with
        def drive():
            last = self.__dict__.get("_slot_last")
            if last is not None or free_running:
                if last is not None:
                    state["trainer"], state["key"] = last
                    state["trainer"].begin()
                    state["trainer"].feed(depth)
                for r in range(R):
                    start_step(r)
                    harvest_ready()
            else:
                for r in range(R):
                    graph_step(r)
                first = [host_step(r) for r in range(R)]
                shapes = [shape_key(p) for p in first if p is not None]
                if shapes:
                    key = max(set(shapes), key=shapes.count)
                    open_plan(next(p for p in first if p is not None and shape_key(p) == key))
                for r in range(R):
                    place(r, first[r])
            t_progress = time.time()
            while not all(finished):
                moved = harvest_ready()
                moved = finish_odd() or moved
                if free_running:
                    moved = collect_ready() or moved
                if moved:
                    t_progress = time.time()
                elif odd and not odd_job:
                    train_odd()
                    t_progress = time.time()
                else:
                    t0 = time.time()
                    time.sleep(2e-5)
                    prof["train"] += time.time() - t0
                    if t0 - t_progress > float(os.environ.get("NFISAM_SLOT_WATCHDOG_S", "60")):
                        tb = state["trainer"]             # a wedged conveyor must not hang the caller for ever
                        raise RuntimeError("replica slots: no clique finished for %.0f s (chunks enqueued %d, mirror %r, waiting %r)" % (
                            t0 - t_progress, tb.enqueued(), tb.peek(), [(r, refill_seq[r]) for r in range(R) if owner[r] is not None]))
            if state["trainer"] is not None:
                state["trainer"].end()
                self._slot_last = (state["trainer"], state["key"])
                prof["chunks"] = prof.get("chunks", 0) + state["trainer"].enqueued()
            self.last_batches = [sum(1 for f in fits if f > k) for k in range(max(fits) if fits else 0)]
            return results if free_running else handles

        try:
            return drive()
        except BaseException:
            if odd_job and odd_job[0][0] is not None:
                odd_job[0][0].join()
            if state["trainer"] is not None:              # no feeder left launching chunks behind an exception; the plan is
                try:                                       # dropped (its slots may hold half-trained cliques)
                    state["trainer"].end()
                    torch.cuda.synchronize()
                finally:
                    self.__dict__.pop("_slot_last", None)
                    for tb in self.__dict__.pop("_slot_plans", {}).values():
                        tb.close()
            raise

    def run_incrementally(self, steps, on_update=None, timers: List[List[float]] = None) -> List[list]:
        """All replicas through the same sequence of incremental steps [(new variables, new factors), ..] (reference:
        `run_incrementally`, FactorGraphSolver.py:760-933, one solver; its drivers loop over dataset variants one after the
        other, example/slam/plaza_dataset/run_nfisam.py:11-21), every replica at its OWN pace: a replica that has the
        posterior of step k stages step k + 1 at once, while the other replicas' cliques keep training in their slots of the
        shared plan -- no replica waits for the slowest one of an update.  -> per replica the posterior samples of every
        step; `on_update(replica, step, samples, seconds)` is called as results arrive."""
        R = len(self.solvers)
        timers = timers if timers is not None else [[] for _ in range(R)]
        prof = self.__dict__.setdefault("profile", {"graphs": 0.0, "simulate+prepare": 0.0, "train": 0.0, "posterior": 0.0})
        return self._update_in_slots(timers, prof, steps=list(steps), on_update=on_update)

    def _slot_plan(self, prep, R):
        """The R-slot training plan for cliques shaped like `prep` (kept for the following updates)."""
        from nfisam_hip import TrainBatch
        a = self.solvers[0]._args
        K, H, B, L = prep["cfg"]
        key = (prep["n"], prep["D"], prep["cfg"], str(prep["device"]), R)
        plans = self.__dict__.setdefault("_slot_plans", {})
        tb = plans.get(key)
        if tb is None:
            while len(plans) >= 2:                          # (a run's cliques have one shape, rarely two)
                plans.pop(next(iter(plans))).close()
            dev = prep["device"]
            tb = TrainBatch([torch.zeros(prep["n"], prep["D"], dtype=torch.float32, device=dev) for _ in range(R)],
                            [torch.zeros_like(prep["kp0"]) for _ in range(R)], K, H, B, L, lr=a.learning_rate,
                            max_iters=a.flow_iterations, average_window=a.average_window, loss_delta_tol=a.loss_delta_tol,
                            early_stop=True)
            # empty slots must look finished: a zeroed state would train the zero problem
            tb.states[:, 1] = 1
            plans[key] = tb
        return tb

    def _update_in_lock_step(self, timers, prof) -> list:
        """All replicas step together; the R pending cliques are trained by ONE batched launch sequence per round
        (`NFISAM_REPLICA_SLOTS=0`; what `_update_in_slots` is measured against)."""
        R = len(self.solvers)
        gens = []
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
        prof["posterior"] += time.time() - t_ph
        return handles

    # ---- one incremental update of every replica -----------------------------------------------------------------
    def update(self, timers: List[List[float]] = None) -> List[Dict]:
        """`update_physical_and_working_graphs` + `incremental_inference` of all replicas
        (reference per replica: FactorGraphSolver.py:803-808).  -> posterior samples per replica."""
        R = len(self.solvers)
        timers = timers if timers is not None else [[] for _ in range(R)]
        prof = self.__dict__.setdefault("profile", {"graphs": 0.0, "simulate+prepare": 0.0, "train": 0.0, "posterior": 0.0})
        if R >= 2 and os.environ.get("NFISAM_REPLICA_SLOTS", "1") != "0":
            handles = self._update_in_slots(timers, prof)
        else:
            handles = self._update_in_lock_step(timers, prof)
        t_ph = time.time()
        out = []
        for r in range(R):
            s = self.solvers[r]
            s._samples = s.posterior_collect(handles[r], timer=timers[r])
            out.append(s._samples)
        prof["posterior"] += time.time() - t_ph
        return out
