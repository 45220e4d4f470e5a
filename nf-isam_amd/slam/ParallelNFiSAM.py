"""slam.ParallelNFiSAM — NF-iSAM with the cliques of an incremental update sharded over the GPUs of one node.

One process per GPU (`torch.distributed`; backend "nccl" = RCCL over xGMI on a multi-GPU node, "gloo" where several
ranks share one GPU, i.e. in the tests).  Every rank runs the same host bookkeeping on the same factor graph (ordering,
Bayes tree, clique re-use: deterministic, a few ms per update) and therefore agrees on the working tree; what is
sharded is the hot path, `fit_clique_density_model` (reference loop: src/slam/FactorGraphSolver.py:409-477):

  upward pass   cliques are dealt to ranks by whole subtrees (`CliqueParallel.assign_subtrees`, sibling subtrees are
                independent).  All ranks walk the same leaves-first order; a rank trains only its own
                cliques.  On a tree edge whose two ends live on different ranks the child's rank draws
                `local_sample_num` samples of the child's separator factor (`FlowsPriorFactor.sample_on_device`, the
                child -> parent message of NF-iSAM, src/slam/NFiSAM.py:271-288) and sends the [n, Ds] fp32 batch
                point-to-point (<= 2000 x 11 x 4 B = 88 KB); the parent's rank consumes it as the prior of those
                variables in its own clique simulation.  No collective on the data path; a rank waits only for what it
                actually consumes, in the reference's order (FactorGraphSolver.py:436-470).  Messages are matched by
                ORDER, not by tag (RCCL has no tags): every rank derives the same list of cross-rank edges and issues its
                sends and receives per peer in that order (`CliqueParallel.EdgeExchange`); both ends know [n, Ds], so
                there is no header; device tensors with "nccl", host tensors with gloo.
  replication   after the pass every rank packs the models it trained (parameter blob + normalisation constants + true
                observations + loss record, ~25 KB per clique) into ONE flat fp32 tensor; one all_gather of the sizes and
                one of the padded tensors later all ranks hold the same `_clique_density_model` / `_implicit_factors`
                for the following updates (SURVEY.md §8e "preferred": smaller than samples and re-usable).
  downward pass `sample_posterior` (FactorGraphSolver.py:497-550).  Default: every rank holds every model, so the
                single-launch tree walk runs replicated under a seed shared from rank 0 (identical samples everywhere,
                nothing to send: the whole walk is ~10 ms).  `posterior="sharded"` follows the reference's order with
                the cliques split by subtree: a clique whose parent was sampled on another rank receives the samples of
                its separator variables [n_post, Ds] point-to-point (FactorGraphSolver.py:524-531), and the per-variable
                samples are gathered at the end.
Chain-shaped trees (`pose_first`, all shipped large runs) have no sibling subtrees: every clique lands on rank 0 and the
other ranks idle through the upward pass ("replicas only") — use replicas / independent problems there.
"""
import time
from typing import Dict, List

import numpy as np
import torch
import torch.distributed as dist

import nfisam_hip as _nh
from flows.flows import NSF_AR
from flows.prior_dist import CustomMultivariateNormal
from slam.CliqueParallel import CliqueTree, EdgeExchange, assign_subtrees
from slam.FactorGraphSolver import CliqueSeparatorFactor
from slam.NFiSAM import FlowsPriorFactor, NFiSAM, NFiSAMArgs, NormalizingFlowModelWithSeparator, _device


class RemoteSeparatorSamples(CliqueSeparatorFactor):
    """Stand-in, on the parent's rank, for the separator factor of a clique trained on another rank during this update:
    its `sample` is the point-to-point receive of the batch the owner drew (edge `key` of the pass's `EdgeExchange`)."""

    def __init__(self, vars: List, key, exchange: EdgeExchange, device):
        super().__init__()
        self._vars, self._key, self._exchange, self._device = vars, key, exchange, device
        self._batch = None

    @property
    def vars(self) -> List:
        return self._vars

    @property
    def is_gaussian(self) -> bool:
        return False

    def _receive(self, n):
        if self._batch is None:
            self._batch = self._exchange.recv(self._key).to(self._device)
            self._exchange = None                      # the batch outlives the pass, the exchange does not
        if self._batch.shape[0] != n:
            raise ValueError("remote separator batch has %d samples, %d requested" % (self._batch.shape[0], n))
        return self._batch

    def sample_on_device(self, num_samples: int) -> "torch.Tensor":
        return self._receive(num_samples)

    def sample(self, num_samples: int, **kwargs) -> np.ndarray:
        return self._receive(num_samples).cpu().numpy().astype(np.float64)


class _PendingSeparator(CliqueSeparatorFactor):
    """Placeholder on a rank that neither trains a clique nor consumes its message in this pass: it keeps the working
    graph's elimination going (the factor's variables are what matters there) until the replicated model replaces it."""

    def __init__(self, vars: List):
        super().__init__()
        self._vars = vars

    @property
    def vars(self) -> List:
        return self._vars

    @property
    def is_gaussian(self) -> bool:
        return False

    def sample(self, num_samples: int, **kwargs):
        raise RuntimeError("separator factor of a clique trained on another rank: not available during this pass")


class ParallelNFiSAM(NFiSAM):
    def __init__(self, args: NFiSAMArgs = None, posterior: str = "replicated"):
        super().__init__(args)
        if not dist.is_initialized():
            raise RuntimeError("ParallelNFiSAM needs torch.distributed to be initialised (one process per GPU)")
        if posterior not in ("replicated", "sharded"):
            raise ValueError("posterior must be 'replicated' or 'sharded'")
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self._posterior_mode = posterior
        self._on_device = dist.get_backend() == "nccl"       # RCCL moves device tensors; gloo needs host tensors
        self.owner_log: List[Dict] = []                      # per update: {clique name: rank} (tests, reports)
        # per update: what the rank-to-rank traffic of the pass was and what it cost THIS rank (bench.py's `exchange` block):
        # cross_rank_edges / bytes are properties of the pass (identical on every rank), the *_ms are this rank's wall clock
        self.exchange_stats: List[Dict] = []
        self._gather_s = 0.0

    # ---- collectives of the replication steps ----------------------------------------------------------------
    def _comm_device(self):
        return torch.device(_device()) if self._on_device else torch.device("cpu")

    def _all_gather_flat(self, mine: "torch.Tensor") -> List["torch.Tensor"]:
        """Every rank's flat fp32 tensor on every rank: one all_gather of the lengths, one of the padded payloads."""
        t0 = time.perf_counter()
        dev = self._comm_device()
        mine = mine.to(device=dev, dtype=torch.float32).contiguous().view(-1)
        size = torch.tensor([mine.numel()], dtype=torch.int64, device=dev)
        sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(self.world)]
        dist.all_gather(sizes, size)
        sizes = [int(t.item()) for t in sizes]
        width = max(max(sizes), 1)
        padded = torch.zeros(width, dtype=torch.float32, device=dev)
        padded[:mine.numel()] = mine
        parts = [torch.empty(width, dtype=torch.float32, device=dev) for _ in range(self.world)]
        dist.all_gather(parts, padded)
        if self._on_device:
            torch.cuda.synchronize()                       # (RCCL collectives are asynchronous: the clock wants them done)
        self._gather_s += time.perf_counter() - t0
        self._gather_bytes = getattr(self, "_gather_bytes", 0) + 4 * width * self.world + 8 * self.world
        return [t[:k] for t, k in zip(parts, sizes)]

    def _shared_seed(self) -> int:
        t = torch.tensor([int(np.random.randint(0, 2 ** 31 - 1))], dtype=torch.int64, device=self._comm_device())
        dist.broadcast(t, src=0)
        return int(t.item())

    # ---- who trains what -------------------------------------------------------------------------------------
    @staticmethod
    def _clique_name(clique) -> str:
        return "".join(sorted(str(v.name) for v in clique.frontal)) + "|" + "".join(sorted(str(v.name) for v in clique.separator))

    def _assign(self, cliques, needs_work) -> Dict:
        """clique -> rank for the cliques of a tree (list in parents-before-children order).  Cost = dim^2 x n for the
        cliques that need work, 0 for the others, so that subtrees are balanced by the work actually done."""
        ids = {id(c): k for k, c in enumerate(cliques)}
        parent = {ids[id(c)]: (ids[id(c.parent)] if c.parent is not None and id(c.parent) in ids else None) for c in cliques}
        cost = {ids[id(c)]: (float(c.dim) ** 2 if needs_work(c) else 0.0) + 1e-6 for c in cliques}
        a = assign_subtrees(CliqueTree(parent, cost), self.world)
        return {id(c): a[ids[id(c)]] for c in cliques}

    # ---- upward pass ---------------------------------------------------------------------------------------
    def _separator_dim(self, clique) -> int:
        return int(sum(v.dim for v in clique.separator))

    def fit_tree_density_models(self, timer: List[float] = None, clique_dim_timer: List[List[float]] = None, *args,
                                **kwargs):
        self._temp_training_loss = {}
        cliques = self._working_bayes_tree.clique_ordering()
        retrain = [c for c in cliques if c not in self._clique_density_model]
        owner = self._assign(cliques, lambda c: c not in self._clique_density_model)
        self.owner_log.append({self._clique_name(c): owner[id(c)] for c in retrain})
        # child -> parent messages that cross ranks, in the order the pass visits the children (leaves first).  Only a
        # parent that is itself trained in this update consumes its children's separator samples.
        key_of = {id(c): k for k, c in enumerate(cliques)}
        n_loc = self._args.local_sample_num
        edges = [(key_of[id(c)], owner[id(c)], owner[id(c.parent)], (n_loc, self._separator_dim(c)))
                 for c in reversed(cliques)
                 if c not in self._clique_density_model and c.separator and c.parent is not None
                 and c.parent not in self._clique_density_model and owner[id(c.parent)] != owner[id(c)]]
        exchange = EdgeExchange(edges, self.rank, device=_device(), on_device=self._on_device)
        crossing = {e[0] for e in edges}
        self.exchange_log = exchange.log
        self._gather_s, self._gather_bytes = 0.0, 0
        trained = []
        t_begin = time.time()
        for clique in reversed(cliques):                          # leaves first, as the reference pops its ordering
            if clique in self._clique_density_model:
                if clique_dim_timer is not None:
                    clique_dim_timer.append([clique.dim, time.time() - t_begin])
                continue
            separator_list = sorted(clique.separator, key=lambda x: self._reverse_ordering_map[x])
            mine = owner[id(clique)] == self.rank
            key = key_of[id(clique)]
            new_separator_factor = None
            if mine:
                t0 = time.time()
                local_samples, sample_var_ordering, true_obs = self.clique_training_sampler(
                    clique, num_samples=self._args.local_sample_num, method=self._args.local_sampling_method)
                if timer is not None:
                    timer.append(time.time() - t0)
                self._clique_true_obs[clique] = true_obs
                model = self.fit_clique_density_model(clique=clique, samples=local_samples,
                                                      var_ordering=sample_var_ordering, timer=timer)
                self._clique_density_model[clique] = model
                if separator_list:
                    new_separator_factor = self.clique_density_to_separator_factor(separator_list, model, true_obs)
                    if key in crossing:                               # the child -> parent message crosses ranks
                        exchange.send(key, new_separator_factor.sample_on_device(n_loc))
            elif separator_list:
                if key in crossing and owner[id(clique.parent)] == self.rank:
                    new_separator_factor = RemoteSeparatorSamples(separator_list, key, exchange, _device())
                else:       # nobody on this rank draws from it during this pass; the replicated model replaces it below
                    new_separator_factor = _PendingSeparator(separator_list)
            if new_separator_factor is not None:
                self._implicit_factors[clique] = new_separator_factor
            self._working_graph = self._working_graph.eliminate_clique_variables(clique=clique,
                                                                                 new_factor=new_separator_factor)
            trained.append(clique)
            if clique_dim_timer is not None:
                clique_dim_timer.append([clique.dim, time.time() - t_begin])
        exchange.drain()
        # ---- replication: every rank ends the update with every model (one flat tensor per rank) ---------------------
        mine = [c for c in trained if owner[id(c)] == self.rank]
        blobs = self._all_gather_flat(self._pack_models(mine))
        for r, blob in enumerate(blobs):
            if r != self.rank:
                self._install_models([c for c in trained if owner[id(c)] == r], blob.cpu().numpy())
        st = exchange.stats
        self.exchange_stats.append(dict(
            cliques_trained=len(retrain), cliques_trained_here=len(mine), cross_rank_edges=len(edges),
            bytes=int(sum(4 * e[3][0] * e[3][1] for e in edges)),          # separator batches [n, Ds] fp32 that crossed ranks
            p2p_messages_here=st["sends"] + st["recvs"], p2p_bytes_here=st["bytes_sent"] + st["bytes_received"],
            p2p_send_ms=1e3 * st["send_s"], p2p_wait_ms=1e3 * st["wait_s"], p2p_ms=1e3 * (st["send_s"] + st["wait_s"]),
            all_gather_ms=1e3 * self._gather_s, all_gather_bytes=int(self._gather_bytes),
            upward_pass_ms=1e3 * (time.time() - t_begin)))

    # A rank's models of one update as one flat fp32 vector.  Per clique (in the pass's order):
    #   [D, n_obs, n_loss, L]  kparams[L * Pk]  mean[D]  std[D]  circular[D]  true_obs as float64 bits (2 words each)  loss[n_loss]
    # Everything else (K, H, B, frontal / separator split) every rank derives from the clique and the solver arguments,
    # which are identical everywhere.
    def _pack_models(self, cliques) -> "torch.Tensor":
        parts = []
        for clique in cliques:
            m = self._clique_density_model[clique]
            f0 = m.flows[0]
            name = "".join(str(v.name) for v in clique.vars)
            loss = np.asarray(self._temp_training_loss.get(name, []), dtype=np.float32)
            obs = np.ascontiguousarray(np.asarray(self._clique_true_obs[clique], dtype=np.float64).reshape(-1))
            mean = m.samples_mean.cpu().numpy() if torch.is_tensor(m.samples_mean) else np.asarray(m.samples_mean)
            std = m.samples_std.cpu().numpy() if torch.is_tensor(m.samples_std) else np.asarray(m.samples_std)
            parts += [np.array([f0.dim, obs.size, loss.size, len(m.flows)], dtype=np.float32),
                      m.kernel_params().detach().cpu().numpy().astype(np.float32).reshape(-1),
                      mean.astype(np.float32).reshape(-1), std.astype(np.float32).reshape(-1),
                      np.asarray(m.circular_dim_list, dtype=np.float32).reshape(-1), obs.view(np.float32), loss]
        flat = np.concatenate(parts) if parts else np.zeros(0, dtype=np.float32)
        return torch.from_numpy(flat)

    def _install_models(self, cliques, flat: np.ndarray):
        device = _device()
        K, H, B = self._args.num_knots, self._args.hidden_dim, 5.0
        off = 0
        for clique in cliques:
            D, n_obs, n_loss, L = (int(v) for v in flat[off:off + 4])
            off += 4
            Pk = _nh.kparam_count(D, K, H)
            kp = torch.from_numpy(flat[off:off + L * Pk].copy()).to(device); off += L * Pk
            mean = flat[off:off + D].copy(); off += D
            std = flat[off:off + D].copy(); off += D
            circular = [bool(v) for v in flat[off:off + D]]; off += D
            true_obs = flat[off:off + 2 * n_obs].copy().view(np.float64); off += 2 * n_obs
            loss = [float(v) for v in flat[off:off + n_loss]]; off += n_loss
            flows = [NSF_AR.from_kernel_params(D, K, B, H, kp[l * Pk:(l + 1) * Pk]) for l in range(L)]
            sep_dim = D - clique.frontal_dim
            model = NormalizingFlowModelWithSeparator(
                flows, CustomMultivariateNormal(dim=D, device=device),
                CustomMultivariateNormal(dim=sep_dim, device=device) if sep_dim > 0 else None, circular,
                torch.from_numpy(mean), torch.from_numpy(std))
            self._clique_density_model[clique] = model
            self._clique_true_obs[clique] = true_obs
            if n_loss:
                self._temp_training_loss["".join(str(v.name) for v in clique.vars)] = loss
            if clique.separator:
                separator_list = sorted(clique.separator, key=lambda x: self._reverse_ordering_map[x])
                self._implicit_factors[clique] = self.clique_density_to_separator_factor(separator_list, model, true_obs)
        if off != flat.size:
            raise RuntimeError("model payload of %d words, %d consumed" % (flat.size, off))

    # ---- downward pass -------------------------------------------------------------------------------------
    def sample_posterior(self, timer: List = None, *args, **kwargs):
        if self._posterior_mode == "sharded":
            return self.sample_posterior_sharded(timer=timer)
        torch.manual_seed(self._shared_seed())
        return super().sample_posterior(timer=timer, *args, **kwargs)

    def sample_posterior_sharded(self, timer: List = None) -> Dict:
        """The reference's root -> leaves loop (FactorGraphSolver.py:497-550) with the cliques split by subtree: the
        frontal variables of a clique are sampled on its rank, conditioned on its true observations and on the samples
        of its separator variables; those arrive point-to-point when the parent was sampled elsewhere (parent -> child
        edges in the pass's order through `EdgeExchange`), and one all_gather of the per-rank sample blocks ends it."""
        start = time.time()
        n = self._args.posterior_sample_num
        cliques = self._physical_bayes_tree.clique_ordering()          # parents before children
        owner = self._assign(cliques, lambda c: True)
        key_of = {id(c): k for k, c in enumerate(cliques)}
        rmap = self._reverse_ordering_map
        # a parent sends right after it was sampled, to its children in their list order: that is the global edge order
        edges = [(key_of[id(ch)], owner[id(c)], owner[id(ch)], (n, self._separator_dim(ch)))
                 for c in cliques for ch in c.children if ch.separator and owner[id(ch)] != owner[id(c)]]
        exchange = EdgeExchange(edges, self.rank, device=_device(), on_device=self._on_device)
        crossing = {e[0] for e in edges}
        self.posterior_exchange_log = exchange.log                  # (tests: the pairs' operation sequences must mirror each other)
        samples: Dict = {}
        for clique in cliques:
            frontal_list = sorted(clique.frontal, key=rmap.__getitem__)
            separator_list = sorted(clique.separator, key=rmap.__getitem__)
            if owner[id(clique)] != self.rank:
                continue
            if key_of[id(clique)] in crossing:
                batch = exchange.recv(key_of[id(clique)]).cpu().numpy()
                col = 0
                for v in separator_list:
                    samples[v] = batch[:, col:col + v.dim]
                    col += v.dim
            model = self._clique_density_model[clique]
            obs = self._clique_true_obs[clique]
            given = [np.tile(obs, (n, 1))] if len(obs) != 0 else []
            given += [samples[v] for v in separator_list]
            if given:
                fs = model.conditional_sample_given_observation(conditional_dim=clique.frontal_dim,
                                                                obs_samples=np.hstack(given))
            else:
                fs = model.conditional_sample_given_observation(conditional_dim=clique.frontal_dim, sample_number=n)
            col = 0
            for v in frontal_list:
                samples[v] = fs[:, col:col + v.dim]
                col += v.dim
            for child in clique.children:                      # parent -> child messages that cross ranks
                if key_of[id(child)] in crossing:
                    sl = sorted(child.separator, key=rmap.__getitem__)
                    exchange.send(key_of[id(child)], torch.from_numpy(np.hstack([samples[v] for v in sl]).astype(np.float32)))
        exchange.drain()
        self._gather_s, self._gather_bytes = 0.0, 0
        # every rank ends with the samples of every variable (what `results()` hands out): one block per rank
        mine = [c for c in cliques if owner[id(c)] == self.rank]
        cols = [samples[v] for c in mine for v in sorted(c.frontal, key=rmap.__getitem__)]
        block = np.hstack(cols).astype(np.float32) if cols else np.zeros((n, 0), dtype=np.float32)
        for r, flat in enumerate(self._all_gather_flat(torch.from_numpy(np.ascontiguousarray(block)))):
            if r == self.rank:
                continue
            theirs = [v for c in cliques if owner[id(c)] == r for v in sorted(c.frontal, key=rmap.__getitem__)]
            blk = flat.cpu().numpy().reshape(n, -1)
            col = 0
            for v in theirs:
                samples[v] = blk[:, col:col + v.dim]
                col += v.dim
        st = exchange.stats
        self.posterior_exchange_stats = dict(
            cross_rank_edges=len(edges), bytes=int(sum(4 * e[3][0] * e[3][1] for e in edges)),
            p2p_send_ms=1e3 * st["send_s"], p2p_wait_ms=1e3 * st["wait_s"], p2p_ms=1e3 * (st["send_s"] + st["wait_s"]),
            all_gather_ms=1e3 * self._gather_s, all_gather_bytes=int(self._gather_bytes), downward_pass_ms=1e3 * (time.time() - start))
        if timer is not None:
            timer.append(time.time() - start)
        return {v: samples[v] for v in self._elimination_ordering}
