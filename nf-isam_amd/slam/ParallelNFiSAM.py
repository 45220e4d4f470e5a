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
                actually consumes, in the reference's order (FactorGraphSolver.py:436-470).
  replication   after the pass every trained clique's model (parameter blob + normalisation constants + true
                observations, ~25 KB) is broadcast from its owner, so that all ranks hold the same
                `_clique_density_model` / `_implicit_factors` for the following updates (SURVEY.md §8e "preferred":
                smaller than samples and re-usable).
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
from slam.CliqueParallel import CliqueTree, assign_subtrees
from slam.FactorGraphSolver import CliqueSeparatorFactor
from slam.NFiSAM import FlowsPriorFactor, NFiSAM, NFiSAMArgs, NormalizingFlowModelWithSeparator, _device


class RemoteSeparatorSamples(CliqueSeparatorFactor):
    """Stand-in, on the parent's rank, for the separator factor of a clique trained on another rank during this update:
    its `sample` is the point-to-point receive of the batch the owner drew."""

    def __init__(self, vars: List, owner: int, tag: int, solver: "ParallelNFiSAM"):
        super().__init__()
        self._vars, self._owner, self._tag, self._solver = vars, owner, tag, solver
        self._batch = None

    @property
    def vars(self) -> List:
        return self._vars

    @property
    def is_gaussian(self) -> bool:
        return False

    def _receive(self, n):
        if self._batch is None:
            self._batch = self._solver._recv_batch(self._owner, self._tag)
        if self._batch.shape[0] != n:
            raise ValueError("remote separator batch has %d samples, %d requested" % (self._batch.shape[0], n))
        return self._batch

    def sample_on_device(self, num_samples: int) -> "torch.Tensor":
        return self._receive(num_samples)

    def sample(self, num_samples: int, **kwargs) -> np.ndarray:
        return self._receive(num_samples).cpu().numpy().astype(np.float64)


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
        self._pending_sends = []

    # ---- point-to-point batches -----------------------------------------------------------------------------
    def _send_batch(self, t: "torch.Tensor", dst: int, tag: int):
        """Non-blocking: the sender goes on with its next clique; the requests (and the buffers they read) are kept
        until `_drain_sends` at the end of the pass."""
        t = t.to(torch.float32).contiguous()
        if not self._on_device:
            t = t.cpu()
        hdr = torch.tensor(list(t.shape), dtype=torch.int64, device=t.device)
        self._pending_sends.append((dist.isend(hdr, dst=dst, tag=tag), hdr))
        self._pending_sends.append((dist.isend(t, dst=dst, tag=tag), t))

    def _drain_sends(self):
        for req, _ in self._pending_sends:
            req.wait()
        self._pending_sends = []

    def _recv_batch(self, src: int, tag: int) -> "torch.Tensor":
        dev = torch.device(_device()) if self._on_device else torch.device("cpu")
        hdr = torch.empty(2, dtype=torch.int64, device=dev)
        dist.recv(hdr, src=src, tag=tag)
        out = torch.empty(int(hdr[0]), int(hdr[1]), dtype=torch.float32, device=dev)
        dist.recv(out, src=src, tag=tag)
        return out.to(_device())

    def _broadcast_object(self, obj, src: int):
        box = [obj if self.rank == src else None]
        dist.broadcast_object_list(box, src=src, device=torch.device(_device()) if self._on_device else None)
        return box[0]

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
    def fit_tree_density_models(self, timer: List[float] = None, clique_dim_timer: List[List[float]] = None, *args,
                                **kwargs):
        self._temp_training_loss = {}
        cliques = self._working_bayes_tree.clique_ordering()
        owner = self._assign(cliques, lambda c: c not in self._clique_density_model)
        tag_of = {id(c): 100 + k for k, c in enumerate(cliques)}
        self.owner_log.append({self._clique_name(c): owner[id(c)] for c in cliques if c not in self._clique_density_model})
        trained = []
        t_begin = time.time()
        for clique in reversed(cliques):                          # leaves first, as the reference pops its ordering
            if clique in self._clique_density_model:
                if clique_dim_timer is not None:
                    clique_dim_timer.append([clique.dim, time.time() - t_begin])
                continue
            separator_list = sorted(clique.separator, key=lambda x: self._reverse_ordering_map[x])
            mine = owner[id(clique)] == self.rank
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
                    p = clique.parent
                    if p is not None and owner[id(p)] != self.rank:     # the child -> parent message crosses ranks
                        self._send_batch(new_separator_factor.sample_on_device(self._args.local_sample_num),
                                         owner[id(p)], tag_of[id(clique)])
            elif separator_list:
                new_separator_factor = RemoteSeparatorSamples(separator_list, owner[id(clique)], tag_of[id(clique)], self)
            if new_separator_factor is not None:
                self._implicit_factors[clique] = new_separator_factor
            self._working_graph = self._working_graph.eliminate_clique_variables(clique=clique,
                                                                                 new_factor=new_separator_factor)
            trained.append(clique)
            if clique_dim_timer is not None:
                clique_dim_timer.append([clique.dim, time.time() - t_begin])
        self._drain_sends()
        # ---- replication: every rank ends the update with every model ------------------------------------------------
        for clique in trained:
            src = owner[id(clique)]
            payload = self._model_payload(clique) if src == self.rank else None
            payload = self._broadcast_object(payload, src)
            if src != self.rank:
                self._install_payload(clique, payload)

    def _model_payload(self, clique) -> Dict:
        m = self._clique_density_model[clique]
        f0 = m.flows[0]
        name = "".join(str(v.name) for v in clique.vars)
        return dict(kparams=m.kernel_params().cpu().numpy(), mean=np.asarray(m.samples_mean.cpu() if torch.is_tensor(m.samples_mean) else m.samples_mean, dtype=np.float32),
                    std=np.asarray(m.samples_std.cpu() if torch.is_tensor(m.samples_std) else m.samples_std, dtype=np.float32),
                    circular=list(m.circular_dim_list), true_obs=np.asarray(self._clique_true_obs[clique], dtype=np.float64),
                    dim=int(f0.dim), K=int(f0.K), H=int(f0.hidden_dim), B=float(f0.B), L=len(m.flows),
                    loss_name=name, loss=self._temp_training_loss.get(name))

    def _install_payload(self, clique, p: Dict):
        device = _device()
        D, K, H, B, L = p["dim"], p["K"], p["H"], p["B"], p["L"]
        kp = torch.from_numpy(p["kparams"]).to(device)
        Pk = _nh.kparam_count(D, K, H)
        flows = [NSF_AR.from_kernel_params(D, K, B, H, kp[l * Pk:(l + 1) * Pk]) for l in range(L)]
        sep_dim = D - clique.frontal_dim
        model = NormalizingFlowModelWithSeparator(
            flows, CustomMultivariateNormal(dim=D, device=device),
            CustomMultivariateNormal(dim=sep_dim, device=device) if sep_dim > 0 else None, p["circular"],
            torch.from_numpy(p["mean"]), torch.from_numpy(p["std"]))
        self._clique_density_model[clique] = model
        self._clique_true_obs[clique] = p["true_obs"]
        if p["loss"] is not None:
            self._temp_training_loss[p["loss_name"]] = p["loss"]
        if clique.separator:
            separator_list = sorted(clique.separator, key=lambda x: self._reverse_ordering_map[x])
            self._implicit_factors[clique] = self.clique_density_to_separator_factor(separator_list, model, p["true_obs"])

    # ---- downward pass -------------------------------------------------------------------------------------
    def sample_posterior(self, timer: List = None, *args, **kwargs):
        if self._posterior_mode == "sharded":
            return self.sample_posterior_sharded(timer=timer)
        seed = self._broadcast_object(int(np.random.randint(0, 2 ** 31 - 1)) if self.rank == 0 else None, 0)
        torch.manual_seed(seed)
        return super().sample_posterior(timer=timer, *args, **kwargs)

    def sample_posterior_sharded(self, timer: List = None) -> Dict:
        """The reference's root -> leaves loop (FactorGraphSolver.py:497-550) with the cliques split by subtree: the
        frontal variables of a clique are sampled on its rank, conditioned on its true observations and on the samples
        of its separator variables; those arrive point-to-point when the parent was sampled elsewhere."""
        start = time.time()
        n = self._args.posterior_sample_num
        cliques = self._physical_bayes_tree.clique_ordering()          # parents before children
        owner = self._assign(cliques, lambda c: True)
        tag_of = {id(c): 20000 + k for k, c in enumerate(cliques)}
        rmap = self._reverse_ordering_map
        samples: Dict = {}
        for clique in cliques:
            frontal_list = sorted(clique.frontal, key=rmap.__getitem__)
            separator_list = sorted(clique.separator, key=rmap.__getitem__)
            mine = owner[id(clique)] == self.rank
            p = clique.parent
            if mine and separator_list and p is not None and owner[id(p)] != self.rank:
                batch = self._recv_batch(owner[id(p)], tag_of[id(clique)]).cpu().numpy()
                col = 0
                for v in separator_list:
                    samples[v] = batch[:, col:col + v.dim]
                    col += v.dim
            if mine:
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
                    if owner[id(child)] != self.rank and child.separator:
                        sl = sorted(child.separator, key=rmap.__getitem__)
                        self._send_batch(torch.from_numpy(np.hstack([samples[v] for v in sl]).astype(np.float32)),
                                         owner[id(child)], tag_of[id(child)])
        self._drain_sends()
        # every rank ends with the samples of every variable (what `results()` hands out)
        for clique in cliques:
            src = owner[id(clique)]
            frontal_list = sorted(clique.frontal, key=rmap.__getitem__)
            block = np.hstack([samples[v] for v in frontal_list]).astype(np.float32) if src == self.rank else None
            block = self._broadcast_object(block, src)
            col = 0
            for v in frontal_list:
                samples[v] = block[:, col:col + v.dim]
                col += v.dim
        if timer is not None:
            timer.append(time.time() - start)
        return {v: samples[v] for v in self._elimination_ordering}
