"""slam.FactorGraphSolver — the incremental-update loop that drives the flow hot path
(reference: src/slam/FactorGraphSolver.py:27-550,760-933).

One incremental update = `update_physical_and_working_graphs()` + `incremental_inference()`:
re-eliminate the part of the Bayes tree touched by the new factors, then, leaves first, for every
clique without a model: simulate its training batch (children enter through their separator
factors), **train its flow** (`fit_clique_density_model`, the hot path), turn it into a separator
factor for the parent; finally sample the posterior root -> leaves with the conditional samplers.

Same method names / argument lists / per-clique call order as the reference.  Plotting,
nested-sampling local samplers and the ccolamd ordering are not rebuilt (out of scope / dead code).
The density back end plugs in through three hooks (`fit_clique_density_model`,
`root_clique_density_model_to_leaf`, `clique_density_to_separator_factor`).
"""
import json
import os
import time
from typing import Dict, List

import numpy as np

from factors.Factors import Factor, ImplicitPriorFactor
from sampler.SimulationBasedSampler import SimulationBasedSampler
from slam.BayesTree import BayesTree, BayesTreeNode
from slam.FactorGraph import FactorGraph
from slam.Variables import Variable, VariableType


class SolverArgs:
    def __init__(self, elimination_method: str = "natural", posterior_sample_num: int = 500,
                 local_sample_num: int = 500, store_clique_samples: bool = False, local_sampling_method="direct",
                 adaptive_posterior_sampling=None, *args, **kwargs):
        self.elimination_method = elimination_method
        self.posterior_sample_num = posterior_sample_num
        self.store_clique_samples = store_clique_samples
        self.local_sampling_method = local_sampling_method
        self.local_sample_num = local_sample_num
        self.adaptive_posterior_sampling = adaptive_posterior_sampling

    def jsonStr(self):
        return json.dumps(self.__dict__)


class ConditionalSampler:
    def conditional_sample_given_observation(self, conditional_dim, obs_samples=None, sample_number=None):
        """Samples of the first `conditional_dim` columns after the columns fixed by `obs_samples`
        ([n, dim]); with `sample_number` instead, unconditional samples of the first columns."""
        raise NotImplementedError("Implementation depends on density estimation method.")


class CliqueSeparatorFactor(ImplicitPriorFactor):
    """Prior over a clique's separator variables induced by its trained density model."""

    def sample(self, num_samples: int, **kwargs):
        raise NotImplementedError("implementation depends on density models")


class FactorGraphSolver:
    def __init__(self, args: SolverArgs):
        self._args = args
        self._physical_graph = FactorGraph()
        self._working_graph = FactorGraph()
        self._physical_bayes_tree = None
        self._working_bayes_tree = None
        self._implicit_factors = {}          # clique -> separator factor
        self._samples = {}                   # variable -> posterior samples
        self._new_nodes = []
        self._new_factors = []
        self._clique_samples = {}
        self._clique_true_obs = {}           # clique -> true observations appended to its flow input
        self._clique_density_model = {}      # clique -> trained model
        self._clique_variable_pattern = {}
        self._elimination_ordering = []
        self._reverse_ordering_map = {}
        self._temp_training_loss = {}

    def set_args(self, args: SolverArgs):
        raise NotImplementedError("Implementation depends on probabilistic modeling approaches.")

    # ---- read-only views ------------------------------------------------------------------------
    @property
    def elimination_method(self) -> str:
        return self._args.elimination_method

    @property
    def elimination_ordering(self) -> List[Variable]:
        return self._elimination_ordering

    @property
    def physical_vars(self) -> List[Variable]:
        return self._physical_graph.vars

    @property
    def new_vars(self) -> List[Variable]:
        return self._new_nodes

    @property
    def working_vars(self) -> List[Variable]:
        return self._working_graph.vars

    @property
    def physical_factors(self) -> List[Factor]:
        return self._physical_graph.factors

    @property
    def new_factors(self) -> List[Factor]:
        return self._new_factors

    @property
    def working_factors(self) -> List[Factor]:
        return self._working_graph.factors

    @property
    def working_factor_graph(self) -> FactorGraph:
        return self._working_graph

    @property
    def physical_factor_graph(self) -> FactorGraph:
        return self._physical_graph

    @property
    def working_bayes_tree(self) -> BayesTree:
        return self._working_bayes_tree

    @property
    def physical_bayes_tree(self) -> BayesTree:
        return self._physical_bayes_tree

    def results(self) -> Dict[Variable, np.ndarray]:
        return self._samples

    # ---- orderings ------------------------------------------------------------------------------
    def generate_natural_ordering(self) -> None:
        self._elimination_ordering = self._physical_graph.vars + self._new_nodes

    def generate_pose_first_ordering(self) -> None:
        """Order of insertion, landmarks eliminated last."""
        self._elimination_ordering = FactorGraph.generate_pose_first_ordering(
            self._physical_graph.vars + self._new_nodes)

    def generate_ordering(self) -> None:
        m = self._args.elimination_method
        if m == "natural":
            self.generate_natural_ordering()
        elif m == "pose_first":
            self.generate_pose_first_ordering()
        else:
            raise NotImplementedError("elimination_method=%r (the reference's ccolamd path is dead code)" % m)
        self._reverse_ordering_map = {v: k for k, v in enumerate(self._elimination_ordering[::-1])}

    # ---- staging --------------------------------------------------------------------------------
    def add_node(self, var: Variable = None, name: str = None, dim: int = None) -> "FactorGraphSolver":
        self._new_nodes.append(var if var else Variable(name, dim))
        return self

    def add_factor(self, factor: Factor) -> "FactorGraphSolver":
        self._new_factors.append(factor)
        return self

    # ---- graph / tree update ------------------------------------------------------------------
    def update_physical_and_working_graphs(self, timer: List[float] = None, device: str = "cpu") -> "FactorGraphSolver":
        """Fold the staged nodes / factors into the graphs and re-eliminate only the part of the Bayes tree they touch
        (what the reference does in :256-358).  Three stages:
          1. check the staged input and compute the new elimination ordering;
          2. cut the unaffected subtrees out of the physical tree (they are moved, not copied), build the working graph
             (affected variables + the separator factors of the cut subtrees + the staged input) and its Bayes tree,
             and hang the cut subtrees under the new tree;
          3. carry trained models over to cliques that reappear with the same variables in the same relative order.
        Stages 1 and 2 are a transaction: the new ordering, the graphs and the trees are committed together once the
        working tree exists; a failure before that restores the ordering, re-attaches the cut subtrees and leaves the
        staged input pending, so a retry sees the solver exactly as it was."""
        start = time.time()
        staged_nodes, staged_factors = list(self._new_nodes), list(self._new_factors)
        # -- 1. validation + ordering
        known = set(self.physical_vars) | set(staged_nodes)
        for f in staged_factors:
            missing = [str(v.name) for v in f.vars if v not in known]
            if missing:
                raise KeyError("factor %s refers to variables that are neither in the graph nor staged: %s"
                               % (f, " ".join(missing)))
        previous_ordering, previous_rmap = self._elimination_ordering, self._reverse_ordering_map
        old_tree = self._physical_bayes_tree
        try:
            self.generate_ordering()                  # raises NotImplementedError for an unknown method
            # -- 2. working graph and trees, built next to the live state
            kept_subtrees = []
            if old_tree is not None:
                touched_old = {v for f in staged_factors for v in f.vars} & set(self.physical_vars)
                affected, kept_subtrees = old_tree.get_affected_vars_and_partial_bayes_trees(vars=touched_old, detach=True)
                working = self._physical_graph.get_sub_factor_graph_with_prior(
                    variables=affected, sub_trees=kept_subtrees, clique_prior_dict=self._implicit_factors)
            else:
                working = FactorGraph()
                for v in self._working_graph.vars:
                    working.add_node(v)
                for f in self._working_graph.factors:
                    working.add_factor(f)
            for node in staged_nodes:
                working.add_node(node)
            for factor in staged_factors:
                working.add_factor(factor)
            in_working = set(working.vars)
            working_tree = working.get_bayes_tree(ordering=[v for v in self._elimination_ordering if v in in_working])
            physical_tree = working_tree.__copy__()
            physical_tree.append_child_bayes_trees(kept_subtrees)
            for node in staged_nodes:                 # (cannot fail: the working graph took the same input above)
                self._physical_graph.add_node(node)
            for factor in staged_factors:
                self._physical_graph.add_factor(factor)
        except BaseException:
            self._elimination_ordering, self._reverse_ordering_map = previous_ordering, previous_rmap
            if old_tree is not None:
                old_tree.reattach_detached()
            raise
        self._working_graph, self._working_bayes_tree, self._physical_bayes_tree = working, working_tree, physical_tree
        # -- 3. trained models of cliques that dropped out of the tree
        self._recycle_models(previous_ordering, device)
        self._new_nodes, self._new_factors = [], []
        if timer is not None:
            timer.append(time.time() - start)
        return self

    def _recycle_models(self, previous_ordering: List[Variable], device) -> None:
        """A clique that vanished from the tree may reappear with the same variables in the same relative elimination
        order but another frontal / separator split (typically last update's root, now a leaf).  Its flow is still the
        right joint density: re-wrap it (`root_clique_density_model_to_leaf`) instead of training again, emit its
        separator factor and eliminate it from the working graph.  Everything else that vanished is forgotten."""
        alive = self._physical_bayes_tree.clique_nodes
        vanished = [c for c in list(self._clique_density_model) if c not in alive]
        if not vanished:
            return
        rank_prev = {v: k for k, v in enumerate(previous_ordering)}
        rank_now = {v: k for k, v in enumerate(self._elimination_ordering)}
        candidates = {}                               # variable set -> new clique (first in tree order wins)
        for c in self._working_bayes_tree.clique_ordering():
            candidates.setdefault(frozenset(c.vars), c)
        reused = set()
        for old in vanished:
            new = candidates.get(frozenset(old.vars))
            same_order = new is not None and all(v in rank_prev for v in old.vars) and \
                sorted(old.vars, key=rank_prev.__getitem__) == sorted(new.vars, key=rank_now.__getitem__)
            if not same_order or new in reused:
                continue
            reused.add(new)
            obs = self._clique_true_obs[old]
            self._clique_true_obs[new] = obs
            for store in (self._clique_variable_pattern, self._clique_samples):
                if old in store:
                    store[new] = store[old]
            model = self.root_clique_density_model_to_leaf(old, new, device)
            self._clique_density_model[new] = model
            factor = None
            if new.separator:
                sep = sorted(new.separator, key=self._reverse_ordering_map.__getitem__)
                factor = self.clique_density_to_separator_factor(sep, model, obs)
                self._implicit_factors[new] = factor
            self._working_graph = self._working_graph.eliminate_clique_variables(clique=new, new_factor=factor)
        working = self._working_bayes_tree.clique_nodes
        for old in vanished:
            if old in working:                        # equal (same frontal + separator) to a re-used clique: the key is shared
                continue
            for store in (self._clique_density_model, self._clique_true_obs, self._clique_variable_pattern,
                          self._clique_samples):
                store.pop(old, None)

    # ---- hooks of the density back end ----------------------------------------------------------
    def fit_clique_density_model(self, clique, samples, var_ordering, timer, *args, **kwargs) -> ConditionalSampler:
        raise NotImplementedError("Implementation depends on probabilistic modeling.")

    def root_clique_density_model_to_leaf(self, old_clique: BayesTreeNode, new_clique: BayesTreeNode,
                                          device) -> ConditionalSampler:
        raise NotImplementedError("Implementation depends on probabilistic modeling")

    def clique_density_to_separator_factor(self, separator_var_list: List[Variable], density_model,
                                           true_obs: np.ndarray) -> CliqueSeparatorFactor:
        raise NotImplementedError("Implementation depends on probabilistic modeling")

    def adaptive_posterior(self, timer: List[float] = None, *args, **kwargs):
        raise NotImplementedError("implementation depends on density models.")

    # ---- inference ------------------------------------------------------------------------------
    def incremental_inference(self, timer: List[float] = None, clique_dim_timer: List[List[float]] = None, *args,
                              **kwargs):
        self.fit_tree_density_models(timer=timer, clique_dim_timer=clique_dim_timer, *args, **kwargs)
        if self._args.adaptive_posterior_sampling is None:
            self._samples = self.sample_posterior(timer=timer, *args, **kwargs)
        else:
            self._samples = self.adaptive_posterior(timer=timer, *args, **kwargs)
        return self._samples

    def fit_tree_density_models(self, timer: List[float] = None, clique_dim_timer: List[List[float]] = None, *args,
                                **kwargs):
        """Leaves first: local sampling and flow training on every clique of the working tree that
        has no model yet (reference :409-477)."""
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
            model = self.fit_clique_density_model(clique=clique, samples=local_samples,
                                                  var_ordering=sample_var_ordering, timer=timer)
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

    def clique_training_sampler(self, clique: BayesTreeNode, num_samples: int, method: str):
        """-> (training samples [n, D], variable ordering incl. observation variables, true observations)."""
        graph = self._working_graph.get_clique_factor_graph(clique)
        variable_pattern = self._working_bayes_tree.clique_variable_pattern(clique)
        if method == "direct":
            sampler = SimulationBasedSampler(factors=graph.factors, vars=variable_pattern)
            backend = self._simulation_backend()
            if backend is not None:
                from sampler.DeviceSimulation import DeviceSimulationUnsupported
                try:
                    return sampler.sample(num_samples, backend=backend)
                except DeviceSimulationUnsupported:
                    pass                       # a factor type without a device sampler: simulate this clique on the host
            return sampler.sample(num_samples)
        raise ValueError("Unknown sampling method (nested-sampling local samplers are not rebuilt).")

    def _simulation_backend(self):
        """Draw backend of the clique training-batch simulator: None = the factors' host (numpy) samplers.
        Density back ends that can consume device batches override this."""
        return None

    def sample_posterior(self, timer: List[float] = None, *args, **kwargs) -> Dict[Variable, np.ndarray]:
        """Root -> leaves: every clique samples its frontal variables conditioned on its true
        observations and the already-sampled separator variables (reference :497-550)."""
        num_samples = self._args.posterior_sample_num
        start = time.time()
        stack = [self._physical_bayes_tree.root]
        samples = {}
        while stack:
            clique = stack.pop()
            frontal_list = sorted(clique.frontal, key=lambda x: self._reverse_ordering_map[x])
            separator_list = sorted(clique.separator, key=lambda x: self._reverse_ordering_map[x])
            model = self._clique_density_model[clique]
            obs = self._clique_true_obs[clique]
            given = [np.tile(obs, (num_samples, 1))] if len(obs) != 0 else []
            given += [samples[v] for v in separator_list]
            if given:
                frontal_samples = model.conditional_sample_given_observation(
                    conditional_dim=clique.frontal_dim, obs_samples=np.hstack(given))
            else:
                frontal_samples = model.conditional_sample_given_observation(
                    conditional_dim=clique.frontal_dim, sample_number=num_samples)
            col = 0
            for v in frontal_list:
                samples[v] = frontal_samples[:, col:col + v.dim]
                col += v.dim
            stack.extend(clique.children)
        if timer is not None:
            timer.append(time.time() - start)
        return samples


def run_incrementally(case_dir: str, solver: FactorGraphSolver, nodes_factors_by_step, truth=None, traj_plot=False,
                      plot_args=None, check_root_transform=False) -> str:
    """Feed the solver step by step and write the reference's per-step result files
    (reference :760-933; figures are not produced).  Returns the run directory."""
    run_count = 1
    while os.path.exists(f"{case_dir}/run{run_count}"):
        run_count += 1
    run_dir = f"{case_dir}/run{run_count}"
    os.makedirs(run_dir)
    with open(f"{run_dir}/parameters", "w+") as f:
        f.write(solver._args.jsonStr())
    step_timer, step_list, posterior_sampling_timer, fitting_timer = [], [], [], []
    mixtures = []
    for i, (step_nodes, step_factors) in enumerate(nodes_factors_by_step):
        for node in step_nodes:
            solver.add_node(node)
        for factor in step_factors:
            solver.add_factor(factor)
            if hasattr(factor, "posterior_weights"):
                mixtures.append(factor)
        step_list.append(i)
        prefix = f"{run_dir}/step{i}"
        detailed_timer, clique_dim_timer = [], []
        start = time.time()
        solver.update_physical_and_working_graphs(timer=detailed_timer)
        cur_sample = solver.incremental_inference(timer=detailed_timer, clique_dim_timer=clique_dim_timer)
        step_timer.append(time.time() - start)
        with open(f"{prefix}_ordering", "w+") as f:
            f.write(" ".join([str(v.name) for v in solver.elimination_ordering]))
        with open(f"{prefix}_split_timing", "w+") as f:
            f.write(" ".join([str(t) for t in detailed_timer]))
        with open(f"{prefix}_step_training_loss", "w+") as f:
            f.write(json.dumps(solver._temp_training_loss))
        posterior_sampling_timer.append(detailed_timer[-1])
        fitting_timer.append(sum(detailed_timer[1:-1]))
        np.savetxt(fname=prefix, X=np.hstack([cur_sample[v] for v in solver.elimination_ordering]))
        np.savetxt(fname=prefix + "_dim_time", X=np.array(clique_dim_timer))
        if mixtures:   # posterior weights of the data-association hypotheses (reference :913-933)
            with open(prefix + ".hypoweights", "w+") as f:
                for factor in mixtures:
                    w = factor.posterior_weights(cur_sample)
                    f.write(" ".join(str(v.name) for v in factor.vars) + " : " + ",".join(str(x) for x in w) + "\n")
        for name, vals in (("step_timing", step_timer), ("step_list", step_list),
                           ("posterior_sampling_timer", posterior_sampling_timer), ("fitting_timer", fitting_timer)):
            with open(f"{run_dir}/{name}", "w+") as f:
                f.write(" ".join(str(t) for t in vals))
    return run_dir
