"""CPU tests of the host bookkeeping around the hot path: `.fg` parsing,
incremental grouping, symbolic elimination -> Bayes tree, affected-subtree extraction, the clique
training-batch simulator's column layout and the solver loop's per-clique call order (with a stand-in
density back end: the real one needs a GPU)."""
import os

import numpy as np
import pytest

from factors.Factors import Factor, SE2R2RangeGaussianLikelihoodFactor, SE2RelativeGaussianLikelihoodFactor, \
    UnarySE2ApproximateGaussianPriorFactor
from geometry.TwoDimension import SE2Pose, se2_compose, se2_exp, se2_inverse, se2_log
from sampler.SimulationBasedSampler import SimulationBasedSampler
from slam.BayesTree import BayesTree, BayesTreeNode
from slam.FactorGraph import FactorGraph
from slam.FactorGraphSimulator import factor_graph_to_string, read_factor_graph_from_file
from slam.FactorGraphSolver import CliqueSeparatorFactor, ConditionalSampler, FactorGraphSolver, SolverArgs
from slam.RunBatch import group_nodes_factors_incrementally
from slam.Variables import R2Variable, SE2Variable, VariableType
from utils.Statistics import MMDb

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def small_graph(tmp_path_factory):
    g = np.load(os.path.join(GOLDEN, "small_range_case1.npz"))
    p = tmp_path_factory.mktemp("fg") / "factor_graph.fg"
    p.write_text(str(g["factor_graph_fg"]))
    return read_factor_graph_from_file(str(p)), str(p)


def test_se2_algebra():
    rng = np.random.RandomState(0)
    v = rng.randn(50, 3) * np.array([2.0, 2.0, 1.0])
    v[0] = [1.0, 2.0, 0.0]                               # zero rotation branch
    p = se2_exp(v)
    np.testing.assert_allclose(se2_log(p), v, atol=1e-10)
    ident = se2_compose(p, se2_inverse(p))
    np.testing.assert_allclose(ident, 0, atol=1e-10)
    a, b = SE2Pose(1, 2, 0.5), SE2Pose(-3, 0.5, 2.9)
    np.testing.assert_allclose((a * b).matrix, a.matrix @ b.matrix, atol=1e-12)
    np.testing.assert_allclose((a / b).matrix, a.matrix @ np.linalg.inv(b.matrix), atol=1e-12)
    np.testing.assert_allclose(SE2Pose.by_exp_map(a.log_map()).array, a.array, atol=1e-12)


def test_fg_roundtrip_and_grouping(small_graph):
    (nodes, truth, factors), path = small_graph
    assert [v.name for v in nodes] == ["X0", "X1", "X2", "X3", "X4", "X5", "L1", "L2"]
    assert len(factors) == 14 and isinstance(factors[0], UnarySE2ApproximateGaussianPriorFactor)
    np.testing.assert_allclose(truth[nodes[2]], [30.0, 30.0, 0.0])
    text = factor_graph_to_string(nodes, factors, truth)
    again = [Factor.construct_from_text(l, nodes) for l in text.splitlines() if l.startswith("Factor")]
    assert [str(f) for f in again] == [str(f) for f in factors]
    steps = group_nodes_factors_incrementally(nodes, factors, incremental_step=1)
    assert [[v.name for v in s[0]] for s in steps] == [["X0", "L1", "L2"], ["X1"], ["X2"], ["X3"], ["X4"], ["X5"]]
    assert [len(s[1]) for s in steps] == [3, 2, 2, 2, 2, 3]
    assert sum(len(s[1]) for s in steps) == len(factors)
    two = group_nodes_factors_incrementally(nodes, factors, incremental_step=2)
    assert [[v.name for v in s[0]] for s in two] == [["X0", "L1", "L2", "X1"], ["X2", "X3"], ["X4", "X5"]]
    with pytest.raises(NotImplementedError):      # a factor type outside the rebuilt scope is refused loudly
        Factor.construct_from_text("Factor SE2BearingLikelihoodFactor X0 L1 1 1", nodes)


def test_factor_sampling_statistics():
    np.random.seed(0)
    X0, X1, L = SE2Variable("X0"), SE2Variable("X1"), R2Variable("L1", VariableType.Landmark)
    cov = np.diag([0.04, 0.0016, 0.0004])
    prior = UnarySE2ApproximateGaussianPriorFactor(X0, SE2Pose(1.0, 2.0, 1.5), cov)
    s0 = prior.sample(20000)
    eps = se2_log(se2_compose(se2_inverse(np.array([1.0, 2.0, 1.5])), s0))
    np.testing.assert_allclose(np.cov(eps.T), cov, atol=3e-3)
    odo = SE2RelativeGaussianLikelihoodFactor(X0, X1, SE2Pose(30, 0, -1.2), covariance=cov)
    s1 = odo.sample(var1=s0, var2=None)
    rel = se2_compose(se2_inverse(s0), s1)
    np.testing.assert_allclose(np.median(rel, 0), [30, 0, -1.2], atol=0.02)
    back = odo.sample(var1=None, var2=s1)          # inverse direction is consistent in distribution
    np.testing.assert_allclose(back.mean(0)[:2], s0.mean(0)[:2], atol=0.05)
    obs = odo.sample(var1=s0, var2=s1)
    assert obs.shape == (20000, 3) and odo.observation_var.circular_dim_list == [False, False, True]
    rng_f = SE2R2RangeGaussianLikelihoodFactor(X0, L, 42.4, sigma=2.0)
    lm = rng_f.sample(var1=s0, var2=None)
    d = np.linalg.norm(lm - s0[:, :2], axis=1)
    assert abs(d.mean() - 42.4) < 0.1 and abs(d.std() - 2.0) < 0.1
    ang = np.arctan2((lm - s0[:, :2])[:, 1], (lm - s0[:, :2])[:, 0])
    assert abs(np.mean(np.cos(ang))) < 0.03               # uniform bearing: ring
    o = rng_f.sample(var1=s0, var2=lm)
    assert o.shape == (20000, 1) and abs((o[:, 0] - d).std() - 2.0) < 0.1


def test_bayes_tree_from_chain_and_affected_subtrees(small_graph):
    (nodes, truth, factors), _ = small_graph
    g = FactorGraph()
    for v in nodes:
        g.add_node(v)
    for f in factors:
        g.add_factor(f)
    order = FactorGraph.generate_pose_first_ordering(nodes)
    assert [v.name for v in order] == ["X0", "X1", "X2", "X3", "X4", "X5", "L1", "L2"]
    tree = g.get_bayes_tree(ordering=order)
    cliques = tree.clique_ordering()
    names = lambda vs: sorted(v.name for v in vs)   # noqa: E731
    # pose_first gives a chain: root {L1,L2,X4,X5}, then X3 | L1 L2 X4, ...
    assert names(cliques[0].frontal) == ["L1", "L2", "X4", "X5"] and not cliques[0].separator
    for k, c in enumerate(cliques[1:]):
        assert names(c.frontal) == ["X%d" % (3 - k)] and names(c.separator) == ["L1", "L2", "X%d" % (4 - k)]
        assert len(c.children) <= 1
    pat = tree.clique_variable_pattern(cliques[1])
    assert [v.name for v in pat] == ["L2", "L1", "X4", "X3"]          # [separator | frontal], reverse elimination order
    # running intersection + every variable is frontal exactly once
    assert sorted(v.name for c in cliques for v in c.frontal) == sorted(v.name for v in nodes)
    cp = tree.__copy__()
    assert cp.clique_nodes == tree.clique_nodes and cp.root is not tree.root
    X1 = nodes[1]
    affected, subs = tree.get_affected_vars_and_partial_bayes_trees({X1})
    assert names(affected) == ["L1", "L2", "X1", "X2", "X3", "X4", "X5"]
    assert len(subs) == 1 and names(subs[0].root.frontal) == ["X0"]
    # hash / equality semantics: same frontal + separator => same clique key
    a = BayesTreeNode(frontal={nodes[0]}, separator={nodes[6]})
    b = BayesTreeNode(frontal={nodes[0]}, separator={nodes[6]})
    assert a == b and hash(a) == hash(b) and a != BayesTreeNode(frontal={nodes[0]})


def test_natural_ordering_gives_a_branching_tree():
    """x0 - x1 - x2 with two landmarks seen from the ends: sibling subtrees exist (sharding unit)."""
    X = [SE2Variable("X%d" % i) for i in range(3)]
    L = [R2Variable("L%d" % i, VariableType.Landmark) for i in range(2)]
    cov = np.eye(3) * 0.01
    g = FactorGraph()
    for v in X + L:
        g.add_node(v)
    g.add_factor(UnarySE2ApproximateGaussianPriorFactor(X[0], SE2Pose(), cov))
    g.add_factor(SE2RelativeGaussianLikelihoodFactor(X[0], X[1], SE2Pose(1, 0, 0), covariance=cov))
    g.add_factor(SE2RelativeGaussianLikelihoodFactor(X[1], X[2], SE2Pose(1, 0, 0), covariance=cov))
    g.add_factor(SE2R2RangeGaussianLikelihoodFactor(X[0], L[0], 1.0, 0.1))
    g.add_factor(SE2R2RangeGaussianLikelihoodFactor(X[2], L[1], 1.0, 0.1))
    tree = g.get_bayes_tree(ordering=[L[0], L[1], X[0], X[2], X[1]])
    root = tree.root
    assert len(tree.clique_ordering()) >= 3 and any(len(c.children) >= 2 for c in tree.clique_ordering()) or \
        len(root.children) >= 1


def test_simulation_sampler_layout(small_graph):
    (nodes, truth, factors), _ = small_graph
    np.random.seed(1)
    name = {v.name: v for v in nodes}
    # the last clique of the real run: X4, X5, L1, L2 with two loop-closing range factors at X5
    prior = UnarySE2ApproximateGaussianPriorFactor(name["X4"], SE2Pose(90, 30, 0), np.eye(3) * 1e-2)
    fs = [prior] + [f for f in factors if set(v.name for v in f.vars) <= {"X4", "X5", "L1", "L2"} and len(f.vars) == 2]
    pattern = [name["L2"], name["L1"], name["X5"], name["X4"]]
    samples, order, obs = SimulationBasedSampler(fs, pattern).sample(500)
    assert [v.name for v in order][-4:] == ["L2", "L1", "X5", "X4"]
    n_obs = len(order) - 4
    assert samples.shape == (500, n_obs + 2 + 2 + 3 + 3) and obs.shape == (n_obs,)
    assert n_obs >= 1 and all(str(v.name).startswith("O") for v in order[:n_obs])
    assert set(np.round(obs, 3)) <= {round(float(f.observation[0]), 3) for f in fs[1:] if len(f.observation) == 1}


class _StubModel(ConditionalSampler):
    def __init__(self, mean):
        self.mean = mean

    def conditional_sample_given_observation(self, conditional_dim, obs_samples=None, sample_number=None):
        n = sample_number if obs_samples is None else obs_samples.shape[0]
        o = 0 if obs_samples is None else obs_samples.shape[1]
        return self.mean[o:o + conditional_dim] + 0.01 * np.random.randn(n, conditional_dim)


class _StubFactor(CliqueSeparatorFactor):
    def __init__(self, vars, model, obs):
        self._v, self.m, self.obs = vars, model, obs

    @property
    def vars(self):
        return self._v

    def sample(self, n, **kw):
        d = sum(v.dim for v in self._v)
        if len(self.obs):
            return self.m.conditional_sample_given_observation(d, obs_samples=np.tile(self.obs, (n, 1)))
        return self.m.conditional_sample_given_observation(d, sample_number=n)


class _StubSolver(FactorGraphSolver):
    def __init__(self, args):
        super().__init__(args)
        self.log = []

    def fit_clique_density_model(self, clique, samples, var_ordering, timer, *a, **k):
        self.log.append(("fit", sorted(v.name for v in clique.frontal), samples.shape[1]))
        return _StubModel(samples.mean(0))

    def root_clique_density_model_to_leaf(self, old, new, device):
        self.log.append(("reuse", sorted(v.name for v in new.frontal)))
        return self._clique_density_model[old]

    def clique_density_to_separator_factor(self, sep, model, obs):
        return _StubFactor(sep, model, obs)


def test_solver_loop_structure_matches_reference_run(small_graph):
    """D = 7 -> 11 -> ... -> 12 and one trained clique per update with root-model reuse, as measured
    on the reference (SURVEY.md §8 'Sizes at the BASELINE configs', §0.4)."""
    (nodes, truth, factors), _ = small_graph
    np.random.seed(0)
    s = _StubSolver(SolverArgs(elimination_method="pose_first", local_sample_num=64, posterior_sample_num=16))
    dims = []
    for vs, fs in group_nodes_factors_incrementally(nodes, factors, 1):
        for v in vs:
            s.add_node(v)
        for f in fs:
            s.add_factor(f)
        s.update_physical_and_working_graphs()
        res = s.incremental_inference()
        assert set(res) == set(s.physical_vars) and all(x.shape[0] == 16 for x in res.values())
        dims.append([e[2] for e in s.log if e[0] == "fit"][-1])
    assert dims == [7, 11, 11, 11, 11, 12]
    assert sum(1 for e in s.log if e[0] == "fit") == 6
    assert [e[1] for e in s.log if e[0] == "reuse"] == [["X0"], ["X1"], ["X2"], ["X3"]]
    assert [v.name for v in s.elimination_ordering] == ["X0", "X1", "X2", "X3", "X4", "X5", "L1", "L2"]
    for v in nodes[:6]:   # stub models return batch means: poses land near the truth
        assert np.linalg.norm(res[v].mean(0)[:2] - truth[v][:2]) < 3.0


def test_failed_update_leaves_the_solver_untouched(small_graph):
    """A staged factor on an unknown variable, or an unsupported ordering method, must fail BEFORE the physical tree is
    cut (the unaffected subtrees are moved out of the live tree, so a late failure would corrupt a retry)."""
    (nodes, truth, factors), _ = small_graph
    np.random.seed(0)
    s = _StubSolver(SolverArgs(elimination_method="pose_first", local_sample_num=64, posterior_sample_num=16))
    steps = group_nodes_factors_incrementally(nodes, factors, 1)
    for vs, fs in steps[:3]:
        for v in vs:
            s.add_node(v)
        for f in fs:
            s.add_factor(f)
        s.update_physical_and_working_graphs()
        s.incremental_inference()
    before = str(s.physical_bayes_tree)
    n_cliques = len(s.physical_bayes_tree.clique_ordering())
    vs, fs = steps[3]
    for f in fs:                              # the new pose is NOT staged: its factors dangle
        s.add_factor(f)
    with pytest.raises(KeyError):
        s.update_physical_and_working_graphs()
    assert str(s.physical_bayes_tree) == before and len(s.physical_bayes_tree.clique_ordering()) == n_cliques
    assert len(s.new_factors) == len(fs) and len(s.physical_factors) == sum(len(q[1]) for q in steps[:3])
    s._args.elimination_method = "ccolamd"    # dead code in the reference: refused, nothing cut
    for v in vs:
        s.add_node(v)
    with pytest.raises(NotImplementedError):
        s.update_physical_and_working_graphs()
    assert str(s.physical_bayes_tree) == before
    s._args.elimination_method = "pose_first"
    s.update_physical_and_working_graphs()    # the retry works on the intact state
    res = s.incremental_inference()
    assert set(res) == set(s.physical_vars) and len(s.physical_vars) == 2 + 4


def test_late_failure_of_an_update_restores_ordering_and_tree(small_graph, monkeypatch):
    """a failure AFTER the ordering was recomputed and the unaffected subtrees were cut (here: building the
    working graph's Bayes tree) must put back the previous ordering and re-attach the subtrees, so that a retry computes
    `previous_ordering` and the recycled models from the right state."""
    from slam.FactorGraph import FactorGraph
    (nodes, truth, factors), _ = small_graph
    np.random.seed(0)
    s = _StubSolver(SolverArgs(elimination_method="pose_first", local_sample_num=64, posterior_sample_num=16))
    steps = group_nodes_factors_incrementally(nodes, factors, 1)
    for vs, fs in steps[:4]:
        for v in vs:
            s.add_node(v)
        for f in fs:
            s.add_factor(f)
        s.update_physical_and_working_graphs()
        s.incremental_inference()
    before_tree = str(s.physical_bayes_tree)
    before_order = [v.name for v in s.elimination_ordering]
    before_rmap = dict(s._reverse_ordering_map)
    parents = {id(c): (id(c.parent) if c.parent is not None else None) for c in s.physical_bayes_tree.clique_ordering()}
    n_factors = len(s.physical_factors)
    vs, fs = steps[4]
    for v in vs:
        s.add_node(v)
    for f in fs:
        s.add_factor(f)
    real = FactorGraph.get_bayes_tree

    def boom(self, *a, **k):
        raise RuntimeError("injected: elimination failed")
    monkeypatch.setattr(FactorGraph, "get_bayes_tree", boom)
    with pytest.raises(RuntimeError):
        s.update_physical_and_working_graphs()
    monkeypatch.setattr(FactorGraph, "get_bayes_tree", real)
    assert [v.name for v in s.elimination_ordering] == before_order and s._reverse_ordering_map == before_rmap
    assert str(s.physical_bayes_tree) == before_tree
    assert {id(c): (id(c.parent) if c.parent is not None else None) for c in s.physical_bayes_tree.clique_ordering()} == parents
    assert len(s.physical_factors) == n_factors and len(s.new_factors) == len(fs) and len(s.new_vars) == len(vs)
    s.update_physical_and_working_graphs()            # the retry works on the restored state
    res = s.incremental_inference()
    assert set(res) == set(s.physical_vars)
    assert [e[1] for e in s.log if e[0] == "reuse"] == [["X0"], ["X1"], ["X2"]]      # as in the undisturbed run


def test_mmd_metric():
    rng = np.random.RandomState(0)
    a, b = rng.randn(500, 2), rng.randn(500, 2)
    assert MMDb(a, b) < 0.1 and MMDb(a, b + 3.0) > 0.5
    assert abs(MMDb(a, a)) < 1e-7


def test_reference_nf_vs_nested_sampling_discrepancy_is_what_the_survey_measured():
    """Calibration of the posterior parity tolerance: MMDb(reference NF-iSAM run, nested sampling)."""
    g = np.load(os.path.join(GOLDEN, "small_range_case1.npz"))
    vals = []
    for i in range(4):
        o_run, o_dyn = str(g["run1_step%d_ordering" % i]).split(), str(g["dyn1_step%d_ordering" % i]).split()
        cols = lambda order, arr: np.hstack([arr[:, _off(order, v):_off(order, v) + 2] for v in sorted(order)])  # noqa: E731
        vals.append(MMDb(cols(o_run, g["run1_step%d" % i]), cols(o_dyn, g["dyn1_step%d" % i])))
    assert all(0.01 < v < 0.35 for v in vals), vals


def _off(order, name):
    off = 0
    for v in order:
        if v == name:
            return off
        off += 3 if v.startswith("X") else 2
    raise KeyError(name)


def test_ambiguous_data_association_factor():
    from factors.Factors import AmbiguousDataAssociationFactor
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    nodes, truth, factors = read_factor_graph_from_file(os.path.join(ROOT, "tests", "data", "Plaza1ADA0.4EFG",
                                                                      "factor_graph.fg"))
    ada = [f for f in factors if isinstance(f, AmbiguousDataAssociationFactor)]
    assert len(nodes) == 782 and len(factors) == 1584 and len(ada) == 279
    f = ada[0]
    assert str(Factor.construct_from_text(str(f), nodes)) == str(f)
    assert f.root_var.name == "X4" and [v.name for v in f.child_vars] == ["L1", "L0", "L2", "L3"]
    np.testing.assert_allclose(f.weights, 0.25)
    steps = group_nodes_factors_incrementally(nodes, factors, incremental_step=5)
    assert len(steps) == 156 and sum(len(s[1]) for s in steps) == len(factors)
    # simulated measurements: a multinomial split of the batch over the four hypotheses
    np.random.seed(0)
    name = {v.name: v for v in nodes}
    centers = {"L0": (0.0, 0.0), "L1": (10.0, 0.0), "L2": (20.0, 0.0), "L3": (30.0, 0.0)}
    n = 4000
    drawn = {name[k]: np.tile(np.array(c), (n, 1)) for k, c in centers.items()}
    drawn[name["X4"]] = np.tile(np.array([0.0, 0.0, 0.3]), (n, 1))
    o = f.sample_observations(drawn)
    assert o.shape == (n, 1)
    for d in (0.0, 10.0, 20.0, 30.0):          # a quarter of the simulated ranges sits near each candidate
        assert abs(np.mean(np.abs(o[:, 0] - d) < 3 * f.components[0].sigma) - 0.25) < 0.03
    # hypothesis weights: put the pose where only L2 explains the measured range
    r = float(f.observation[0])
    drawn[name["X4"]] = np.tile(np.array([20.0 - r, 0.0, 0.0]), (n, 1))
    w = f.posterior_weights(drawn)
    assert abs(w.sum() - 1) < 1e-9 and np.argmax(w) == 2 and w[2] > 0.9
    # the clique sampler turns a fully-sampled ADA factor into an observation column
    X3, X4 = name["X3"], name["X4"]
    odo = [g for g in factors if isinstance(g, SE2RelativeGaussianLikelihoodFactor) and g.var2 == X4][0]
    prior = UnarySE2ApproximateGaussianPriorFactor(X3, SE2Pose(*truth[X3]), np.eye(3) * 1e-4)

    class _LmPrior(UnarySE2ApproximateGaussianPriorFactor.__mro__[1]):   # ExplicitPriorFactor
        def __init__(self, v, c):
            self._v, self._c = v, c

        @property
        def vars(self):
            return [self._v]

        def sample(self, n, **kw):
            return np.tile(np.array(self._c), (n, 1)) + 0.1 * np.random.randn(n, 2)
    fs = [prior, odo, f] + [_LmPrior(name[k], truth[name[k]]) for k in centers]
    pattern = [name[k] for k in ("L3", "L2", "L1", "L0")] + [X4, X3]
    samples, order, obs = SimulationBasedSampler(fs, pattern).sample(300)
    assert samples.shape == (300, 1 + 8 + 6) and [v.name for v in order][0].startswith("O")
    np.testing.assert_allclose(obs, f.observation)


def test_simulation_plan_and_fused_op_table(monkeypatch):
    """The clique simulation schedule (SimulationBasedSampler.plan) and its compilation into `nfisam_sim_op`s
    (sampler.DeviceSimulation.FusedSimulationBackend) without a GPU: the kernel launch is replaced by a recorder."""
    import torch
    import nfisam_hip as nh
    from factors.Factors import AmbiguousDataAssociationFactor, UnarySE2ApproximateGaussianPriorFactor
    from sampler.DeviceSimulation import DeviceSimulationUnsupported, FusedSimulationBackend
    X = [SE2Variable("X%d" % i) for i in range(3)]
    L0, L1 = R2Variable("L0", VariableType.Landmark), R2Variable("L1", VariableType.Landmark)
    cov = np.diag([0.2, 0.04, 0.02]) ** 2
    fs = [UnarySE2ApproximateGaussianPriorFactor(X[0], np.array([1.0, -2.0, 0.3]), np.diag([0.3, 0.2, 0.05]) ** 2),
          SE2RelativeGaussianLikelihoodFactor(X[0], X[1], np.array([5.0, 0.5, 0.4]), cov),
          SE2RelativeGaussianLikelihoodFactor(X[1], X[2], np.array([5.0, -0.5, -0.2]), cov),
          SE2R2RangeGaussianLikelihoodFactor(X[0], L0, 12.0, 0.5),
          SE2R2RangeGaussianLikelihoodFactor(X[1], L1, 9.0, 0.5),
          SE2R2RangeGaussianLikelihoodFactor(X[2], L0, 11.0, 0.5),
          AmbiguousDataAssociationFactor(X[2], [L0, L1], np.array([0.25, 0.75]), SE2R2RangeGaussianLikelihoodFactor, 10.0,
                                         0.5)]
    order = [L0, L1] + X
    sampler = SimulationBasedSampler(fs, order)
    steps = sampler.plan()
    assert [s[0] for s in steps] == ["prior", "draw", "draw", "draw", "draw", "observe", "assoc_obs"]
    assert [str(s[2].name) for s in steps if s[0] == "draw"] == ["X1", "X2", "L0", "L1"]
    # the host execution of the same plan keeps the documented layout [obs | variables in pattern order]
    np.random.seed(0)
    batch, vs, true_obs = sampler.sample(50)
    assert batch.shape == (50, 2 + 2 + 2 + 9) and [str(v.name) for v in vs[2:]] == ["L0", "L1", "X0", "X1", "X2"]
    np.testing.assert_allclose(true_obs, [11.0, 10.0])

    captured = {}

    def fake_launch(ops, n, D_out, D_total, seed, device):
        captured.update(ops=list(ops), n=n, D_out=D_out, D_total=D_total, seed=seed)
        return torch.zeros(n, D_out)
    monkeypatch.setattr(nh, "simulate_clique", fake_launch)
    x, vs2, obs2 = sampler.sample(50, backend=FusedSimulationBackend("cpu"))
    assert tuple(x.shape) == (50, 15) and [str(v.name) for v in vs2] == [str(v.name) for v in vs]
    np.testing.assert_allclose(obs2, true_obs)
    ops = captured["ops"]
    assert captured["D_out"] == captured["D_total"] == 15 and 0 <= captured["seed"] < 2 ** 62
    assert [o.code for o in ops] == [nh.SIM_PRIOR_SE2, nh.SIM_REL_FWD, nh.SIM_REL_FWD, nh.SIM_RING, nh.SIM_RING,
                                     nh.SIM_RANGE_OBS, nh.SIM_ADA_OBS]
    # columns: obs 0,1 | L0 2-3 | L1 4-5 | X0 6-8 | X1 9-11 | X2 12-14
    assert (ops[0].c, ops[1].a, ops[1].c, ops[2].a, ops[2].c) == (6, 6, 9, 9, 12)
    assert (ops[3].a, ops[3].c, ops[4].a, ops[4].c) == (6, 2, 9, 4)
    assert (ops[5].a, ops[5].b, ops[5].c) == (12, 2, 0)
    ada = ops[6]
    assert (ada.a, ada.c, ada.k, list(ada.cand)[:2]) == (12, 1, 2, [2, 4])
    np.testing.assert_allclose(list(ada.p)[:5], [0.25, 1.0, 1.0, 1.0, 0.5], rtol=1e-6)
    np.testing.assert_allclose(list(ops[1].p)[:3], [5.0, 0.5, 0.4], rtol=1e-6)
    np.testing.assert_allclose(list(ops[1].p)[3:], [0.2, 0, 0.04, 0, 0, 0.02], atol=1e-7)
    np.testing.assert_allclose(list(ops[3].p)[:2], [12.0, 0.5])

    # a factor type without a device sampler makes the backend decline (the solver then simulates on the host)
    class Odd(SE2RelativeGaussianLikelihoodFactor):
        pass
    odd = Odd(X[0], X[1], np.array([1.0, 0, 0]), cov)
    odd._correlated_Rt = False
    with pytest.raises(DeviceSimulationUnsupported):
        SimulationBasedSampler([fs[0], odd], [X[0], X[1]]).sample(10, backend=FusedSimulationBackend("cpu"))


def test_null_hypothesis_factor_and_its_place_in_the_schedule():
    """BinaryFactorWithNullHypo (reference: src/factors/Factors.py:3300-3462): a range measurement that is an outlier
    with probability w1 (same measurement, sigma inflated by null_sigma_scale); text round trip, mixture statistics, and
    the sampler schedules such factors after all plain binary factors (src/sampler/SimulationBasedSampler.py:38-45)."""
    from factors.Factors import BinaryFactorWithNullHypo, UnarySE2ApproximateGaussianPriorFactor
    X0, X1 = SE2Variable("X0"), SE2Variable("X1")
    L0 = R2Variable("L0", VariableType.Landmark)
    nhf = BinaryFactorWithNullHypo(X0, L0, np.array([0.8, 0.2]), SE2R2RangeGaussianLikelihoodFactor, 10.0, 0.5,
                                   null_sigma_scale=8.0)
    again = Factor.construct_from_text(str(nhf), [X0, X1, L0])
    assert isinstance(again, BinaryFactorWithNullHypo) and str(again) == str(nhf) and again.null_sigma_scale == 8.0
    np.random.seed(0)
    x = np.zeros((40000, 3))
    lm = nhf.sample(var1=x)
    r = np.hypot(lm[:, 0], lm[:, 1])
    assert abs(r.mean() - 10.0) < 0.05 and abs(r.std() - np.sqrt(0.8 * 0.25 + 0.2 * 16.0)) < 0.05
    obs = nhf.sample(var1=x, var2=np.tile([10.0, 0.0], (40000, 1)))
    assert obs.shape == (40000, 1) and abs(obs.std() - np.sqrt(0.8 * 0.25 + 0.2 * 16.0)) < 0.05
    # kurtosis of a two-scale mixture is far from Gaussian
    z = (obs[:, 0] - obs.mean()) / obs.std()
    assert (z ** 4).mean() > 6.0
    prior = UnarySE2ApproximateGaussianPriorFactor(X0, np.zeros(3), np.diag([0.1, 0.1, 0.01]))
    odo = SE2RelativeGaussianLikelihoodFactor(X0, X1, np.array([5.0, 0, 0]), np.diag([0.04, 0.01, 0.001]))
    rng_f = SE2R2RangeGaussianLikelihoodFactor(X1, L0, 7.0, 0.5)
    steps = SimulationBasedSampler([prior, nhf, odo, rng_f], [L0, X0, X1]).plan()
    # plain binaries first: odometry draws X1, the plain range draws L0 from X1; the null-hypothesis factor then has
    # both ends sampled and becomes a simulated measurement
    assert [(s[0], type(s[1]).__name__) for s in steps] == [
        ("prior", "UnarySE2ApproximateGaussianPriorFactor"), ("draw", "SE2RelativeGaussianLikelihoodFactor"),
        ("draw", "SE2R2RangeGaussianLikelihoodFactor"), ("observe", "BinaryFactorWithNullHypo")]
    batch, vs, true_obs = SimulationBasedSampler([prior, nhf, odo, rng_f], [L0, X0, X1]).sample(200)
    assert batch.shape == (200, 1 + 2 + 3 + 3) and np.allclose(true_obs, [10.0])
