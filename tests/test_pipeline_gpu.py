"""End-to-end pipeline parity against runs of the REFERENCE ITSELF (BASELINE configs 0 / 3 / 4, + the ICRA case).

Fixtures: tests/golden/pipeline_{small_range,plaza1,plaza1ada,manhattan136,icra}.npz, written by
tests/golden/make_pipeline_fixture.py in the build container: the reference's `run_incrementally`
(src/slam/FactorGraphSolver.py:760-933) over the first updates of each dataset WITH THE ARGUMENTS AND THE ITERATION BUDGET OF
THE REFERENCE'S OWN RUN SCRIPTS (2000 iterations + window early stop; Manhattan: 500 fixed; ICRA: run1/parameters), 5-8
seeds, storing (i) what it fed to `fit_clique_density_model` (FactorGraphSolver.py:479-495: variable ordering, true
observations, a row subsample of the training batch, iterations run) and (ii) the posterior samples + ordering of every step.
The reference imports TransportMaps at module level (absent here); the generator supplies a restatement of its
`GaussianDistribution` (the one class the NF-iSAM path calls), so the fixtures pin the pipeline IN DISTRIBUTION ONLY --
by construction, and unavoidably: the reference never seeds torch.

Here this repository's solver runs the same updates with the SAME arguments (3 seeds) and is compared
  (i)  per trained clique: same variable ordering and true observations as the reference; MMD of the simulated training
       batch (device simulator: csrc/clique_sim.hip) against the reference's batches of that clique;
  (ii) per step: same elimination ordering; then
       * joint MMDb (the reference's estimator: RBF, sigma = sqrt(dim), xy columns, src/utils/Statistics.py:68-84) on columns
         STANDARDISED by the reference's pooled spread, and ONLY while it has dynamic range (<= 20 columns).  In raw metres --
         the reference's own usage -- an RBF of width sqrt(dim) m sees no two samples of a 100 m world as neighbours and the
         statistic sits at its floor sqrt(2 / n) whatever the samples are (measured: small-range step 0, six columns: ours
         0.0632, reference spread 0.0636, floor 0.0632); the same happens beyond ~20 standardised columns;
       * BLOCK-WISE MMDb at every size: every (pose_k xy, landmark_j xy) block and every consecutive-pose block
         (pose_k xy, pose_k+1 xy), 4 columns each, standardised by the reference's pooled spread of those columns; statistic =
         the LARGEST block value (and the mean over blocks), which keeps the cross-variable structure the joint metric
         was meant to see (a landmark mode that does not move with its pose shows up in that block);
       * every variable's standardised xy marginal (mean over variables).
Tolerance: statistic = median over the reference's seeds of MMD(ours, reference seed); bound =
max(0.08, 1.5 x the reference's own spread), spread = the same statistic of the reference's own runs (each seed against the
other ones; the largest leave-one-out value) at that step / clique / block set.
`compare_case` returns every row it evaluated; scripts/pipeline_report.py dumps them (profiles/r04_pipeline_parity_vs_reference.json).
"""
import json
import os
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
DATA = os.path.join(HERE, "data")

# case -> graph under tests/data (None: the graph text travels inside a fixture)
CASES = {"small_range": None, "icra": None, "plaza1": "Plaza1EFG", "plaza1ada": "Plaza1ADA0.4EFG",
         "manhattan136": "ManhattanPlaza136"}
JOINT_MAX_COLUMNS = 20


def _xy_block(names, dims, arr, keep_obs=True):
    """obs columns (names starting with 'O', dim 1) + the xy columns of every variable, variables in sorted-name order."""
    cols, off = {}, 0
    for v, d in zip(names, dims):
        cols[v] = arr[:, off:off + (d if d < 3 else 2)]
        off += d
    order = [v for v in names if v.startswith("O")] if keep_obs else []
    order += sorted(v for v in names if not v.startswith("O"))
    return np.hstack([cols[v] for v in order]).astype(np.float64)


def _mmd(a, b):
    from utils.Statistics import MMDb
    return float(MMDb(a, b))


def _band(ours, refs):
    """-> (median MMD of `ours` to the reference's seeds, the reference's own spread in the SAME statistic: the largest
    leave-one-out value among its seeds, i.e. how far one reference run sits from the other reference runs)"""
    to_ref = [_mmd(ours, r) for r in refs]
    n = len(refs)
    pair = np.zeros((n, n))
    for a in range(n):
        for b in range(a + 1, n):
            pair[a, b] = pair[b, a] = _mmd(refs[a], refs[b])
    loo = [float(np.median([pair[a, b] for b in range(n) if b != a])) for a in range(n)]
    return float(np.median(to_ref)), float(max(loo))


def _pose_key(v):
    return int(v[1:]) if v[1:].isdigit() else v


def _blocks(order):
    """4-column blocks of a step: (pose, landmark) for every pair, (pose_k, pose_k+1) along the trajectory."""
    poses = sorted([v for v in order if v.startswith("X")], key=_pose_key)
    lms = sorted(v for v in order if not v.startswith("X"))
    return [(p, l) for p in poses for l in lms] + list(zip(poses[:-1], poses[1:]))


def _graph_path(tmp_path, case, fx):
    if case == "small_range":
        path = tmp_path / "factor_graph.fg"
        path.write_text(str(np.load(os.path.join(GOLDEN, "small_range_case1.npz"))["factor_graph_fg"]))
        return str(path)
    if case == "icra":                              # the reference-held graph file of icra_paper/case1 (data)
        path = tmp_path / "factor_graph"
        path.write_text(str(fx["held_factor_graph"]))
        return str(path)
    return os.path.join(DATA, CASES[case], "factor_graph.fg")


def _run(tmp_path, case, fx, seed):
    from slam.FactorGraphSolver import run_incrementally
    from slam.NFiSAM import NFiSAM, NFiSAMArgs
    from slam.RunBatch import graph_file_parser, group_nodes_factors_incrementally
    kwargs = json.loads(str(fx["arguments"]))
    kwargs["cuda_training"] = True
    n_steps = int(fx["seed0_n_steps"])
    path = _graph_path(tmp_path, case, fx)
    random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)
    nodes, truth, factors = graph_file_parser(path, "fg", prior_cov_scale=0.1)
    steps = group_nodes_factors_incrementally(nodes, factors, incremental_step=int(fx["incremental_step"]))[:n_steps]
    solver = NFiSAM(NFiSAMArgs(**kwargs))
    fits, update_no = [], [0]
    orig_fit, orig_update = NFiSAM.fit_clique_density_model, NFiSAM.update_physical_and_working_graphs

    def fit(self, clique, samples, var_ordering, timer, *a, **k):
        s = samples.detach().cpu().numpy() if torch.is_tensor(samples) else np.asarray(samples)
        fits.append(dict(update=update_no[0] - 1, vars=[str(v.name) for v in var_ordering], dims=[int(v.dim) for v in var_ordering],
                         true_obs=np.asarray(self._clique_true_obs[clique], dtype=np.float64), batch=s.astype(np.float64),
                         name="".join(str(v.name) for v in clique.vars)))
        return orig_fit(self, clique, samples, var_ordering, timer, *a, **k)

    def update(self, *a, **k):
        update_no[0] += 1
        return orig_update(self, *a, **k)
    NFiSAM.fit_clique_density_model, NFiSAM.update_physical_and_working_graphs = fit, update
    try:
        run_dir = run_incrementally(str(tmp_path), solver, steps, truth)
    finally:
        NFiSAM.fit_clique_density_model, NFiSAM.update_physical_and_working_graphs = orig_fit, orig_update
    # iterations run = non-zero entries of the zero-padded loss record (NFiSAM.py:496-497), written per update by run_incrementally
    curves = [json.load(open(os.path.join(run_dir, "step%d_step_training_loss" % i))) for i in range(n_steps)]
    for f in fits:
        f["iterations"] = int(np.count_nonzero(np.asarray(curves[f["update"]].get(f["name"], []), dtype=np.float64)))
    return run_dir, fits, n_steps


def _compare_step(check, seed, i, order, S, ref_raw, max_blocks=None):
    """The step statistics of the module docstring for ONE step: ours `S` [n, columns in `order`] against the reference's
    seeds `ref_raw`.  `max_blocks`: evaluate at most that many 4-column blocks (late steps of a long run have thousands of
    (pose, landmark) pairs: a fixed, evenly spaced subset that always contains the consecutive-pose blocks' ends)."""
    dims = [3 if v.startswith("X") else 2 for v in order]
    assert S.shape[1] == sum(dims) and np.all(np.isfinite(S))
    rr = np.random.RandomState(100 + i).permutation(S.shape[0])[:ref_raw[0].shape[0]]
    Sr = S[rr]
    floor = round(float(np.sqrt(2.0 / Sr.shape[0])), 4)
    if 2 * len(order) <= JOINT_MAX_COLUMNS:
        refs_xy = [_xy_block(order, dims, r) for r in ref_raw]
        sc = np.maximum(np.vstack(refs_xy).std(0), 1e-3)
        m, spread = _band(_xy_block(order, dims, Sr) / sc, [r / sc for r in refs_xy])
        check("step-joint", seed, i, m, spread, columns=2 * len(order), floor=floor)
    # xy columns of every variable: ours, the reference's seeds, the pooled reference scale
    off, col = 0, {}
    for v, d in zip(order, dims):
        col[v] = (off, off + 2)
        off += d
    pooled = np.vstack(ref_raw)
    blocks = _blocks(order)
    if max_blocks is not None and len(blocks) > max_blocks:
        blocks = [blocks[int(q)] for q in np.unique(np.linspace(0, len(blocks) - 1, max_blocks).round())]
    per_block = []
    for a, b in blocks:
        idx = list(range(*col[a])) + list(range(*col[b]))
        sc = np.maximum(pooled[:, idx].std(0), 1e-3)
        mb, sb = _band(Sr[:, idx] / sc, [r[:, idx] / sc for r in ref_raw])
        per_block.append((mb, sb, a + "-" + b))
    if per_block:
        worst = max(per_block, key=lambda t: t[0])
        check("step-blocks-max", seed, i, worst[0], max(t[1] for t in per_block), blocks=len(per_block), worst_block=worst[2], floor=floor)
        check("step-blocks-mean", seed, i, float(np.mean([t[0] for t in per_block])), float(np.mean([t[1] for t in per_block])),
              blocks=len(per_block), floor=floor)
    ours_v, ref_v = [], []
    for v in order:
        idx = list(range(*col[v]))
        sc = np.maximum(pooled[:, idx].std(0), 1e-3)
        mv, sv = _band(Sr[:, idx] / sc, [r[:, idx] / sc for r in ref_raw])
        ours_v.append(mv); ref_v.append(sv)
    check("step-marginals", seed, i, float(np.mean(ours_v)), float(np.mean(ref_v)), variables=len(order), floor=floor)


def compare_case(tmp_path, case, seeds=(0, 1, 2)):
    """-> (rows, failures, timing): every comparison as a dict(kind, seed, index, ours, spread, bound, ...)."""
    fx = np.load(os.path.join(GOLDEN, "pipeline_%s.npz" % case))
    ref_seeds = [int(s) for s in fx["seeds"]]
    rows, failures, timing = [], [], []

    def check(kind, seed, idx, m, spread, **extra):
        bound = max(0.08, 1.5 * spread)
        row = dict(kind=kind, seed=seed, index=idx, ours=round(m, 4), spread=round(spread, 4), bound=round(bound, 4), **extra)
        rows.append(row)
        if not m <= bound:
            failures.append(row)

    for seed in seeds:
        sub = tmp_path / ("seed%d" % seed)
        sub.mkdir()
        run_dir, fits, n_steps = _run(sub, case, fx, seed)
        timing.append([float(t) for t in open(os.path.join(run_dir, "step_timing")).read().split()])
        # ---- (i) what goes into fit_clique_density_model -----------------------------------------------------------
        n_fits = int(fx["seed0_n_fits"])
        assert len(fits) == n_fits, (len(fits), n_fits)
        for j, f in enumerate(fits):
            meta = json.loads(str(fx["seed0_fit%d_meta" % j]))
            obs_names = [v for v in meta["vars"] if v.startswith("O")]
            ref_vars = [v for v in meta["vars"] if not v.startswith("O")]
            assert f["update"] == meta["update"] and f["vars"] in (ref_vars, meta["vars"]), (j, f["vars"], meta["vars"])
            np.testing.assert_allclose(f["true_obs"], fx["seed0_fit%d_true_obs" % j], atol=1e-6)   # same observations, same order
            dims = f["dims"] if f["vars"] == meta["vars"] else [1] * len(obs_names) + f["dims"]
            assert f["batch"].shape[1] == sum(meta["dims"]) == sum(dims)
            refs = [_xy_block(meta["vars"], meta["dims"], fx["seed%d_fit%d_batch" % (s, j)]) for s in ref_seeds]
            scale = np.maximum(np.vstack(refs).std(0), 1e-3)
            rr = np.random.RandomState(j).permutation(f["batch"].shape[0])[:refs[0].shape[0]]
            ours = _xy_block(meta["vars"], dims, f["batch"][rr])
            m, spread = _band(ours / scale, [r / scale for r in refs])
            ref_it = [json.loads(str(fx["seed%d_fit%d_meta" % (s, j)])).get("iterations", -1) for s in ref_seeds]
            check("fit", seed, j, m, spread, vars=meta["vars"], iterations=f["iterations"], reference_iterations=ref_it)
        # ---- (ii) per-step posteriors ------------------------------------------------------------------------------
        for i in range(n_steps):
            order = open(os.path.join(run_dir, "step%d_ordering" % i)).read().split()
            assert order == [str(v) for v in fx["seed0_step%d_ordering" % i]], (i, order)
            S = np.loadtxt(os.path.join(run_dir, "step%d" % i))
            _compare_step(check, seed, i, order, S, [fx["seed%d_step%d_samples" % (s, i)].astype(np.float64) for s in ref_seeds])
    return rows, failures, timing


@pytest.mark.timeout(1800)
@pytest.mark.parametrize("case", list(CASES))
def test_pipeline_matches_reference_runs(tmp_path, case):
    rows, failures, _ = compare_case(tmp_path, case)
    print(case, [(r["kind"], r["seed"], r["index"], r["ours"], r["spread"]) for r in rows])
    # MMDb of two n-sample sets that share no neighbour under the kernel is exactly sqrt(2 / n): no statistic may sit there
    sat = [r for r in rows if r["kind"] != "fit" and abs(r["ours"] - r["floor"]) < 0.01 * r["floor"] and abs(r["spread"] - r["floor"]) < 0.01 * r["floor"]]
    assert not sat, ("a statistic sits at its saturation floor sqrt(2 / n)", sat)
    assert not failures, failures


# ---- the ICRA case against the results the REFERENCE HOLDS for it ---------------------------------------------------------
def _xy_reorder(order, arr, ref_order):
    """icra_paper/compute_mmd.py:72-95 (`reorder_samples`): the xy columns of every variable, in `ref_order`."""
    off, col = 0, {}
    for v in order:
        col[v] = arr[:, off:off + 2]
        off += 3 if v.startswith("X") else 2
    return np.hstack([col[v] for v in ref_order]).astype(np.float64)


@pytest.mark.timeout(900)
def test_icra_case_against_the_reference_held_results(tmp_path):
    """example/slam/small_range_gaussian_problem/icra_paper/case1 with the reference-held run1/parameters (K = 5, n = 600,
    80 iterations, lr .02, 500 posterior samples): per step, with the reference's own statistic (`mmd`,
    src/utils/Statistics.py:13-45, 500 x 500 samples, xy columns in the reference solution's ordering, compute_mmd.py:97-150)
      * steps 0-2 against the reference-held nested-sampling solution (reference/step_0..2; steps 3-5 hold no samples):
        bar = 1.5 x the value the reference published for its own NF-iSAM run at that step (run1/mmd: 0.0149 / 0.125 /
        0.036 -- this file's `mmd` and column order reproduce them from the held files: 0.0149 / 0.13-0.14 depending on the
        500-row subsample / 0.035), or 1.5 x the LARGEST value re-runs of the reference reach (8 seeds, same arguments:
        0.039-0.096 / 0.08-0.13 / 0.032-0.037 -- the stored run's 0.0149 at step 0 is a favourable draw of an 80-iteration
        fit, its own re-runs do not reach it), whichever is larger, floor 0.08;
      * steps 0-5 against the reference-held NF-iSAM posteriors (run1/batch1..6): bar = 1.5 x the LARGEST distance of a
        re-run of the reference (8 seeds, same arguments, fixture) to that stored run at that step, floor 0.08 -- a
        single stored run of an 80-iteration fit is one draw, its distance to another draw is what re-runs say it is."""
    from utils.Statistics import mmd as _mmd_ref

    def mmd(a, b):
        """The reference's statistic (sqrt of the unbiased combination e1 + e2 - 2 e3, src/utils/Statistics.py:13-45) returns
        NaN when that combination is negative (two sample sets closer than the estimator's noise): that is a distance of 0,
        not a missing value -- a NaN must neither pass `assert not m > bar` silently nor drop out of a nanmax."""
        m = float(_mmd_ref(a, b)[0])
        return (0.0 if np.isnan(m) else m,)
    fx = np.load(os.path.join(GOLDEN, "pipeline_icra.npz"))
    ref_seeds = [int(s) for s in fx["seeds"]]
    held_mmd = np.asarray(fx["held_run1_mmd"], dtype=np.float64)
    assert held_mmd.shape == (6,)
    report = []
    for seed in range(3):
        sub = tmp_path / ("seed%d" % seed)
        sub.mkdir()
        run_dir, fits, n_steps = _run(sub, "icra", fx, seed)
        assert n_steps == 6 and [f["iterations"] for f in fits] == [80] * len(fits)          # 80 < 2 windows: the budget runs out
        for i in range(6):
            order = open(os.path.join(run_dir, "step%d_ordering" % i)).read().split()
            assert order == [str(v) for v in fx["held_run1_batch%d_ordering" % (i + 1)]], (i, order)
            S = np.loadtxt(os.path.join(run_dir, "step%d" % i))[:500]
            assert S.shape[0] == 500 and np.all(np.isfinite(S))
            held = fx["held_run1_batch%d" % (i + 1)].astype(np.float64)
            m_run = float(mmd(_xy_reorder(order, S, order), _xy_reorder(order, held, order))[0])
            reruns = [float(mmd(_xy_reorder(order, fx["seed%d_step%d_samples" % (s, i)].astype(np.float64), order),
                                _xy_reorder(order, held, order))[0]) for s in ref_seeds]
            assert np.isfinite(m_run) and np.all(np.isfinite(reruns)), (seed, i, m_run, reruns)
            bar_run = max(0.08, 1.5 * max(reruns))
            report.append(("vs run1/batch%d" % (i + 1), seed, round(m_run, 4), round(float(max(reruns)), 4)))
            assert m_run <= bar_run, (seed, i, m_run, reruns)
            if i <= 2:
                ref_order = [str(v) for v in fx["held_reference_step%d_ordering" % i]]
                R = fx["held_reference_step%d" % i].astype(np.float64)
                R = R[np.random.RandomState(i).permutation(R.shape[0])[:500]]
                Rxy = _xy_reorder(ref_order, R, ref_order)
                m_ns = float(mmd(_xy_reorder(order, S, ref_order), Rxy)[0])
                reruns_ns = [float(mmd(_xy_reorder(order, fx["seed%d_step%d_samples" % (s, i)].astype(np.float64), ref_order), Rxy)[0])
                             for s in ref_seeds]
                assert np.isfinite(m_ns) and np.all(np.isfinite(reruns_ns)), (seed, i, m_ns, reruns_ns)
                bar_ns = max(0.08, 1.5 * held_mmd[i], 1.5 * max(reruns_ns))
                report.append(("vs reference/step_%d" % i, seed, round(m_ns, 4), round(float(held_mmd[i]), 4), round(float(max(reruns_ns)), 4)))
                assert m_ns <= bar_ns, (seed, i, m_ns, held_mmd[i], reruns_ns)
    print("icra", report)


# ---- LONG HORIZON, structural: EVERY update of the large datasets (round 5) ---------------------------------------------------
STRUCTURE_CASES = {"manhattan136_structure": "ManhattanPlaza136", "plaza1_structure": "Plaza1EFG", "plaza1ada_structure": "Plaza1ADA0.4EFG",
                   # round 6: the other six cases of the reference's Plaza loop (example/slam/plaza_dataset/run_nfisam.py:11-12)
                   "plaza2_structure": "Plaza2EFG", "plaza2ada02_structure": "Plaza2ADA0.2EFG", "plaza2ada04_structure": "Plaza2ADA0.4EFG",
                   "plaza2ada06_structure": "Plaza2ADA0.6EFG", "plaza1ada02_structure": "Plaza1ADA0.2EFG", "plaza1ada06_structure": "Plaza1ADA0.6EFG"}


def _run_structure(case, fx):
    """This repository's solver over ALL updates of the dataset with the fixture's arguments (flow_iterations = 20: the fit is
    irrelevant to the structure), recording per update what the reference run recorded (make_pipeline_fixture.py, LONG_CASES):
    elimination ordering, every retrained clique (column pattern, dims, D, D_s, true observations) in training order, every
    firing of `root_clique_density_model_to_leaf`."""
    from slam.NFiSAM import NFiSAM, NFiSAMArgs
    from slam.RunBatch import graph_file_parser, group_nodes_factors_incrementally
    kwargs = json.loads(str(fx["arguments"]))
    kwargs["cuda_training"] = True
    path = os.path.join(DATA, STRUCTURE_CASES[case], "factor_graph.fg")
    random.seed(0); np.random.seed(0); torch.manual_seed(0)
    nodes, truth, factors = graph_file_parser(path, "fg", prior_cov_scale=0.1)
    steps = group_nodes_factors_incrementally(nodes, factors, incremental_step=int(fx["incremental_step"]))
    solver = NFiSAM(NFiSAMArgs(**kwargs))
    fits, reuses, orderings, update_no = [], [], [], [-1]
    orig_fit, orig_reuse = NFiSAM.fit_clique_density_model, NFiSAM.root_clique_density_model_to_leaf

    def fit(self, clique, samples, var_ordering, timer, *a, **k):
        D = int(samples.shape[1])
        fits.append(dict(update=update_no[0], vars=[str(v.name) for v in var_ordering], dims=[int(v.dim) for v in var_ordering],
                         frontal=sorted(str(v.name) for v in clique.frontal), separator=sorted(str(v.name) for v in clique.separator),
                         D=D, Ds=D - sum(int(v.dim) for v in clique.frontal),
                         true_obs=np.asarray(self._clique_true_obs[clique], dtype=np.float64).reshape(-1)))
        return orig_fit(self, clique, samples, var_ordering, timer, *a, **k)

    def reuse(self, old_clique, new_clique, device, *a, **k):
        reuses.append(dict(update=update_no[0], vars=sorted(str(v.name) for v in new_clique.vars),
                           old_frontal=sorted(str(v.name) for v in old_clique.frontal),
                           new_frontal=sorted(str(v.name) for v in new_clique.frontal),
                           new_separator=sorted(str(v.name) for v in new_clique.separator)))
        return orig_reuse(self, old_clique, new_clique, device, *a, **k)
    NFiSAM.fit_clique_density_model, NFiSAM.root_clique_density_model_to_leaf = fit, reuse
    try:
        for vs, fs in steps:
            update_no[0] += 1
            for v in vs:
                solver.add_node(v)
            for f in fs:
                solver.add_factor(f)
            solver.update_physical_and_working_graphs()
            res = solver.incremental_inference()
            orderings.append([str(v.name) for v in solver.elimination_ordering])
        last = np.hstack([res[v] for v in solver.elimination_ordering])
    finally:
        NFiSAM.fit_clique_density_model, NFiSAM.root_clique_density_model_to_leaf = orig_fit, orig_reuse
    return fits, reuses, orderings, last


@pytest.mark.timeout(1800)
@pytest.mark.parametrize("case", list(STRUCTURE_CASES))
def test_structure_of_every_update_matches_the_reference(case):
    """Long-horizon STRUCTURAL parity (VERDICT r4 missing #1): the reference's `run_incrementally`
    (src/slam/FactorGraphSolver.py:760-933) was run over ALL updates of Manhattan-136 (136), Plaza1 and Plaza1-ADA-0.4 (156
    each, 778 poses) with `flow_iterations = 20`, PYTHONHASHSEED = 0; per update the fixture holds the elimination ordering,
    every retrained clique's variable pattern / dims / D / D_s / true observations in training order, and every firing of
    `root_clique_density_model_to_leaf` (src/slam/NFiSAM.py:550-577, decided at FactorGraphSolver.py:306-340).  This repository's
    solver must reproduce ALL of it: the late-run machinery -- re-elimination, re-use of the old root's model, re-attached
    subtrees, 100+-clique trees -- is decided by the graph alone, so equality is exact (observations to 1e-6)."""
    path = os.path.join(GOLDEN, "pipeline_%s.npz" % case)
    if not os.path.exists(path):
        pytest.skip("fixture %s not generated (tests/golden/make_pipeline_fixture.py %s)" % (os.path.basename(path), case))
    fx = np.load(path)
    n_steps, n_fits = int(fx["seed0_n_steps"]), int(fx["seed0_n_fits"])
    fits, reuses, orderings, last = _run_structure(case, fx)
    assert len(orderings) == n_steps
    for i in range(n_steps):
        assert orderings[i] == [str(v) for v in fx["seed0_step%d_ordering" % i]], (i, orderings[i][:8])
    ref_reuses = json.loads(str(fx["seed0_reuses"]))
    assert reuses == ref_reuses, (len(reuses), len(ref_reuses), [r for r in reuses if r not in ref_reuses][:2], [r for r in ref_reuses if r not in reuses][:2])
    assert len(fits) == n_fits, (len(fits), n_fits)
    for j, f in enumerate(fits):
        meta = json.loads(str(fx["seed0_fit%d_meta" % j]))
        obs_names = [v for v in meta["vars"] if v.startswith("O")]
        ref_vars = [v for v in meta["vars"] if not v.startswith("O")]
        assert f["update"] == meta["update"] and f["vars"] in (ref_vars, meta["vars"]), (j, f["update"], meta["update"], f["vars"], meta["vars"])
        dims = f["dims"] if f["vars"] == meta["vars"] else [1] * len(obs_names) + f["dims"]
        assert dims == meta["dims"] and f["D"] == meta["D"] and f["Ds"] == meta["Ds"], (j, f["D"], meta["D"], f["Ds"], meta["Ds"])
        assert f["frontal"] == meta["frontal"] and f["separator"] == meta["separator"], (j, f["frontal"], meta["frontal"])
        np.testing.assert_allclose(f["true_obs"], np.asarray(fx["seed0_fit%d_true_obs" % j], dtype=np.float64).reshape(-1), atol=1e-6)
    assert np.all(np.isfinite(last))
    print(case, "updates", n_steps, "retrained cliques", n_fits, "re-used roots", len(reuses), "widest clique", max(f["D"] for f in fits))


# ---- LONG HORIZON, distributional: Manhattan-136 COMPLETE at the reference's own budget (round 5) -------------------------------
LATE_STEPS = (20, 60, 135)


def compare_late(case="manhattan136_full", seeds=(0, 1, 2), rmse_seeds=(0, 1, 2, 3, 4, 5), data="ManhattanPlaza136", late_steps=LATE_STEPS):
    """-> (rows, failures): this repository's solver over ALL 136 updates with the reference's own arguments (500 fixed
    iterations per fit, manhattan_plaza/run_nfisam.py) against the reference's complete runs (fixture
    pipeline_manhattan136_full.npz: posteriors kept at updates 20 / 60 / 135), with the block-wise / marginal statistics and the
    bound max(0.08, 1.5 x the reference's own spread) of `compare_case` for `seeds`, and -- over `rmse_seeds` (a run takes ~3 s) -- the
    RMSE of the posterior-mean trajectory against the .fg ground truth: MEDIAN over our seeds <= 1.5 x the worst reference seed
    + 0.25 m (single runs scatter widely on this graph, ours and the reference's alike: 59 ambiguous associations)."""
    from slam.NFiSAM import NFiSAM, NFiSAMArgs
    from slam.RunBatch import graph_file_parser, group_nodes_factors_incrementally
    fx = np.load(os.path.join(GOLDEN, "pipeline_%s.npz" % case))
    ref_seeds = [int(s) for s in fx["seeds"]]
    kwargs = json.loads(str(fx["arguments"]))
    kwargs["cuda_training"] = True
    rows, failures = [], []

    def check(kind, seed, idx, m, spread, **extra):
        bound = max(0.08, 1.5 * spread)
        row = dict(kind=kind, seed=seed, index=idx, ours=round(m, 4), spread=round(spread, 4), bound=round(bound, 4), **extra)
        rows.append(row)
        if not m <= bound:
            failures.append(row)
    path = os.path.join(DATA, data, "factor_graph.fg")
    LATE_STEPS = late_steps                                   # (the checkpoints of THIS case)
    rm_ours, rm_ref = {i: [] for i in LATE_STEPS}, {}
    for seed in sorted(set(seeds) | set(rmse_seeds)):
        random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)
        nodes, truth, factors = graph_file_parser(path, "fg", prior_cov_scale=0.1)
        steps = group_nodes_factors_incrementally(nodes, factors, incremental_step=int(fx["incremental_step"]))[:int(fx["seed0_n_steps"])]
        assert len(steps) == int(fx["seed0_n_steps"]) and (case != "manhattan136_full" or len(steps) == 136)
        solver = NFiSAM(NFiSAMArgs(**kwargs))
        for i, (vs, fs) in enumerate(steps):
            for v in vs:
                solver.add_node(v)
            for f in fs:
                solver.add_factor(f)
            solver.update_physical_and_working_graphs()
            res = solver.incremental_inference()
            order = [str(v.name) for v in solver.elimination_ordering]
            assert order == [str(v) for v in fx["seed0_step%d_ordering" % i]], (i, order[:6])
            if i in LATE_STEPS:
                S = np.hstack([res[v] for v in solver.elimination_ordering])
                ref_raw = [fx["seed%d_step%d_samples" % (s, i)].astype(np.float64) for s in ref_seeds]
                if seed in seeds:
                    _compare_step(check, seed, i, order, S, ref_raw, max_blocks=60)
                off, cols = 0, {}
                for v in solver.elimination_ordering:
                    cols[str(v.name)] = (off, v)
                    off += v.dim
                poses = [n for n in order if n.startswith("X")]
                tr = np.array([truth[cols[n][1]][:2] for n in poses])

                def rmse(A):
                    mean = np.array([A[:, cols[n][0]:cols[n][0] + 2].mean(0) for n in poses])
                    return float(np.sqrt(((mean - tr) ** 2).sum(1).mean()))
                if seed in rmse_seeds:
                    rm_ours[i].append(rmse(S))
                rm_ref[i] = [rmse(r) for r in ref_raw]
    for i in LATE_STEPS:
        row = dict(kind="trajectory-rmse", seed=-1, index=i, ours=round(float(np.median(rm_ours[i])), 3), ours_per_seed=[round(v, 3) for v in rm_ours[i]],
                   reference_per_seed=[round(v, 3) for v in rm_ref[i]], bound=round(1.5 * max(rm_ref[i]) + 0.25, 3), floor=0.0,
                   spread=round(max(rm_ref[i]) - min(rm_ref[i]), 3))
        rows.append(row)
        if not row["ours"] <= row["bound"]:
            failures.append(row)
    return rows, failures


@pytest.mark.timeout(1800)
def test_late_posteriors_of_the_complete_manhattan_run_match_the_reference():
    """Long-horizon DISTRIBUTIONAL parity (VERDICT r4 missing #1, ii): the reference ran Manhattan-136 to its end -- 136 updates,
    139 trained cliques, 131 re-used roots -- at its own budget (500 fixed iterations per fit, 65-75 CPU-minutes per seed, 6 seeds).
    This repository's solver does the same 136 updates (3 seeds) and is held to the reference's late posteriors at updates 20,
    60 and 135 (the last: 136 poses + 4 landmarks): every variable's standardised xy marginal, and MMDb on (pose, landmark) and
    consecutive-pose blocks (an evenly spaced subset of 60 of the thousands of pairs), bound = max(0.08, 1.5 x the largest
    leave-one-out value among the reference's own seeds).  What this sees that the first six updates cannot: drift accumulated
    through 130 re-uses of the previous root's model, re-eliminated landmark cliques late in the run, the 100+-clique walk.
    Next to them an ACCURACY row per checkpoint: RMSE of the posterior-mean trajectory against the .fg ground truth, median over
    six seeds here, held to 1.5 x the worst of the reference's seeds + 0.25 m (a seed study beside it, DESIGN.md 5: 96 runs here
    4.59 / 5.45 / 6.60 m on average, 24 of the reference 4.87 / 5.47 / 6.52 m, one-sided rank-sum p = 0.83 / 0.76 / 0.48; single runs
    are heavy-tailed, 2-17 m here and 2-13 m there at update 135 -- scripts/exp/late_rmse.py, profiles/r05_manhattan136_late_rmse_*).
    How sharp the distributional rows are, is the REFERENCE'S doing: its own six seeds sit 0.60 / 0.88 / 1.04 apart (largest leave-one-out block MMDb
    at updates 20 / 60 / 135; sqrt 2 = unrelated), so the bound at update 135 is 1.30-1.56 -- a collapse or a displaced trajectory
    fails, a subtle late bias does not; measured here: 0.42-0.53 / 0.64-1.05 / 0.95-1.23, at the reference's own level
    (profiles/r05_pipeline_parity_vs_reference.json).  The sharp test of the late-run machinery is the structural one above."""
    path = os.path.join(GOLDEN, "pipeline_manhattan136_full.npz")
    if not os.path.exists(path):
        pytest.skip("fixture pipeline_manhattan136_full.npz not generated (tests/golden/make_pipeline_fixture.py manhattan136_full)")
    rows, failures = compare_late()
    print("manhattan136_full", [(r["kind"], r["seed"], r["index"], r["ours"], r["spread"], r["bound"]) for r in rows])
    print([r for r in rows if r["kind"] == "trajectory-rmse"])
    sat = [r for r in rows if r["kind"] != "trajectory-rmse" and abs(r["ours"] - r["floor"]) < 0.01 * r["floor"] and abs(r["spread"] - r["floor"]) < 0.01 * r["floor"]]
    assert not sat, ("a statistic sits at its saturation floor sqrt(2 / n)", sat)
    assert not failures, failures


@pytest.mark.timeout(1800)
def test_late_trajectory_error_over_many_seeds_is_distributed_like_the_reference_s():
    """The seed study of DESIGN.md 5 as a test.  A single complete Manhattan-136 run ends anywhere between 2 and 17 m of trajectory
    RMSE (the reference's: 2-13 m) -- a run that settles on a wrong association or heading early keeps it -- so the accuracy of the
    late run is a statement about a DISTRIBUTION: 32 seeds here (seconds each on the GPU) against every reference seed there is
    (tests/golden/manhattan136_full_rmse_reference_seeds.json: 65-100 CPU-minutes each) at updates 20 / 60 / 135.  Held: the two-sided
    rank-sum test does not reject at 0.5 % and the means differ by less than 3 standard errors of their difference, per checkpoint.
    History: with the reference's first twelve seeds this read 6.6 against 4.9 m at update 135 (p = 0.03 one-sided); with twenty-four,
    6.60 against 6.52.  Sensitivity: ~1.7 m of systematic late error (a quarter of the run-to-run sd x 3 se) would fail the mean
    criterion; the structural test above and the distributional rows of the previous test are the sharper ones for everything
    they cover."""
    from scipy.stats import mannwhitneyu
    path = os.path.join(GOLDEN, "manhattan136_full_rmse_reference_seeds.json")
    if not (os.path.exists(path) and os.path.exists(os.path.join(GOLDEN, "pipeline_manhattan136_full.npz"))):
        pytest.skip("reference seed study not generated (scripts/exp/late_summary.py)")
    ref = json.load(open(path))
    rows, _ = compare_late(seeds=(), rmse_seeds=tuple(range(32)))
    failures = []
    for r in rows:
        assert r["kind"] == "trajectory-rmse"
        a, b = np.array(r["ours_per_seed"]), np.array(ref["rmse"][str(r["index"])])
        assert len(a) == 32 and len(b) == len(ref["seeds"]) >= 18
        p = float(mannwhitneyu(a, b, alternative="two-sided").pvalue)
        gap = abs(a.mean() - b.mean()) / np.sqrt(a.var(ddof=1) / len(a) + b.var(ddof=1) / len(b))
        print("update %d: ours %.2f +- %.2f (median %.2f), reference %.2f +- %.2f (median %.2f, %d seeds), rank-sum p %.3f, mean gap %.2f se" % (
            r["index"], a.mean(), a.std(ddof=1) / np.sqrt(len(a)), np.median(a), b.mean(), b.std(ddof=1) / np.sqrt(len(b)), np.median(b), len(b), p, gap))
        if not (p >= 0.005 and gap <= 3.0):
            failures.append((r["index"], p, gap))
    assert not failures, failures


# ---- LONG HORIZON on the graph BASELINE's north_star names: Plaza1 through update 30 at the reference's own budget (round 6) ----------
PLAZA_LATE_STEPS = (10, 20, 30)


@pytest.mark.timeout(1800)
def test_late_posteriors_of_plaza1_match_the_reference():
    """VERDICT r5 missing #2 / next #3(i): the reference ran `Plaza1EFG` (778 poses; example/slam/plaza_dataset/run_nfisam.py:5-21:
    K = 9, n = 2000, <= 2000 iterations + window rule, lr .01, incremental_step 5) through update 30 -- 155 poses, 4 landmarks, 154
    trained cliques, 30 re-used roots -- at its own budget, SIX seeds (their number fixed before any run was looked at;
    tests/golden/make_pipeline_fixture.py plaza1_late; per-seed provenance -- generator commit, PYTHONHASHSEED, thread counts, host --
    inside the fixture).  This repository's solver does the same 31 updates (3 seeds for the distributional rows, 12 for the accuracy
    row) and is held at updates 10, 20 and 30 to: every variable's standardised xy marginal and MMDb on (pose, landmark) and
    consecutive-pose blocks (an evenly spaced subset of 60), bound = max(0.08, 1.5 x the largest leave-one-out value among the
    reference's own seeds); the RMSE of the posterior-mean trajectory against the .fg ground truth, median over our seeds <= 1.5 x the
    worst reference seed + 0.25 m; and a two-sided rank-sum test of our twelve RMSEs against the reference's six per checkpoint,
    p >= 0.01 (thresholds written down before the fixture existed)."""
    from scipy.stats import mannwhitneyu
    path = os.path.join(GOLDEN, "pipeline_plaza1_late.npz")
    if not os.path.exists(path):
        pytest.skip("fixture pipeline_plaza1_late.npz not generated (tests/golden/make_pipeline_fixture.py plaza1_late: ~2 h of CPU per seed)")
    rows, failures = compare_late("plaza1_late", seeds=(0, 1, 2), rmse_seeds=tuple(range(12)), data="Plaza1EFG", late_steps=PLAZA_LATE_STEPS)
    print("plaza1_late", [(r["kind"], r["seed"], r["index"], r["ours"], r["spread"], r["bound"]) for r in rows])
    sat = [r for r in rows if r["kind"] != "trajectory-rmse" and abs(r["ours"] - r["floor"]) < 0.01 * r["floor"] and abs(r["spread"] - r["floor"]) < 0.01 * r["floor"]]
    assert not sat, ("a statistic sits at its saturation floor sqrt(2 / n)", sat)
    for r in rows:
        if r["kind"] == "trajectory-rmse":
            a, b = np.array(r["ours_per_seed"]), np.array(r["reference_per_seed"])
            p = float(mannwhitneyu(a, b, alternative="two-sided").pvalue)
            print("update %d: trajectory RMSE ours %.2f +- %.2f m (median %.2f, %d seeds), reference %.2f +- %.2f (median %.2f, %d seeds), rank-sum p %.3f"
                  % (r["index"], a.mean(), a.std(ddof=1) / np.sqrt(len(a)), np.median(a), len(a), b.mean(), b.std(ddof=1) / np.sqrt(len(b)), np.median(b), len(b), p))
            if not p >= 0.01:
                failures.append(dict(r, rank_sum_p=p))
    assert not failures, failures


def test_lazy_posterior_returns_the_same_samples_as_the_synchronous_call(tmp_path):
    """`NFiSAMArgs(lazy_posterior=True)` (round 6; not in the reference): `incremental_inference` returns a mapping that waits for
    the tree walk when it is first read, so the next update's host work runs under the walk.  Same seed, same problem, same
    kernels in the same order on the same stream: the samples of every update are EQUAL to the synchronous solver's, whether they
    are read at once or only after the next update's graph work; the mapping behaves like the dict (keys, length, membership);
    the update's loss curves (`_temp_training_loss`, which the lazy path copies to the host in front of the walk) are equal too."""
    from slam.NFiSAM import LazyPosterior, NFiSAM, NFiSAMArgs
    from slam.RunBatch import graph_file_parser, group_nodes_factors_incrementally
    fx = np.load(os.path.join(GOLDEN, "pipeline_small_range.npz"), allow_pickle=False)
    kwargs = json.loads(str(fx["arguments"]))
    kwargs["cuda_training"] = True
    path = _graph_path(tmp_path, "small_range", fx)

    def solve(lazy, read_late):
        random.seed(3); np.random.seed(3); torch.manual_seed(3)
        nodes, truth, factors = graph_file_parser(path, "fg", prior_cov_scale=0.1)
        steps = group_nodes_factors_incrementally(nodes, factors, incremental_step=int(fx["incremental_step"]))[:4]
        solver = NFiSAM(NFiSAMArgs(lazy_posterior=lazy, **kwargs))
        out, pending, curves = [], None, []
        for vs, fs in steps:
            for v in vs: solver.add_node(v)
            for f in fs: solver.add_factor(f)
            solver.update_physical_and_working_graphs()
            if pending is not None:                               # (the previous update's samples, read under this update)
                out.append({str(v.name): np.array(a) for v, a in pending.items()})
                pending = None
            s = solver.incremental_inference()
            assert isinstance(s, LazyPosterior) == lazy
            curves.append({k: list(v) for k, v in solver._temp_training_loss.items()})   # (lazy: copied in front of the walk)
            if read_late:
                pending = s
            else:
                assert len(s) == len(solver.elimination_ordering) and all(v in s for v in solver.elimination_ordering)
                out.append({str(v.name): np.array(s[v]) for v in solver.elimination_ordering})
        if pending is not None:
            out.append({str(v.name): np.array(a) for v, a in pending.items()})
        return out, curves

    ref, ref_curves = solve(False, False)
    assert all(len(c) > 0 for c in ref_curves)
    for lazy, late in ((True, False), (True, True)):
        got, got_curves = solve(lazy, late)
        assert got_curves == ref_curves                          # the loss record of every update's fits, to the bit
        assert len(got) == len(ref) == 4
        for a, b in zip(got, ref):
            assert a.keys() == b.keys()
            for k in a:
                np.testing.assert_array_equal(a[k], b[k], err_msg=k)


def test_async_fits_leave_the_same_posteriors_and_loss_curves_as_the_synchronous_solver(tmp_path):
    """`NFiSAMArgs(async_fits=True)` (round 6; not in the reference): every fit of an update is ENQUEUED as one window-spanning launch
    that evaluates the early-stop rule on the device (`nfisam_nsf_train_plan_launch_async`), the host runs ahead through the upward
    pass and reads the fits' outcomes once behind it.  The window-spanning launch is bit-identical to one launch per chunk
    (tests/test_hip_parity.py), the later cliques read the trained models on the device in stream order, so the whole run is EQUAL to
    the synchronous solver's: posterior samples of every update, the loss curves (hence the iterations run) of every fit -- also
    together with the lazy posterior.  The fits really took the asynchronous path (counted)."""
    from slam.NFiSAM import NFiSAM, NFiSAMArgs
    from slam.RunBatch import graph_file_parser, group_nodes_factors_incrementally
    fx = np.load(os.path.join(GOLDEN, "pipeline_small_range.npz"), allow_pickle=False)
    kwargs = json.loads(str(fx["arguments"]))
    kwargs["cuda_training"] = True
    path = _graph_path(tmp_path, "small_range", fx)

    def solve(**extra):
        random.seed(5); np.random.seed(5); torch.manual_seed(5)
        nodes, truth, factors = graph_file_parser(path, "fg", prior_cov_scale=0.1)
        steps = group_nodes_factors_incrementally(nodes, factors, incremental_step=int(fx["incremental_step"]))[:5]
        solver = NFiSAM(NFiSAMArgs(**extra, **kwargs))
        out, curves = [], []
        for vs, fs in steps:
            for v in vs: solver.add_node(v)
            for f in fs: solver.add_factor(f)
            solver.update_physical_and_working_graphs()
            s = solver.incremental_inference()
            curves.append({k: list(v) for k, v in solver._temp_training_loss.items()})
            out.append({str(v.name): np.array(s[v]) for v in solver.elimination_ordering})
        return out, curves, getattr(solver, "async_fits_enqueued", 0), solver.last_fit_iterations

    ref, ref_curves, n0, it0 = solve()
    assert n0 == 0 and sum(len(c) for c in ref_curves) > 0
    for extra in (dict(async_fits=True), dict(async_fits=True, lazy_posterior=True)):
        got, curves, n_async, it1 = solve(**extra)
        assert n_async == sum(len(c) for c in ref_curves), (n_async, [len(c) for c in ref_curves])     # every fit was enqueued
        assert curves == ref_curves and it1 == it0
        for a, b in zip(got, ref):
            assert a.keys() == b.keys()
            for k in a:
                np.testing.assert_array_equal(a[k], b[k], err_msg=k)
