"""End-to-end pipeline parity against runs of the REFERENCE ITSELF (BASELINE configs 0 / 3 / 4).

Fixtures: tests/golden/pipeline_{small_range,plaza1,manhattan136}.npz, written by tests/golden/make_pipeline_fixture.py in
the build container: the reference's `run_incrementally` (src/slam/FactorGraphSolver.py:760-933) over the first updates of
each dataset with a reduced iteration budget, 5 seeds, storing (i) what it fed to `fit_clique_density_model`
(FactorGraphSolver.py:479-495: variable ordering, true observations, a row subsample of the training batch) and (ii) the
posterior samples + ordering of every step.

Here this repository's solver runs the same updates with the SAME arguments (3 seeds) and is compared
  (i)  per trained clique: same variable ordering and true observations as the reference; MMD of the simulated training
       batch (device simulator: csrc/clique_sim.hip) against the reference's batches of that clique;
  (ii) per step: same elimination ordering; MMDb (the reference's metric: RBF, sigma = sqrt(dim), xy columns,
       src/utils/Statistics.py:68-84) of the posterior against the reference's seed band.
Tolerance: statistic = median over the reference's seeds of MMD(ours, reference seed); bound =
max(0.08, 1.5 x the reference's own spread), spread = the same statistic of the reference's own runs (each seed against the
other four; the largest of the five) at that step / clique.
The reference is badly under-trained at this budget on the multi-modal steps (its own spread reaches 0.3-0.7 there), so
the bound is wide exactly where the reference does not agree with itself.
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

CASES = {"small_range": None, "plaza1": "Plaza1EFG", "manhattan136": "ManhattanPlaza136"}


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


def _run(tmp_path, case, fx, seed):
    from slam.FactorGraphSolver import run_incrementally
    from slam.NFiSAM import NFiSAM, NFiSAMArgs
    from slam.RunBatch import graph_file_parser, group_nodes_factors_incrementally
    kwargs = json.loads(str(fx["arguments"]))
    kwargs["cuda_training"] = True
    n_steps = int(fx["seed0_n_steps"])
    if CASES[case] is None:
        path = tmp_path / "factor_graph.fg"
        path.write_text(str(np.load(os.path.join(GOLDEN, "small_range_case1.npz"))["factor_graph_fg"]))
        path = str(path)
    else:
        path = os.path.join(DATA, CASES[case], "factor_graph.fg")
    random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)
    nodes, truth, factors = graph_file_parser(path, "fg", prior_cov_scale=0.1)
    steps = group_nodes_factors_incrementally(nodes, factors, incremental_step=int(fx["incremental_step"]))[:n_steps]
    solver = NFiSAM(NFiSAMArgs(**kwargs))
    fits, update_no = [], [0]
    orig_fit, orig_update = NFiSAM.fit_clique_density_model, NFiSAM.update_physical_and_working_graphs

    def fit(self, clique, samples, var_ordering, timer, *a, **k):
        s = samples.detach().cpu().numpy() if torch.is_tensor(samples) else np.asarray(samples)
        fits.append(dict(update=update_no[0] - 1, vars=[str(v.name) for v in var_ordering], dims=[int(v.dim) for v in var_ordering],
                         true_obs=np.asarray(self._clique_true_obs[clique], dtype=np.float64), batch=s.astype(np.float64)))
        return orig_fit(self, clique, samples, var_ordering, timer, *a, **k)

    def update(self, *a, **k):
        update_no[0] += 1
        return orig_update(self, *a, **k)
    NFiSAM.fit_clique_density_model, NFiSAM.update_physical_and_working_graphs = fit, update
    try:
        run_dir = run_incrementally(str(tmp_path), solver, steps, truth)
    finally:
        NFiSAM.fit_clique_density_model, NFiSAM.update_physical_and_working_graphs = orig_fit, orig_update
    return run_dir, fits, n_steps


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("case", list(CASES))
def test_pipeline_matches_reference_runs(tmp_path, case):
    fx = np.load(os.path.join(GOLDEN, "pipeline_%s.npz" % case))
    ref_seeds = [int(s) for s in fx["seeds"]]
    report = []
    for seed in range(3):
        sub = tmp_path / ("seed%d" % seed)
        sub.mkdir()
        run_dir, fits, n_steps = _run(sub, case, fx, seed)
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
            rows = np.random.RandomState(j).permutation(f["batch"].shape[0])[:refs[0].shape[0]]
            ours = _xy_block(meta["vars"], dims, f["batch"][rows])
            m, spread = _band(ours / scale, [r / scale for r in refs])
            report.append(("fit", seed, j, round(m, 3), round(spread, 3)))
            assert m <= max(0.08, 1.5 * spread), (case, seed, "fit", j, meta["vars"], m, spread)
        # ---- (ii) per-step posteriors ------------------------------------------------------------------------------
        for i in range(n_steps):
            order = open(os.path.join(run_dir, "step%d_ordering" % i)).read().split()
            assert order == [str(v) for v in fx["seed0_step%d_ordering" % i]], (i, order)
            S = np.loadtxt(os.path.join(run_dir, "step%d" % i))
            dims = [3 if v.startswith("X") else 2 for v in order]
            assert S.shape[1] == sum(dims) and np.all(np.isfinite(S))
            refs = [_xy_block(order, dims, fx["seed%d_step%d_samples" % (s, i)]) for s in ref_seeds]
            rows = np.random.RandomState(100 + i).permutation(S.shape[0])[:refs[0].shape[0]]
            m, spread = _band(_xy_block(order, dims, S[rows]), refs)
            report.append(("step", seed, i, round(m, 3), round(spread, 3)))
            assert m <= max(0.08, 1.5 * spread), (case, seed, "step", i, m, spread)
            # The joint metric saturates at sqrt(2 / n) once the step has tens of variables (an RBF of width sqrt(dim) metres
            # sees no two samples as neighbours), so every variable's xy marginal is also compared on its own, standardised
            # by the reference's pooled spread of that variable: mean over the variables of the same two statistics.
            ours_v, ref_v = [], []
            off = 0
            for v, d in zip(order, dims):
                rv = [fx["seed%d_step%d_samples" % (s2, i)][:, off:off + 2].astype(np.float64) for s2 in ref_seeds]
                sc = np.maximum(np.vstack(rv).std(0), 1e-3)
                mv, sv = _band(S[rows][:, off:off + 2] / sc, [x / sc for x in rv])
                ours_v.append(mv); ref_v.append(sv)
                off += d
            mv, sv = float(np.mean(ours_v)), float(np.mean(ref_v))
            report.append(("step-marginals", seed, i, round(mv, 3), round(sv, 3)))
            assert mv <= max(0.08, 1.5 * sv), (case, seed, "step marginals", i, mv, sv)
    print(case, report)
