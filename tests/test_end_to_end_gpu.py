"""End-to-end posterior parity on BASELINE config[0]: example/slam/small_range_gaussian_problem
(journal_paper/case1: 6 SE2 poses, 2 landmarks, 8 range factors) run incrementally through
`NFiSAM_empirial_study` with the reference's arguments (example run_nfisam.py:12-27 / run1/parameters),
on the MI355X density back end.

Oracle for this test = the reference's own stored results for this exact problem and arguments
(tests/golden/small_range_case1.npz, see make_small_range_fixture.py): its NF-iSAM posterior samples
(run1, steps 0-5) and the dynamic-nested-sampling "ground truth" posteriors (dyn1, steps 0-3).

Metric = the reference's estimator (biased MMD, RBF kernel, sigma = sqrt(dim), xy columns: src/utils/Statistics.py:68-84;
mmd_rmse_time_da_plot_grid.py:167,245) on columns STANDARDISED by the spread of the sample set compared against.  In raw
metres -- the reference's usage -- the statistic is at its floor sqrt(1/n + 1/m) at steps 0-1 whatever the samples are (an RBF
of width sqrt(dim) m in a 100 m world: eight re-runs of the reference all score 0.055 against the stored run AND against
nested sampling there), so round 3's "<= 0.08 at steps 0-2" could not fail; the raw values are still printed.

Tolerances (training is stochastic; the reference never seeds torch, so parity is distributional): statistic = MMDb of
this run to the stored result; bound per step = max(0.08, 1.5 x the LARGEST value the reference's own re-runs reach against
that same stored result) -- eight runs of the reference with these arguments (tests/golden/pipeline_small_range.npz,
make_pipeline_fixture.py; round 3 used a scalar 0.45 / 0.55 at steps 4-5 that had been moved once).  Steps 0-3 are held to
the nested-sampling posteriors the same way; steps 4-5 also to first / second moments of the known ground-truth geometry.
Every variant is run with three seeds and every seed has to meet the tolerances.
"""
import json
import os
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def xy(order, arr):
    cols, off = {}, 0
    for v in order:
        cols[v] = arr[:, off:off + 2]
        off += 3 if v.startswith("X") else 2
    return np.hstack([cols[v] for v in sorted(order)]), cols


@pytest.mark.parametrize("device_simulation", [False, True], ids=["host-simulator", "fused-device-simulator"])
def test_small_range_problem_incremental_posteriors(tmp_path, device_simulation):
    """device simulator: the clique training batches are simulated (one fused kernel per clique) and normalised on the
    GPU (sampler.DeviceSimulation + nfisam_normalize_columns) instead of by the factors' numpy samplers."""
    for seed in (0, 1, 2):          # three seeds per variant: a tolerance met by one lucky seed proves nothing
        sub = tmp_path / ("seed%d" % seed)
        sub.mkdir()
        _one_run(sub, device_simulation, seed)


def _one_run(tmp_path, device_simulation, seed):
    from slam.NFiSAM import NFiSAM_empirial_study
    from utils.Statistics import MMDb
    g = np.load(os.path.join(GOLDEN, "small_range_case1.npz"))
    (tmp_path / "factor_graph.fg").write_text(str(g["factor_graph_fg"]))
    random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)
    ref_args = json.loads(str(g["run1_parameters"]))
    run_dirs = NFiSAM_empirial_study([ref_args["num_knots"]], [ref_args["flow_iterations"]],
                                     [ref_args["local_sample_num"]], [ref_args["learning_rate"]],
                                     [ref_args["hidden_dim"]], str(tmp_path), "factor_graph.fg", "fg",
                                     incremental_step=1, cuda_training=True, elimination_method="pose_first",
                                     training_set_frac=1.0, loss_delta_tol=ref_args["loss_delta_tol"],
                                     posterior_sample_num=ref_args["posterior_sample_num"],
                                     device_simulation=device_simulation)
    rd = run_dirs[0]
    assert json.loads(open(os.path.join(rd, "parameters")).read())["num_knots"] == 9
    fx = np.load(os.path.join(GOLDEN, "pipeline_small_range.npz"))       # eight re-runs of the reference, same arguments

    def standardised(a_xy, b_xy):
        sc = np.maximum(b_xy.std(0), 1e-3)
        return MMDb(a_xy / sc, b_xy / sc)

    def bar(i, target_xy, order):
        """1.5 x the largest distance of a reference re-run to `target_xy` at step i (floor 0.08)"""
        vals = []
        for sd in fx["seeds"]:
            assert [str(v) for v in fx["seed%d_step%d_ordering" % (sd, i)]] == order
            vals.append(standardised(xy(order, fx["seed%d_step%d_samples" % (sd, i)].astype(np.float64))[0], target_xy))
        return max(0.08, 1.5 * max(vals)), max(vals)
    truth = {"X0": (0, 0), "X1": (0, 30), "X2": (30, 30), "X3": (60, 30), "X4": (90, 30), "X5": (90, 0),
             "L1": (30, -30), "L2": (60, -30)}
    expected_cols = [7, 10, 13, 16, 19, 22]
    for i in range(6):
        ours = np.loadtxt(os.path.join(rd, "step%d" % i))
        order = open(os.path.join(rd, "step%d_ordering" % i)).read().split()
        assert order == str(g["run1_step%d_ordering" % i]).split()           # same elimination ordering
        assert ours.shape == (1000, expected_cols[i]) and np.all(np.isfinite(ours))
        loss = json.load(open(os.path.join(rd, "step%d_step_training_loss" % i)))
        assert len(loss) == 1                                                  # one trained clique per update
        l = np.array(list(loss.values())[0])
        it = int(np.count_nonzero(l))
        assert len(l) == 2000 and it % 50 == 0 and 100 <= it <= 2000 and np.all(l[it:] == 0)
        ours_xy, cols = xy(order, ours)
        ref_xy, _ = xy(order, g["run1_step%d" % i])
        m_ref, (tol_ref, worst_rerun) = standardised(ours_xy, ref_xy), bar(i, ref_xy, order)
        print("step %d vs the stored run: %.3f (reference re-runs reach %.3f; raw metres %.3f)" % (i, m_ref, worst_rerun, MMDb(ours_xy, ref_xy)))
        assert m_ref <= tol_ref, (i, m_ref, tol_ref)
        if i <= 3:
            dyn_order = str(g["dyn1_step%d_ordering" % i]).split()
            assert sorted(dyn_order) == sorted(order)
            dyn_xy, _ = xy(dyn_order, g["dyn1_step%d" % i])
            m_dyn, (tol_dyn, worst_dyn) = standardised(ours_xy, dyn_xy), bar(i, dyn_xy, order)
            print("step %d vs nested sampling: %.3f (reference re-runs reach %.3f)" % (i, m_dyn, worst_dyn))
            assert m_dyn <= tol_dyn, (i, m_dyn, tol_dyn)
        # poses are well determined by odometry: within 4 sigma of the ground truth
        for v in order:
            if v.startswith("X"):
                err = np.abs(cols[v].mean(0) - np.array(truth[v]))
                assert np.all(err < 4 * np.maximum(cols[v].std(0), 0.2)), (i, v, err)
    # after the last update both landmarks are resolved (unimodal, near the truth)
    for lm in ("L1", "L2"):
        assert np.linalg.norm(cols[lm].mean(0) - np.array(truth[lm])) < 6.0, cols[lm].mean(0)
    # timers exist and are consistent
    st = [float(t) for t in open(os.path.join(rd, "step_timing")).read().split()]
    ft = [float(t) for t in open(os.path.join(rd, "fitting_timer")).read().split()]
    assert len(st) == len(ft) == 6 and all(f <= s for f, s in zip(ft, st))
