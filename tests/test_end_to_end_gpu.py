"""End-to-end posterior parity on BASELINE config[0]: example/slam/small_range_gaussian_problem
(journal_paper/case1: 6 SE2 poses, 2 landmarks, 8 range factors) run incrementally through
`NFiSAM_empirial_study` with the reference's arguments (example run_nfisam.py:12-27 / run1/parameters),
on the MI355X density back end.

Oracle for this test = the reference's own stored results for this exact problem and arguments
(tests/golden/small_range_case1.npz, see make_small_range_fixture.py): its NF-iSAM posterior samples
(run1, steps 0-5) and the dynamic-nested-sampling "ground truth" posteriors (dyn1, steps 0-3).

Metric = the reference's: biased MMD with an RBF kernel, sigma = sqrt(dim), on the xy columns
(src/utils/Statistics.py:68-84; mmd_rmse_time_da_plot_grid.py:167,245).

Stated tolerances (training is stochastic; the reference never seeds torch, so parity is distributional):
  * steps 0-2 (near-Gaussian / single ring): MMDb <= 0.08 against nested sampling AND against the
    reference's NF run — the noise floor of two 1000-sample sets is 0.045-0.063;
  * step 3 (bimodal landmark): <= 0.315 = 1.5 x 0.21, the reference's own run-to-run spread at this step
    (stored run vs re-run 0.21; stored run vs nested 0.14);
  * steps 4-5 (no nested-sampling blobs in the checkout, a single stored reference run): <= 0.55 against
    that run (8 seeds on MI355X: 0.18-0.37, one seed 0.45 after a rounding-level kernel change), plus first/second-moment
    checks against the known ground-truth geometry.  0.55 = 1.5 x the REFERENCE'S OWN run-to-run spread at these steps
    (tests/golden/pipeline_small_range.npz, five reference runs: median pairwise MMDb 0.34 / 0.38, maximum 0.57 / 0.70);
    tests/test_pipeline_gpu.py holds the same steps to that five-seed band directly.
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
        m_ref = MMDb(ours_xy, ref_xy)
        # steps 4-5 are compared with the reference's SINGLE stored run of a multi-modal posterior: loose
        tol_ref = 0.08 if i <= 2 else (0.315 if i == 3 else 0.55)
        assert m_ref <= tol_ref, (i, m_ref)
        if i <= 3:
            dyn_xy, _ = xy(str(g["dyn1_step%d_ordering" % i]).split(), g["dyn1_step%d" % i])
            m_dyn = MMDb(ours_xy, dyn_xy)
            assert m_dyn <= (0.08 if i <= 2 else 0.315), (i, m_dyn)
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
