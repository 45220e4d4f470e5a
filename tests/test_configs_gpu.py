"""GPU parity tests at the sizes of BASELINE.json configs[2..4] ("configs_untested"):

  config[2] "C3"       8 ragged cliques D = 6 8 8 10 10 12 12 12, n = 2000, K = 9 — ONE batched launch sequence
                       against the C oracle per clique (20 Adam iterations), then 500 iterations + properties
  config[3]/[4] shapes n = 2000, D in {15, 16, 17}, L = 1, K = 9 (Manhattan / Plaza1 cliques): NLL gradients against the
                       float64 oracle and a 10-step Adam trajectory against the float32 oracle, for both training-kernel
                       families and for the single-wave-block `grid.z` launch explicitly
  config[3]/[4] end to end: the first updates of tests/data/Plaza1EFG and tests/data/Manhattan200 (ambiguous
                       data-association factors) through `NFiSAM_empirial_study` with the reference's arguments
                       (example/slam/plaza_dataset/run_nfisam.py:5-21, manhattan_plaza/run_nfisam.py:5-50)
  a12                  `separator_forward` / `FlowsPriorFactor.log_pdf` against the oracle's marginal-flow log-prob

Tolerances as in tests/test_hip_parity.py: loss 5e-4 abs, parameters after k Adam steps q99 < 2e-3
(Adam turns fp32-noise-level gradients into +-lr steps on a few coordinates in either implementation), gradients
1e-3 rel + 2e-5 abs of the largest entry.
"""
import json
import os
import random

import numpy as np
import pytest
import torch

import bench as BM
import nfisam_hip as nh
from oracle import c_oracle as CO
from oracle import nsf_torch as O

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
K, H, B = 9, 8, 5.0
DATA = os.path.join(os.path.dirname(__file__), "data")


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32).to(DEV)


def clique_problem(shape, n, seed):
    s, circ = BM.ring_clique(n, *shape, np.random.RandomState(seed))
    x, _, _ = BM.normalize(s, circ)
    D = x.shape[1]
    return x, BM.init_blob_np(D, K, H, 1, seed), D


@pytest.fixture(params=["auto", "wide", "split"])
def family(request):
    old = os.environ.get("NFISAM_TRAIN")
    if request.param == "auto":
        os.environ.pop("NFISAM_TRAIN", None)
    else:
        os.environ["NFISAM_TRAIN"] = request.param
    yield request.param
    if old is None:
        os.environ.pop("NFISAM_TRAIN", None)
    else:
        os.environ["NFISAM_TRAIN"] = old


# ---------------------------------------------------------------------------------------------------------
# config[2]: C3
# ---------------------------------------------------------------------------------------------------------
def test_c3_batched_ragged_cliques_against_oracle():
    """20 batched iterations of the 8 cliques (one launch sequence, grid.y = clique) == 8 independent oracle runs."""
    n, iters, lr = 2000, 20, 0.01
    probs = [clique_problem(sh, n, 100 + c) for c, sh in enumerate(BM.C3_SHAPES)]
    assert [p[2] for p in probs] == [6, 8, 8, 10, 10, 12, 12, 12]
    tb = nh.TrainBatch([dev(x) for x, _, _ in probs], [nh.pack(dev(b), D, K, H, 1) for _, b, D in probs], K, H, B, 1,
                       lr=lr, max_iters=iters, early_stop=False)
    assert tb.run(use_graph=True) == [iters] * 8
    for c, (x, blob, D) in enumerate(probs):
        bc, lc, ic, _, _ = CO.train(x, blob, K, H, B, 1, lr=lr, max_iters=iters, early_stop=False, dtype=np.float32)
        np.testing.assert_allclose(tb.iter_loss[c].cpu().numpy(), lc, atol=5e-4, rtol=2e-4)
        err = np.abs(nh.unpack(tb.kparams[c], D, K, H).cpu().numpy() - bc)
        assert np.quantile(err, 0.99) < 2e-3, (c, D, np.quantile(err, 0.99))
        assert err.max() < iters * lr + 1e-3


def test_c3_full_length_run_properties():
    """The configuration as benchmarked (500 fixed iterations): finite monotone-on-average losses, every clique fits
    (loss drops by > 3 nats), the trained flows invert their own forward, and the batched run equals 8 single-clique
    runs of the same kernels bit for bit per clique launch family (gradient slabs: no atomics on the gradient path)."""
    n, iters, lr = 2000, 500, 0.01
    probs = [clique_problem(sh, n, 100 + c) for c, sh in enumerate(BM.C3_SHAPES)]
    xs = [dev(x) for x, _, _ in probs]
    kps = [nh.pack(dev(b), D, K, H, 1) for _, b, D in probs]
    tb = nh.TrainBatch(xs, [k.clone() for k in kps], K, H, B, 1, lr=lr, max_iters=iters, early_stop=False)
    assert tb.run(use_graph=True) == [iters] * 8
    for c, (x, _, D) in enumerate(probs):
        il = tb.iter_loss[c].cpu().numpy()
        assert np.all(np.isfinite(il)) and il[-1] < il[0] - 3.0, (c, il[0], il[-1])
        w = il.reshape(10, 50).mean(1)
        assert np.all(np.diff(w) < 0.1), (c, w)                     # window means decrease (late Adam spikes of ~0.05 nats come and go with kernel rounding)
        z, ld, lp = nh.forward(xs[c], tb.kparams[c], K, H, B, 1, want_logprob=True)
        assert abs(-lp.mean().item() - il[-1]) < 0.3                # last recorded loss ~ NLL of the final model (one Adam step apart)
        xb = nh.inverse(z, None, tb.kparams[c], K, H, B, 1)
        inside = (xs[c].abs().max(1).values < 4.9)
        e = (xb - xs[c])[inside].abs()
        assert float(e.mean()) < 5e-5 and float(e.max()) < 5e-2, (c, float(e.mean()), float(e.max()))
        # the float64 oracle evaluates the trained parameters to the same NLL
        blob = nh.unpack(tb.kparams[c], D, K, H).cpu().numpy()
        lossc, _, _, _ = CO.nll_grad(x, blob, K, H, B, 1, dtype=np.float64)
        assert abs(lossc + lp.mean().item()) < 5e-4, (c, lossc, -lp.mean().item())


# ---------------------------------------------------------------------------------------------------------
# config[3] / config[4] clique shapes
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("D", [15, 16, 17])
def test_plaza_shape_gradients_and_adam_against_oracle(D, family):
    n = 2000
    x, blob, D_ = clique_problem(BM.SHAPE_OF_D[D], n, 300 + D)
    assert D_ == D
    kp = nh.pack(dev(blob), D, K, H, 1)
    lossc, gradc, _, _ = CO.nll_grad(x, blob, K, H, B, 1, dtype=np.float64)
    kg, _, loss = nh.backward(dev(x), kp, K, H, B, 1, nll_mode=True)
    assert abs(loss.item() / n + 0.5 * D * np.log(2 * np.pi) - lossc) < 3e-4
    grad = nh.unpack(kg, D, K, H).cpu().numpy() / n
    np.testing.assert_allclose(grad, gradc, rtol=1e-3, atol=2e-5 * max(1.0, float(np.abs(gradc).max())))
    bc, lc, _, _, _ = CO.train(x, blob, K, H, B, 1, lr=0.01, max_iters=10, early_stop=False, dtype=np.float32)
    tb = nh.TrainBatch([dev(x)], [kp.clone()], K, H, B, 1, lr=0.01, max_iters=10, early_stop=False)
    assert tb.run(use_graph=True) == [10]
    np.testing.assert_allclose(tb.iter_loss[0].cpu().numpy(), lc, atol=5e-4, rtol=2e-4)
    err = np.abs(nh.unpack(tb.kparams[0], D, K, H).cpu().numpy() - bc)
    assert np.quantile(err, 0.99) < 2e-3 and err.max() < 10 * 0.01 + 1e-3, (np.quantile(err, 0.99), err.max())


def test_plaza_shape_single_wave_blocks_equal_grouped_blocks():
    """n = 2000, D = 15, L = 1 launches every (tile, dim) unit as its own single-wave block (grid.z = dim); a batch of
    three such cliques exceeds the single-wave-block limit and keeps 8 dims per block.  Same kernels, same tiles, same
    slab order: the two launch shapes must give bitwise identical parameters for the shared clique."""
    n, D = 2000, 15
    probs = [clique_problem(BM.PLAZA_SHAPE, n, 400 + c) for c in range(3)]
    os.environ["NFISAM_TRAIN"] = "split"
    try:
        tb1 = nh.TrainBatch([dev(probs[0][0])], [nh.pack(dev(probs[0][1]), D, K, H, 1)], K, H, B, 1, lr=0.01, max_iters=6,
                            early_stop=False)
        assert tb1.run(use_graph=False) == [6]
        tb3 = nh.TrainBatch([dev(p[0]) for p in probs], [nh.pack(dev(p[1]), D, K, H, 1) for p in probs], K, H, B, 1,
                            lr=0.01, max_iters=6, early_stop=False)
        assert tb3.run(use_graph=False) == [6] * 3
    finally:
        del os.environ["NFISAM_TRAIN"]
    assert torch.equal(tb1.kparams[0], tb3.kparams[0])
    bc, lc, _, _, _ = CO.train(probs[0][0], probs[0][1], K, H, B, 1, lr=0.01, max_iters=6, early_stop=False,
                               dtype=np.float32)
    np.testing.assert_allclose(tb1.iter_loss[0].cpu().numpy(), lc, atol=5e-4, rtol=2e-4)


# ---------------------------------------------------------------------------------------------------------
# a12: marginal-flow log-density of the separator factor against the oracle
# ---------------------------------------------------------------------------------------------------------
def test_separator_forward_and_log_pdf_match_oracle_marginal_flow():
    """`separator_forward` (src/slam/NFiSAM.py:157-173) pushes the first `separator_dim` columns through the flow
    truncated to those dims; `FlowsPriorFactor.log_pdf` (NFiSAM.py:233-251) = prior log-prob + log-det of it.  Oracle:
    O.log_prob of the truncated blob on the normalised columns; gradient w.r.t. x by torch autograd through the oracle."""
    from flows.flows import NSF_AR
    from flows.prior_dist import CustomMultivariateNormal
    from slam.NFiSAM import FlowsPriorFactor, NormalizingFlowModelWithSeparator
    from slam.Variables import R2Variable, SE2Variable, VariableType
    D, n_obs, sep_dim, n = 9, 1, 5, 64                    # columns [obs | L0 xy | X0 x y th | X1 x y th]
    rng = np.random.RandomState(4)
    blob = BM.init_blob_np(D, K, H, 1, 11) + 0.2 * rng.randn(O.param_count(D, K, H)).astype(np.float32)
    circ = [False, False, False, False, False, True, False, False, True]
    mean = (rng.randn(D) * 3).astype(np.float32); std = (0.5 + rng.rand(D)).astype(np.float32)
    flow = NSF_AR.from_kernel_params(D, K, B, H, nh.pack(dev(blob), D, K, H, 1))
    Ds = n_obs + sep_dim
    model = NormalizingFlowModelWithSeparator([flow], CustomMultivariateNormal(dim=D, device=DEV),
                                              CustomMultivariateNormal(dim=Ds, device=DEV), circ, torch.tensor(mean),
                                              torch.tensor(std))
    xs = (mean[:Ds] + std[:Ds] * rng.randn(n, Ds) * 1.2).astype(np.float32)
    z, lp, ld = model.separator_forward(xs.copy())
    xn = O.normalize_samples(xs, mean, std, np.array(circ), 0)
    Pt = O.param_count(Ds, K, H)
    zo, ldo = O.forward(torch.tensor(xn), torch.tensor(blob[:Pt]), K, H, B, 1)
    np.testing.assert_allclose(z.cpu().numpy(), zo.numpy(), atol=1e-4)
    np.testing.assert_allclose(ld.cpu().numpy(), ldo.numpy(), atol=2e-4)
    lpo = O.log_prob(torch.tensor(xn), torch.tensor(blob[:Pt]), K, H, B, 1)
    np.testing.assert_allclose((lp + ld).cpu().numpy(), lpo.numpy(), atol=3e-4)
    # the factor seen from the parent: true observation prepended, log_pdf of the separator variables
    L0, X0 = R2Variable("L0", VariableType.Landmark), SE2Variable("X0")
    true_obs = xs[0, :n_obs].astype(np.float64)
    fac = FlowsPriorFactor([L0, X0], model, true_obs, circ[n_obs:Ds])
    xq = xs[:, n_obs:].astype(np.float64)
    aug = np.concatenate([np.tile(true_obs, (n, 1)), xq], 1).astype(np.float32)
    augn = torch.tensor(O.normalize_samples(aug, mean, std, np.array(circ), 0), requires_grad=True)
    lpo2 = O.log_prob(augn, torch.tensor(blob[:Pt]), K, H, B, 1)
    np.testing.assert_allclose(fac.log_pdf(xq), lpo2.detach().numpy(), atol=3e-4)
    (go,) = torch.autograd.grad(lpo2.sum(), augn)
    gref = go.numpy()[:, n_obs:] / std[n_obs:Ds]
    g = fac.grad_x_log_pdf(xq)
    # the analytic gradient jumps at knots (C1 only between them): compare where both agree on the bin, i.e. the bulk
    err = np.abs(g - gref)
    assert np.quantile(err, 0.95) < 2e-3 * max(1.0, np.abs(gref).max()), np.quantile(err, 0.95)


def test_unif_to_sample_is_the_conditional_inverse_of_the_normal_quantiles():
    """`FlowsPriorFactor.unif_to_sample` (nested-sampling prior transform, src/slam/NFiSAM.py:290-303): u in (0,1)^d ->
    z = Phi^-1(u) -> conditional inverse given the true observation, un-normalised.  Oracle: the float64 C oracle's inverse
    of the same flow on the normalised observation, un-normalised (and angle-wrapped) by hand; with and without observation."""
    import scipy.stats
    from flows.flows import NSF_AR
    from flows.prior_dist import CustomMultivariateNormal
    from slam.NFiSAM import FlowsPriorFactor, NormalizingFlowModelWithSeparator
    from slam.Variables import R2Variable, SE2Variable, VariableType
    rng = np.random.RandomState(9)
    for n_obs in (1, 0):
        D, sep_dim = n_obs + 5 + 3, 5                     # columns [obs | L0 xy | X0 x y th | X1 x y th]
        blob = BM.init_blob_np(D, K, H, 1, 5) + 0.2 * rng.randn(O.param_count(D, K, H)).astype(np.float32)
        circ = [False] * n_obs + [False, False, False, False, True, False, False, True]
        mean = (rng.randn(D) * 3).astype(np.float32); std = (0.5 + rng.rand(D)).astype(np.float32)
        flow = NSF_AR.from_kernel_params(D, K, B, H, nh.pack(dev(blob), D, K, H, 1))
        Ds = n_obs + sep_dim
        model = NormalizingFlowModelWithSeparator([flow], CustomMultivariateNormal(dim=D, device=DEV),
                                                  CustomMultivariateNormal(dim=Ds, device=DEV), circ, torch.tensor(mean),
                                                  torch.tensor(std))
        true_obs = np.array([mean[0] + 0.3 * std[0]], dtype=np.float64)[:n_obs]
        fac = FlowsPriorFactor([R2Variable("L0", VariableType.Landmark), SE2Variable("X0")], model, true_obs, circ[n_obs:Ds])
        for _ in range(4):
            u = rng.uniform(0.05, 0.95, size=sep_dim)
            got = np.asarray(fac.unif_to_sample(u), dtype=np.float64)
            assert got.shape == (sep_dim,)
            z = scipy.stats.norm.ppf(u)[None, :]
            obs_n = ((true_obs - mean[:n_obs]) / std[:n_obs])[None, :]
            Pt = O.param_count(Ds, K, H)
            xn = CO.inverse(z, obs_n if n_obs else None, blob[:Pt], K, H, B, 1, dtype=np.float64)[0][0]
            ref = xn * std[n_obs:Ds] + mean[n_obs:Ds]
            cm = np.array(circ[n_obs:Ds])
            ref[cm] = (ref[cm] + np.pi) % (2 * np.pi) - np.pi
            np.testing.assert_allclose(got, ref, atol=2e-3)


# ---------------------------------------------------------------------------------------------------------
# config[3] / config[4] end to end: first updates through NFiSAM_empirial_study
# ---------------------------------------------------------------------------------------------------------
def _run_first_updates(tmp_path, dataset, n_updates, seed, **study_kwargs):
    """Truncate the dataset to its first `n_updates` incremental updates by writing the leading nodes/factors back
    as an .fg file is not needed: `NFiSAM_empirial_study` takes the whole file, so the solver is driven step by step
    exactly as `run_incrementally` does (src/slam/FactorGraphSolver.py:760-933) and stopped after n_updates."""
    from slam.FactorGraphSolver import run_incrementally
    from slam.NFiSAM import NFiSAM, NFiSAMArgs
    from slam.RunBatch import graph_file_parser, group_nodes_factors_incrementally
    random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)
    step = study_kwargs.pop("incremental_step")
    nodes, truth, factors = graph_file_parser(os.path.join(DATA, dataset, "factor_graph.fg"), "fg", prior_cov_scale=0.1)
    steps = group_nodes_factors_incrementally(nodes, factors, incremental_step=step)[:n_updates]
    solver = NFiSAM(NFiSAMArgs(**study_kwargs))
    run_dir = run_incrementally(str(tmp_path), solver, steps, truth)
    return run_dir, solver, truth, steps


def _trajectory_rmse(solver, truth):
    poses = [v for v in solver.physical_vars if str(v.name).startswith("X")]
    res = solver.results()
    err = np.array([res[v][:, :2].mean(0) - truth[v][:2] for v in poses])
    return float(np.sqrt((err ** 2).sum(1).mean())), len(poses)


PLAZA_ARGS = dict(num_knots=9, flow_iterations=2000, local_sample_num=2000, learning_rate=.01, hidden_dim=8,
                  cuda_training=True, elimination_method="pose_first", training_set_frac=1.0, loss_delta_tol=.01,
                  average_window=50, incremental_step=5)          # plaza_dataset/run_nfisam.py:5-21
MANHATTAN_ARGS = dict(num_knots=9, flow_iterations=500, local_sample_num=2000, learning_rate=.01, hidden_dim=8,
                      cuda_training=True, elimination_method="pose_first", data_parallel=False, training_set_frac=1.0,
                      loss_delta_tol=1e-9, average_window=50, incremental_step=1)   # manhattan_plaza/run_nfisam.py:5-50


@pytest.mark.parametrize("dataset,args,n_updates,rmse_each,rmse_median",
                         [("Plaza1EFG", PLAZA_ARGS, 10, 7.0, 4.0),
                          ("Manhattan200", MANHATTAN_ARGS, 12, 3.5, 2.2)],
                         ids=["plaza1-first-10-updates", "manhattan200-first-12-updates"])
def test_dataset_first_updates_end_to_end(tmp_path, dataset, args, n_updates, rmse_each, rmse_median):
    """Multi-seed band (training and simulation are stochastic; the reference ships no stored results for these
    datasets, so the yardstick is the ground truth in the .fg file): over 3 seeds every run's trajectory RMSE of the
    posterior means stays below `rmse_each` metres and the median below `rmse_median`.  Calibration (8 seeds on
    MI355X, gpurun_out/r2c): Plaza1 after 10 updates (50 poses) 0.5-4.5 m, median 2.1 m -- the landmarks are still
    unresolved there, so the trajectory is odometry-bound: dead reckoning alone is 1.55 m off the GPS truth over
    these poses; a mirrored-mode failure shows as > 10 m (the median of 3 draws from that spread moves with every
    change of kernel rounding: 2.1-3.3 m seen, hence 4.0).  Manhattan-200 after 12 updates: 1.3-2.2 m, median 1.4 m."""
    rmses = []
    for seed in range(3):
        sub = tmp_path / ("seed%d" % seed)
        sub.mkdir()
        run_dir, solver, truth, steps = _run_first_updates(sub, dataset, n_updates, seed, **dict(args))
        files = set(os.listdir(run_dir))
        expect = {"parameters", "step_timing", "step_list", "posterior_sampling_timer", "fitting_timer"}
        for i in range(n_updates):
            expect |= {"step%d" % i, "step%d_ordering" % i, "step%d_split_timing" % i, "step%d_step_training_loss" % i,
                       "step%d_dim_time" % i}
        assert expect <= files, sorted(expect - files)
        seen = []
        for i, (vs, fs) in enumerate(steps):
            seen += [str(v.name) for v in vs]
            order = open(os.path.join(run_dir, "step%d_ordering" % i)).read().split()
            # pose_first (src/slam/FactorGraph.py:108-120): poses in insertion order, then landmarks in insertion order
            assert order == [v for v in seen if v.startswith("X")] + [v for v in seen if not v.startswith("X")]
            S = np.loadtxt(os.path.join(run_dir, "step%d" % i))
            dim = sum(3 if v.startswith("X") else 2 for v in order)
            assert S.shape == (500, dim) and np.all(np.isfinite(S))
            loss = json.load(open(os.path.join(run_dir, "step%d_step_training_loss" % i)))
            assert 1 <= len(loss) <= 8
            for l in loss.values():
                l = np.array(l)
                it = int(np.count_nonzero(l))
                assert len(l) == args["flow_iterations"] and it % 50 == 0 and it >= 100 and np.all(l[it:] == 0)
                assert np.all(np.isfinite(l)) and l[it - 1] < l[0]
        r, n_pose = _trajectory_rmse(solver, truth)
        assert n_pose == len([v for v in seen if v.startswith("X")])
        rmses.append(r)
    assert max(rmses) < rmse_each and float(np.median(rmses)) < rmse_median, rmses


# ---------------------------------------------------------------------------------------------------------
# config[2] as a REAL graph: toy range-only SLAM in R2 (20 poses, 4 landmarks) with the R2 factor family
# ---------------------------------------------------------------------------------------------------------
def _toy_r2_graph(seed=0):
    """The generative recipe of example/slam/toy_examples/R2RangeGaussian_example (R2 poses on a path, displacement
    odometry, range factors to R2 landmarks, Gaussian prior on the first pose) at BASELINE config[2]'s size."""
    from factors.Factors import (R2RangeGaussianLikelihoodFactor, R2RelativeGaussianLikelihoodFactor,
                                 UnaryR2GaussianPriorFactor)
    from slam.Variables import R2Variable, VariableType
    rng = np.random.RandomState(seed)
    poses = [R2Variable("X%d" % i) for i in range(20)]
    lms = [R2Variable("L%d" % j, VariableType.Landmark) for j in range(4)]
    lm_xy = np.array([[15.0, 20.0], [45.0, -15.0], [75.0, 25.0], [100.0, -10.0]])
    xy = np.stack([np.array([6.0 * i, 8.0 * np.sin(i / 3.0)]) for i in range(20)])
    truth = {v: xy[i] for i, v in enumerate(poses)}
    truth.update({v: lm_xy[j] for j, v in enumerate(lms)})
    factors = [UnaryR2GaussianPriorFactor(poses[0], xy[0], covariance=np.eye(2) * 0.04)]
    odo_cov = np.eye(2) * 0.04
    for i in range(19):
        factors.append(R2RelativeGaussianLikelihoodFactor(poses[i], poses[i + 1], xy[i + 1] - xy[i] + 0.2 * rng.randn(2),
                                                          covariance=odo_cov))
    for i in range(20):
        j = int(np.argmin(np.linalg.norm(lm_xy - xy[i], axis=1)))
        for jj in {j, (j + 1) % 4 if i % 3 == 0 else j}:
            factors.append(R2RangeGaussianLikelihoodFactor(poses[i], lms[jj], float(np.linalg.norm(lm_xy[jj] - xy[i]) +
                                                                                     0.5 * rng.randn()), 0.5))
    return poses + lms, truth, factors


def test_toy_r2_range_only_graph_runs_on_the_device_simulator(tmp_path):
    """BASELINE config[2] ("toy_examples range-only SLAM, 20 poses / 4 landmarks") as a factor graph: `.fg` text round trip,
    incremental updates through NFiSAM with every clique batch simulated by `nfisam_simulate_clique` (R2 ops 11-15),
    posterior means near the truth once every landmark has been ranged from several poses."""
    import sampler.DeviceSimulation as DS
    from slam.FactorGraphSimulator import factor_graph_to_string, read_factor_graph_from_file
    from slam.NFiSAM import NFiSAM, NFiSAMArgs
    from slam.RunBatch import group_nodes_factors_incrementally
    nodes, truth, factors = _toy_r2_graph()
    p = tmp_path / "factor_graph.fg"
    p.write_text(factor_graph_to_string(nodes, factors, truth))
    nodes2, truth2, factors2 = read_factor_graph_from_file(str(p))
    assert [str(f) for f in factors2] == [str(f) for f in factors] and len(nodes2) == 24
    steps = group_nodes_factors_incrementally(nodes2, factors2, incremental_step=5)
    assert len(steps) == 4
    unsupported = []
    orig = DS.FusedSimulationBackend.run_plan

    def spy(self, steps_, pattern, n):
        try:
            return orig(self, steps_, pattern, n)
        except DS.DeviceSimulationUnsupported as e:      # would silently fall back to the host samplers
            unsupported.append(str(e))
            raise
    DS.FusedSimulationBackend.run_plan = spy
    try:
        random.seed(1); np.random.seed(1); torch.manual_seed(1)
        solver = NFiSAM(NFiSAMArgs(num_knots=9, flow_iterations=1500, local_sample_num=2000, learning_rate=.02,
                                   hidden_dim=8, cuda_training=True, elimination_method="pose_first",
                                   training_set_frac=1.0, loss_delta_tol=.01, posterior_sample_num=500))
        trained = 0
        for vs, fs in steps:
            for v in vs:
                solver.add_node(v)
            for f in fs:
                solver.add_factor(f)
            solver.update_physical_and_working_graphs()
            res = solver.incremental_inference()
            trained += len(solver._temp_training_loss)
    finally:
        DS.FusedSimulationBackend.run_plan = orig
    assert unsupported == []
    assert trained >= 8                                   # "8 cliques" of config[2]: at least that many flows were trained
    name = {str(v.name): v for v in solver.physical_vars}
    err = np.array([res[name["X%d" % i]].mean(0) - truth2[name["X%d" % i]] for i in range(20)])
    assert np.sqrt((err ** 2).sum(1).mean()) < 1.5, np.sqrt((err ** 2).sum(1).mean())
    near = 0
    for j in range(4):
        s = res[name["L%d" % j]]
        assert s.shape == (500, 2) and np.all(np.isfinite(s))
        # range-only landmark posteriors are arcs or two mirrored modes (> 40 m apart): the truth must carry weight (a
        # twentieth of the samples within 6 m; over 6 solver seeds the true mode of L2 holds 8-58 % with either kernel
        # family, scripts/exp/toy_r2_seeds.py, the other landmarks 95-100 %) ...
        d = np.linalg.norm(s - truth2[name["L%d" % j]], axis=1)
        assert np.quantile(d, 0.05) < 6.0, (j, np.median(s, 0), np.quantile(d, 0.05))
        near += int(np.linalg.norm(np.median(s, 0) - truth2[name["L%d" % j]]) < 8.0)
    assert near >= 3, near                                  # ... and at least three of the four are resolved (median within 8 m)
