"""GPU tests of the drop-in module surface (flows.*, slam.NFiSAM): the reference's own usage
patterns — module forward/inverse, `loss.backward()` + `torch.optim.Adam` (NFiSAM.py:425,469-475),
the solver hooks — must give the reference's numbers (golden vectors) through the HIP kernels."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda:0"


def load(name):
    g = dict(np.load(os.path.join(GOLDEN, name)))
    n, D, K, H, seed = [int(v) for v in g["meta"]]
    return g, n, D, K, H, float(g["B"])


def golden_flow(g, D, K, H, prefix="p0"):
    from flows.flows import NSF_AR
    f = NSF_AR(dim=D, K=K, hidden_dim=H)
    f.load_state_dict({k[len(prefix) + 2:].replace("__", "."): torch.tensor(v) for k, v in g.items()
                       if k.startswith(prefix + "__")})
    return f.to(DEV)


@pytest.mark.parametrize("name", ["nsf_n128_d6_k9.npz", "nsf_n64_d2_k5.npz", "nsf_n128_d16_k12.npz"])
def test_module_forward_inverse_autograd(name):
    from flows.models import NormalizingFlowModel
    from flows.prior_dist import CustomMultivariateNormal
    g, n, D, K, H, B = load(name)
    flow = golden_flow(g, D, K, H)
    x = torch.tensor(g["x"]).to(DEV)
    with torch.no_grad():
        z, ld = flow(x)
        np.testing.assert_allclose(z.cpu().numpy(), g["z"], atol=1e-4)
        np.testing.assert_allclose(ld.cpu().numpy(), g["logdet"], atol=2e-4)
        xi, ldi = flow.inverse(torch.tensor(g["zlat"]).to(DEV))
        np.testing.assert_allclose(xi.cpu().numpy(), g["zlat_inv_x"], atol=2e-4)
        np.testing.assert_allclose(ldi.cpu().numpy(), g["zlat_inv_logdet"], atol=3e-4)
        if "igs3_x" in g:
            xf = flow.inverse_given_separator(torch.tensor(g["zlat"][:, 3:]).to(DEV), x[:, :3])
            np.testing.assert_allclose(xf.cpu().numpy(), g["igs3_x"], atol=2e-4)
    # the reference's training step, verbatim: model(x) -> loss -> backward -> Adam.step
    model = NormalizingFlowModel(CustomMultivariateNormal(D, device=DEV), [flow])
    opt = torch.optim.Adam(model.parameters(), lr=float(g["adam_lr"]))
    losses = []
    for it in range(2):
        opt.zero_grad()
        zz, plp, ldd = model(x)
        loss = -torch.mean(plp + ldd)
        losses.append(loss.item())
        loss.backward()
        if it == 0:
            for k, p in flow.named_parameters():
                ref = g["g0__" + k.replace(".", "__")]
                np.testing.assert_allclose(p.grad.cpu().numpy(), ref, rtol=1e-3,
                                           atol=2e-5 * max(1.0, float(np.abs(ref).max())))
        opt.step()
    np.testing.assert_allclose(losses, g["adam_losses"][:2], atol=5e-4)
    for k, p in flow.named_parameters():
        np.testing.assert_allclose(p.detach().cpu().numpy(), g["p2__" + k.replace(".", "__")], atol=4e-4, rtol=1e-3)
    # no-grad model forward uses the fused L-layer kernel: same numbers
    with torch.no_grad():
        z2, plp2, ld2 = model(x)
        z3, ld3 = flow(x)
        assert torch.allclose(z2, z3, atol=1e-6) and torch.allclose(ld2, ld3, atol=1e-5)
        assert torch.allclose(plp2, CustomMultivariateNormal(D, device=DEV).log_prob(z2), atol=1e-4)


def test_reference_scramble_option_reproduces_reference_return_value():
    g, n, D, K, H, B = load("nsf_n128_d6_k9.npz")
    flow = golden_flow(g, D, K, H)
    flow.reference_scramble = True
    with torch.no_grad():
        z, _ = flow(torch.tensor(g["x"]).to(DEV))
    np.testing.assert_allclose(z.cpu().numpy(), g["z_raw"], atol=1e-4)


def test_utils_rqs_against_reference_golden():
    from flows.utils import RQS, unconstrained_RQS
    g = dict(np.load(os.path.join(GOLDEN, "rqs_direct.npz")))
    for tag in "abcd":
        W, Hh, Dd, inp = (torch.tensor(g["%s_%s" % (tag, k)]).to(DEV) for k in ("W", "H", "D", "inp"))
        tb = float(g[tag + "_tb"])
        y, ld = unconstrained_RQS(inp, W, Hh, Dd, inverse=False, tail_bound=tb)
        np.testing.assert_allclose(y.cpu().numpy(), g[tag + "_fwd"], atol=1e-4)
        # random N(0,1.5) logits produce bins down to the 1e-3 minimum width: the position inside such
        # a bin carries the fp32 rounding of the cumulative knots (in the reference too), which the
        # log-derivative amplifies -> 1e-3 on log-dets here, 2e-4 on the flow-level tests
        np.testing.assert_allclose(ld.cpu().numpy(), g[tag + "_fwd_ld"], atol=1e-3)
        xi, ldi = unconstrained_RQS(inp, W, Hh, Dd, inverse=True, tail_bound=tb)
        np.testing.assert_allclose(xi.cpu().numpy(), g[tag + "_inv"], atol=1e-4)
        np.testing.assert_allclose(ldi.cpu().numpy(), g[tag + "_inv_ld"], atol=1e-3)
    W, Hh, Dd, inp = (torch.tensor(g["rqs_" + k]).to(DEV) for k in ("W", "H", "D", "inp"))
    y, ld = RQS(inp, W, Hh, Dd, inverse=False)
    np.testing.assert_allclose(y.cpu().numpy(), g["rqs_fwd"], atol=2e-5)
    np.testing.assert_allclose(ld.cpu().numpy(), g["rqs_fwd_ld"], atol=2e-4)
    xi, ldi = RQS(inp, W, Hh, Dd, inverse=True)
    np.testing.assert_allclose(xi.cpu().numpy(), g["rqs_inv"], atol=2e-5)
    np.testing.assert_allclose(ldi.cpu().numpy(), g["rqs_inv_ld"], atol=3e-4)
    with pytest.raises(ValueError):
        RQS(inp + 2.0, W, Hh, Dd)        # "Input outside domain" (utils.py:74-76)


class FakeClique:
    """Duck-typed BayesTreeNode: what fit_clique_density_model reads (frontal_dim, vars)."""

    def __init__(self, frontal, separator):
        self.frontal, self.separator = frontal, separator
        self.vars = set(frontal) | set(separator)
        self.frontal_dim = sum(v.dim for v in frontal)
        self.separator_dim = sum(v.dim for v in separator)
        self.dim = self.frontal_dim + self.separator_dim


def ring_clique(n, rng):
    """[obs | landmark (sep) | pose (frontal)]: range measurement between a pose and a landmark."""
    pose = rng.randn(n, 3) * np.array([0.3, 0.3, 0.2]) + np.array([1.0, -2.0, 3.0])   # theta near +-pi: wraps
    pose[:, 2] = (pose[:, 2] + np.pi) % (2 * np.pi) - np.pi
    r = 8.0 + 0.5 * rng.randn(n)
    phi = rng.uniform(-np.pi, np.pi, n)
    lm = pose[:, :2] + np.stack([r * np.cos(phi), r * np.sin(phi)], 1)
    obs = np.linalg.norm(lm - pose[:, :2], axis=1, keepdims=True) + 0.5 * rng.randn(n, 1)
    return np.concatenate([obs, lm, pose], 1)


def mmd_rbf(a, b, sigma):
    def k(x, y):
        d = ((x[:, None, :] - y[None, :, :]) ** 2).sum(-1)
        return np.exp(-d / (2 * sigma ** 2))
    return float(np.sqrt(max(k(a, a).mean() + k(b, b).mean() - 2 * k(a, b).mean(), 0)))


def test_fit_clique_density_model_and_conditional_sampling():
    from slam.NFiSAM import NFiSAM, NFiSAMArgs, FlowsPriorFactor
    from slam.Variables import R2Variable, SE2Variable, VariableType
    rng = np.random.RandomState(0)
    np.random.seed(0); torch.manual_seed(0)
    n = 2000
    samples = ring_clique(n, rng)
    L0, X0 = R2Variable("L0", VariableType.Landmark), SE2Variable("X0")
    clique = FakeClique(frontal=[X0], separator=[L0])
    solver = NFiSAM(NFiSAMArgs(flow_iterations=600, num_knots=9, learning_rate=0.02, local_sample_num=n,
                               average_window=50, loss_delta_tol=1e-2))
    timer = []
    keep = samples.copy()
    model = solver.fit_clique_density_model(clique, samples, [L0, X0], timer)
    np.testing.assert_array_equal(samples, keep)
    iters = solver.last_fit_iterations
    # the loss curve stays on the device until somebody reads the reference's `_temp_training_loss` (one copy for all fits
    # recorded since the last look); what the attribute hands out is the reference's plain {clique name: list of floats}
    assert len(solver.__dict__["_loss_pending"]) == 1 and solver.__dict__["_loss_record"] == {}
    record = solver._temp_training_loss
    assert type(record) is dict and solver.__dict__["_loss_pending"] == [] and record is solver._temp_training_loss
    assert all(isinstance(v, list) and isinstance(v[0], float) for v in record.values())
    loss = np.array(solver._temp_training_loss["".join(str(v.name) for v in clique.vars)])
    assert len(loss) == 600 and iters % 50 == 0 and 100 <= iters <= 600
    assert np.all(loss[iters:] == 0) and loss[iters - 1] < loss[0] - 0.3
    assert len(timer) == 1 and timer[0] > 0
    assert model.dim == 6 and model.separator_dim == 3
    # unconditional joint samples reproduce the training distribution (MMD with RBF sigma = sqrt(dim))
    xs = model.conditional_sample_given_observation(conditional_dim=6, sample_number=1500)
    assert xs.shape == (1500, 6) and np.all(np.isfinite(xs))
    assert np.all(np.abs(xs[:, 5]) <= np.pi + 1e-5)
    ref = samples[:1500]
    floor = mmd_rbf(samples[:1000, 1:5], samples[1000:2000, 1:5], 2.0)
    got = mmd_rbf(xs[:1000, 1:5], ref[:1000, 1:5], 2.0)
    assert got < max(0.08, 3 * floor), (got, floor)
    # conditional: fix obs + landmark, sample the pose; ring geometry must hold: |lm - pose| ~ obs
    obs_lm = np.tile(np.array([[8.0, 9.0, -2.0]]), (800, 1))
    pose = model.conditional_sample_given_observation(conditional_dim=3, obs_samples=obs_lm)
    assert pose.shape == (800, 3)
    d = np.linalg.norm(obs_lm[:, 1:3] - pose[:, :2], axis=1)
    assert abs(np.median(d) - 8.0) < 1.0, np.median(d)
    # separator factor = child->parent message: samples of the landmark given the true observation
    fac = solver.clique_density_to_separator_factor([L0], model, np.array([8.0]))
    assert isinstance(fac, FlowsPriorFactor) and fac.dim == 2 and fac.circular_dim_list == [False, False]
    lm = fac.sample(500)
    assert lm.shape == (500, 2) and np.all(np.isfinite(lm))
    # log_pdf / grad_x_log_pdf consistency (finite differences on the device model)
    x0 = lm[:5].astype(np.float64)
    lp = fac.log_pdf(x0)
    g = fac.grad_x_log_pdf(x0)
    eps = 1e-2
    for c in range(2):
        xp = x0.copy(); xp[:, c] += eps
        xm = x0.copy(); xm[:, c] -= eps
        # the log-density of a rational-quadratic spline flow is C1 in x only between knots (the slope of
        # log dz/dx jumps at a knot), so the analytic gradient has to lie between the two one-sided differences
        fdp = (fac.log_pdf(xp) - lp) / eps
        fdm = (lp - fac.log_pdf(xm)) / eps
        tol = 5e-2 + 5e-2 * np.abs(g[:, c])
        assert np.all(g[:, c] >= np.minimum(fdp, fdm) - tol) and np.all(g[:, c] <= np.maximum(fdp, fdm) + tol), \
            (g[:, c], fdp, fdm)
    assert lp.shape == (5,)
    # model reuse: same variables, new split
    solver._clique_density_model[clique] = model
    new_clique = FakeClique(frontal=[L0, X0], separator=[])
    m2 = solver.root_clique_density_model_to_leaf(clique, new_clique)
    assert m2.separator_dim == 1 and m2.flows[0] is model.flows[0]


@pytest.mark.parametrize("hidden", [6, 12])
def test_hidden_dims_of_the_references_own_grid_through_the_solver_surface(hidden):
    """`hidden_dim` 6 / 12 (the reference's grid, example/slam/manhattan_world_with_range/lawnmower_4x4/run_nfisam.py:5-6) through
    the drop-in surface: `NFiSAM(NFiSAMArgs(hidden_dim=...))` fits a clique and samples from it (zero-padded kernels of width 8 /
    16 underneath, ABI 1500), the returned `NSF_AR` modules carry the REFERENCE'S shapes in `state_dict()` -- `network.0.weight`
    [hidden, i], `network.2.weight` [hidden, hidden], `network.4.weight` [3K - 1, hidden] -- and a module rebuilt from that
    state_dict gives the same forward values."""
    from flows.flows import NSF_AR
    from slam.NFiSAM import NFiSAM, NFiSAMArgs
    from slam.Variables import R2Variable, SE2Variable, VariableType
    rng = np.random.RandomState(1)
    np.random.seed(1); torch.manual_seed(1)
    n, K = 2000, 9
    samples = ring_clique(n, rng)
    L0, X0 = R2Variable("L0", VariableType.Landmark), SE2Variable("X0")
    clique = FakeClique(frontal=[X0], separator=[L0])
    solver = NFiSAM(NFiSAMArgs(flow_iterations=400, num_knots=K, hidden_dim=hidden, learning_rate=0.02, local_sample_num=n,
                               average_window=50, loss_delta_tol=1e-2))
    model = solver.fit_clique_density_model(clique, samples, [L0, X0], [])
    loss = np.array(solver._temp_training_loss["".join(str(v.name) for v in clique.vars)])
    iters = solver.last_fit_iterations
    assert 100 <= iters <= 400 and loss[iters - 1] < loss[0] - 0.3 and np.all(np.isfinite(loss))
    xs = model.conditional_sample_given_observation(conditional_dim=6, sample_number=1500)
    assert xs.shape == (1500, 6) and np.all(np.isfinite(xs))
    floor = mmd_rbf(samples[:1000, 1:5], samples[1000:2000, 1:5], 2.0)
    assert mmd_rbf(xs[:1000, 1:5], samples[:1000, 1:5], 2.0) < max(0.08, 3 * floor)
    flow = model.flows[0]
    sd = flow.state_dict()
    assert tuple(sd["init_param"].shape) == (3 * K - 1,)
    for j in range(5):                                        # conditioner of dim i = j + 1
        assert tuple(sd["layers.%d.network.0.weight" % j].shape) == (hidden, j + 1)
        assert tuple(sd["layers.%d.network.0.bias" % j].shape) == (hidden,)
        assert tuple(sd["layers.%d.network.2.weight" % j].shape) == (hidden, hidden)
        assert tuple(sd["layers.%d.network.4.weight" % j].shape) == (3 * K - 1, hidden)
    twin = NSF_AR(6, K=K, B=5.0, hidden_dim=hidden).to("cuda:0")
    twin.load_state_dict({k: v.to("cuda:0") for k, v in sd.items()})
    x = torch.from_numpy(np.clip(np.random.RandomState(2).randn(200, 6), -3, 3).astype(np.float32)).to("cuda:0")
    z1, l1 = flow.forward(x)
    z2, l2 = twin.forward(x)
    np.testing.assert_allclose(z1.detach().cpu().numpy(), z2.detach().cpu().numpy(), atol=1e-5)
    np.testing.assert_allclose(l1.detach().cpu().numpy(), l2.detach().cpu().numpy(), atol=1e-5)


def test_fit_with_validation_split_and_multilayer():
    from slam.NFiSAM import NFiSAM, NFiSAMArgs
    from slam.Variables import R2Variable, SE2Variable, VariableType
    rng = np.random.RandomState(1)
    np.random.seed(1)
    samples = ring_clique(1500, rng)
    L0, X0 = R2Variable("L0", VariableType.Landmark), SE2Variable("X0")
    clique = FakeClique(frontal=[X0], separator=[L0])
    solver = NFiSAM(NFiSAMArgs(flow_iterations=120, num_knots=9, learning_rate=0.02, flow_number=2,
                               training_set_frac=0.8, validation_interval=10))
    model = solver.fit_clique_density_model(clique, samples, [L0, X0], None)
    loss = np.array(solver._temp_training_loss["".join(str(v.name) for v in clique.vars)])
    it = solver.last_fit_iterations
    assert 10 <= it <= 120 and loss[it - 1] < loss[0]
    assert len(model.flows) == 2
    # the rule ran on the device (tests/test_hip_parity.py holds it to the reference's own loop): every evaluation is on
    # record, and a run that ended early ended where the rule scheduled it
    vals = solver.last_validation_losses.cpu().numpy()
    n_eval = int(np.count_nonzero(vals))
    assert 1 <= n_eval <= it // 10 + 1 and np.all(np.isfinite(vals))
    if it < 120:
        first_rise = next(k for k in range(1, n_eval) if vals[k] > vals[:k].min() and vals[k] > vals[k - 1])
        assert it == 2 * 10 * (first_rise + 1) - 1, (it, vals[:n_eval])
    # a rate the device rule does not cover is stepped from the host: same interface, same kind of result
    solver15 = NFiSAM(NFiSAMArgs(flow_iterations=60, num_knots=9, learning_rate=0.02, training_set_frac=0.8, validation_interval=10,
                                 slower_stop_rate=1.5))
    solver15.fit_clique_density_model(clique, samples, [L0, X0], None)
    assert 10 <= solver15.last_fit_iterations <= 60
    xs = model.conditional_sample_given_observation(conditional_dim=6, sample_number=64)
    assert xs.shape == (64, 6) and np.all(np.isfinite(xs))
    with pytest.raises(NotImplementedError):
        NFiSAM(NFiSAMArgs(flow_type="NSF_AR_CS")).fit_clique_density_model(clique, samples, [L0, X0], None)


def test_multilayer_fit_through_the_training_plan_on_both_multilayer_kernels():
    """`flow_number = 3` (src/slam/NFiSAM.py:387-390) with the full batch as the training set: the fit runs as a training plan of
    the two-dims-per-wave kernel (D = 6: its panel image and parked forward state come into play from the second iteration
    of a chunk on) and, with `NFISAM_PAIR=0`, of the split kernel.  Same early-stop rule on the same batch: the loss records
    stay within rounding-amplified noise of each other, both fits pass the window rule, and the model samples."""
    from slam.NFiSAM import NFiSAM, NFiSAMArgs
    from slam.Variables import R2Variable, SE2Variable, VariableType
    samples = ring_clique(2000, np.random.RandomState(3))
    L0, X0 = R2Variable("L0", VariableType.Landmark), SE2Variable("X0")
    clique = FakeClique(frontal=[X0], separator=[L0])
    res = {}
    for pair in ("0", None):
        old = os.environ.get("NFISAM_PAIR")
        if pair is None:
            os.environ.pop("NFISAM_PAIR", None)
        else:
            os.environ["NFISAM_PAIR"] = pair
        try:
            np.random.seed(5); torch.manual_seed(5)
            solver = NFiSAM(NFiSAMArgs(flow_iterations=400, num_knots=9, learning_rate=0.01, flow_number=3, loss_delta_tol=0.01))
            model = solver.fit_clique_density_model(clique, samples, [L0, X0], None)
            loss = np.array(solver._temp_training_loss["".join(str(v.name) for v in clique.vars)])
            it = solver.last_fit_iterations
            xs = model.conditional_sample_given_observation(conditional_dim=6, sample_number=64)
            res[pair] = (it, loss[:it].copy(), xs)
        finally:
            if old is None:
                os.environ.pop("NFISAM_PAIR", None)
            else:
                os.environ["NFISAM_PAIR"] = old
    for it, loss, xs in res.values():
        assert 100 <= it <= 400 and it % 50 == 0 and np.all(np.isfinite(loss)) and loss[-1] < loss[0] - 1.0
        assert len(model.flows) == 3 and xs.shape == (64, 6) and np.all(np.isfinite(xs))
    a, b = res["0"][1], res[None][1]
    np.testing.assert_allclose(a[:5], b[:5], atol=2e-3)                # same initialisation, same first steps
    m = min(len(a), len(b))
    assert abs(a[m - 1] - b[m - 1]) < 0.15, (a[m - 1], b[m - 1])       # (trajectories drift apart by rounding, the fits agree)


def test_device_normalisation_matches_reference_golden():
    """nfisam_normalize_columns (f-3) against the vectors produced by the reference's
    NFiSAM.normalize_training_samples (tests/golden/make_golden.py), float32 input."""
    import nfisam_hip as nh
    g = dict(np.load(os.path.join(GOLDEN, "normalize.npz")))
    x = torch.from_numpy(g["samples"].astype(np.float32)).to(DEV)
    xn, mu, sd = nh.normalize_columns(x, g["circular"])
    np.testing.assert_allclose(mu.cpu().numpy(), g["mean"], atol=2e-6, rtol=2e-6)
    np.testing.assert_allclose(sd.cpu().numpy(), g["std"], atol=2e-6, rtol=2e-6)
    # column 5 of the fixture is constant up to 1e-7 (std clipped at 1e-5): dividing the float32 rounding of the
    # INPUT by 1e-5 makes it incomparable; every other column agrees to 2e-5
    live = g["std"] > 1e-4
    np.testing.assert_allclose(xn.cpu().numpy()[:, live], g["train_norm"][:, live], atol=2e-5)
    assert np.abs(xn.cpu().numpy()[:, ~live]).max() < 0.1
    # all-Euclidean call, constant column -> std clipped at 1e-5, in-place allowed
    y = torch.cat([x[:, :2], torch.full((x.shape[0], 1), 3.0, device=DEV)], 1).contiguous()
    yn, m2, s2 = nh.normalize_columns(y)
    assert abs(float(s2[2]) - 1e-5) < 1e-9 and float(yn[:, 2].abs().max()) == 0.0 and abs(float(m2[2]) - 3.0) < 1e-6


@pytest.mark.parametrize("backend_name", ["FusedSimulationBackend"])
def test_device_batch_simulator_matches_host_simulator_in_distribution(backend_name):
    """sampler.DeviceSimulation (f-2: one fused kernel per clique) draws the same joint
    distribution as the factors' numpy samplers: a clique with
    an SE(2) prior, two odometry steps, range factors to two landmarks (one creates the landmark on a ring, the later
    ones become simulated-observation columns), a 2-way ambiguous association and a possibly-outlier range."""
    from factors.Factors import (AmbiguousDataAssociationFactor, BinaryFactorWithNullHypo,
                                 SE2R2RangeGaussianLikelihoodFactor, SE2RelativeGaussianLikelihoodFactor,
                                 UnarySE2ApproximateGaussianPriorFactor)
    import sampler.DeviceSimulation as DS
    from sampler.SimulationBasedSampler import SimulationBasedSampler
    from slam.Variables import R2Variable, SE2Variable, VariableType
    X = [SE2Variable("X%d" % i) for i in range(3)]
    L0, L1 = R2Variable("L0", VariableType.Landmark), R2Variable("L1", VariableType.Landmark)
    odom_cov = np.diag([0.2, 0.04, 0.02]) ** 2
    fs = [UnarySE2ApproximateGaussianPriorFactor(X[0], np.array([1.0, -2.0, 0.3]), np.diag([0.3, 0.2, 0.05]) ** 2),
          SE2RelativeGaussianLikelihoodFactor(X[0], X[1], np.array([5.0, 0.5, 0.4]), odom_cov),
          SE2RelativeGaussianLikelihoodFactor(X[1], X[2], np.array([5.0, -0.5, -0.2]), odom_cov),
          SE2R2RangeGaussianLikelihoodFactor(X[0], L0, 12.0, 0.5),
          SE2R2RangeGaussianLikelihoodFactor(X[1], L1, 9.0, 0.5),
          SE2R2RangeGaussianLikelihoodFactor(X[2], L0, 11.0, 0.5),
          AmbiguousDataAssociationFactor(X[2], [L0, L1], np.array([0.5, 0.5]), SE2R2RangeGaussianLikelihoodFactor, 10.0,
                                         0.5),
          BinaryFactorWithNullHypo(X[1], L0, np.array([0.7, 0.3]), SE2R2RangeGaussianLikelihoodFactor, 11.5, 0.5,
                                   null_sigma_scale=6.0)]
    order = [L0, L1] + X
    n = 6000
    np.random.seed(0); torch.manual_seed(0)
    host, hv, hobs = SimulationBasedSampler(fs, order).sample(n)
    dev_s, dv, dobs = SimulationBasedSampler(fs, order).sample(n, backend=getattr(DS, backend_name)(DEV))
    assert dev_s.is_cuda and dev_s.dtype == torch.float32 and tuple(dev_s.shape) == host.shape
    assert [str(v.name) for v in hv] == [str(v.name) for v in dv]
    np.testing.assert_array_equal(hobs, dobs)
    d = dev_s.cpu().numpy().astype(np.float64)
    # moments column by column (angles: compare resultant vectors), then a joint two-sample statistic
    for c in range(host.shape[1]):
        sd = host[:, c].std()
        assert abs(host[:, c].mean() - d[:, c].mean()) < 0.08 * sd + 5e-3, (c, host[:, c].mean(), d[:, c].mean())
        assert abs(d[:, c].std() / sd - 1.0) < 0.06, (c, sd, d[:, c].std())
    # (the numpy sampler assigns mixture components to contiguous row blocks: shuffle before taking subsets)
    perm = np.random.RandomState(1).permutation(n)
    host, d = host[perm], d[perm]
    scale = host.std(0)
    a, b = host[:1500] / scale, d[:1500] / scale
    floor = mmd_rbf(host[:1500] / scale, host[1500:3000] / scale, np.sqrt(host.shape[1]))
    assert mmd_rbf(a, b, np.sqrt(host.shape[1])) < max(0.03, 3 * floor)


def test_fused_simulator_ops_are_exact_without_noise():
    """With (almost) no noise the simulation is deterministic apart from ring bearings: every SE(2) op of the fused kernel
    (prior, odometry forward and backward, odometry as a measurement, simulated range) must reproduce the host
    factors' numpy algebra."""
    from factors.Factors import (SE2R2RangeGaussianLikelihoodFactor, SE2RelativeGaussianLikelihoodFactor,
                                 UnarySE2ApproximateGaussianPriorFactor)
    from sampler.DeviceSimulation import FusedSimulationBackend
    from sampler.SimulationBasedSampler import SimulationBasedSampler
    from slam.Variables import R2Variable, SE2Variable, VariableType
    X = [SE2Variable("X%d" % i) for i in range(4)]
    L0 = R2Variable("L0", VariableType.Landmark)
    tiny = np.diag([1e-18, 1e-18, 1e-18])
    fs = [UnarySE2ApproximateGaussianPriorFactor(X[1], np.array([3.0, -1.0, 2.8]), tiny),
          SE2RelativeGaussianLikelihoodFactor(X[1], X[2], np.array([4.0, 0.7, 1.1]), tiny),      # forward: X2 from X1
          SE2RelativeGaussianLikelihoodFactor(X[0], X[1], np.array([2.0, -0.5, -2.9]), tiny),    # backward: X0 from X1
          SE2RelativeGaussianLikelihoodFactor(X[2], X[3], np.array([1.0, 1.0, 3.0]), tiny),
          SE2RelativeGaussianLikelihoodFactor(X[0], X[3], np.array([0.0, 0.0, 0.0]), tiny),      # both ends drawn: measurement
          SE2R2RangeGaussianLikelihoodFactor(X[2], L0, 6.0, 1e-9),                               # ring (random bearing)
          SE2R2RangeGaussianLikelihoodFactor(X[0], L0, 5.0, 1e-9)]                               # simulated range
    order = [L0] + X
    n = 300
    np.random.seed(1)
    host, hv, _ = SimulationBasedSampler(fs, order).sample(n)
    devb, dv, _ = SimulationBasedSampler(fs, order).sample(n, backend=FusedSimulationBackend(DEV))
    d = devb.cpu().numpy().astype(np.float64)
    assert [str(v.name) for v in hv] == [str(v.name) for v in dv] and host.shape == d.shape == (n, 4 + 2 + 12)
    names = [str(v.name) for v in hv]
    # columns: [odometry measurement (3) | range measurement (1) | L0 (2) | X0..X3 (3 each)]
    pose_cols = slice(6, 18)
    ang = [8, 11, 14, 17]
    lin = [c for c in range(6, 18) if c not in ang]
    np.testing.assert_allclose(d[:, lin], host[:, lin], atol=2e-5)
    dth = np.abs((d[:, ang] - host[:, ang] + np.pi) % (2 * np.pi) - np.pi)
    assert dth.max() < 2e-6
    # the odometry measurement between X0 and X3 (deterministic), angle compared modulo 2 pi
    np.testing.assert_allclose(d[:, 0:2], host[:, 0:2], atol=5e-5)
    assert np.abs((d[:, 2] - host[:, 2] + np.pi) % (2 * np.pi) - np.pi).max() < 5e-6
    # the landmark sits on the ring of radius 6 around X2 (bearing random), and the simulated range is its distance to X0
    x2, x0 = d[:, 12:14], d[:, 6:8]
    np.testing.assert_allclose(np.linalg.norm(d[:, 4:6] - x2, axis=1), 6.0, atol=1e-4)
    np.testing.assert_allclose(d[:, 3], np.linalg.norm(d[:, 4:6] - x0, axis=1), atol=1e-4)
    phi = np.arctan2(d[:, 5] - x2[:, 1], d[:, 4] - x2[:, 0])
    assert phi.min() < -2.5 and phi.max() > 2.5 and abs(np.mean(np.cos(phi))) < 0.2


def test_fused_simulator_ops_match_reference_se2_algebra():
    """The SE(2) ops of `nfisam_simulate_clique` (csrc/clique_sim.hip) against vectors produced by the REFERENCE's factor
    `sample` bodies with zero noise (tests/golden/se2_factors.npz, src/factors/Factors.py:1196-1317, 2610-2619 executed by
    tests/golden/make_golden.py): odometry forward T_j = T_i * obs, backward T_i = T_j / obs, odometry as a measurement
    T_i^-1 T_j, simulated range, SE(2) prior.  The op list is written by hand so that every op gets the golden inputs."""
    import nfisam_hip as nh
    g = dict(np.load(os.path.join(GOLDEN, "se2_factors.npz")))
    n = g["f_x1"].shape[0]
    src = torch.from_numpy(np.hstack([g["f_x1"], g["f_x2"], g["f_lm"]]).astype(np.float32)).to(DEV).contiguous()

    def op(code, a=0, b=0, c=0, k=0, p=(), srcp=0):
        o = nh.SimOp()
        o.code, o.a, o.b, o.c, o.k, o.src = code, a, b, c, k, srcp
        for i, v in enumerate(p):
            o.p[i] = float(v)
        return o
    z6 = [0.0] * 6
    obs = list(g["rel_obs_value"])
    ops = [op(nh.SIM_COPY, a=8, b=0, c=4, k=3, srcp=src.data_ptr()), op(nh.SIM_COPY, a=8, b=3, c=7, k=3, srcp=src.data_ptr()),
           op(nh.SIM_COPY, a=8, b=6, c=10, k=2, srcp=src.data_ptr()),
           op(nh.SIM_REL_FWD, a=4, c=12, p=obs + z6), op(nh.SIM_REL_BWD, a=7, c=15, p=obs + z6),
           op(nh.SIM_REL_OBS, a=4, b=7, c=0, p=[0, 0, 0] + z6), op(nh.SIM_RANGE_OBS, a=4, b=10, c=3, p=[0.0]),
           op(nh.SIM_PRIOR_SE2, c=18, p=list(g["prior_pose"]) + z6)]
    out = nh.simulate_clique(ops, n, 21, 21, 12345, DEV).cpu().numpy().astype(np.float64)

    def pose_close(got, ref):
        np.testing.assert_allclose(got[:, :2], ref[:, :2], atol=4e-5)
        d = np.abs((got[:, 2] - ref[:, 2] + np.pi) % (2 * np.pi) - np.pi)
        assert np.minimum(d, 2 * np.pi - d).max() < 5e-6
    np.testing.assert_allclose(out[:, 4:12], src.cpu().numpy(), atol=0)
    pose_close(out[:, 12:15], g["rel_fwd0"])
    pose_close(out[:, 15:18], g["rel_bwd0"])
    pose_close(out[:, 0:3], g["rel_meas0"])
    np.testing.assert_allclose(out[:, 3], g["range_meas0"][:, 0], atol=4e-5)
    pose_close(out[:, 18:21], np.tile(g["prior_pose"], (n, 1)))


def test_fused_simulator_mixture_ops_match_reference_statistics():
    """k-way association (`SIM_ADA_OBS`) and null-hypothesis (`SIM_NH_OBS`, `SIM_NH_RING`) ops: component frequencies and
    conditional residual moments equal the ones the REFERENCE's `sample_observations` / `sample_var2_from_var1` bodies
    (src/factors/Factors.py:3146-3157, 3300-3380) produce on the same inputs (tests/golden/se2_factors.npz).  The
    candidates are 30 / 45 / 60 m from the poses, so the chosen component of a sample is identified by its value."""
    import nfisam_hip as nh
    g = dict(np.load(os.path.join(GOLDEN, "se2_factors.npz")))
    N = g["ada_pose"].shape[0]
    w, sigma, mix_obs = g["ada_weights"], float(g["ada_sigma"]), float(g["mix_obs"])
    cands = [g["ada_c0"], g["ada_c1"], g["ada_c2"]]
    src = torch.from_numpy(np.hstack([g["ada_pose"]] + cands).astype(np.float32)).to(DEV).contiguous()   # [N, 9]

    def op(code, a=0, b=0, c=0, k=0, p=(), cand=(), srcp=0):
        o = nh.SimOp()
        o.code, o.a, o.b, o.c, o.k, o.src = code, a, b, c, k, srcp
        for i, v in enumerate(p):
            o.p[i] = float(v)
        for i, v in enumerate(cand):
            o.cand[i] = int(v)
        return o
    cum = list(np.cumsum(w)) + [1.0]
    nw, ns, scale = g["nh_weights"], float(g["nh_sigma"]), float(g["nh_scale"])
    # columns: 0 ada obs | 1 nh obs | 2..4 pose | 5..10 candidates | 11..12 nh ring
    ops = [op(nh.SIM_COPY, a=9, b=0, c=2, k=3, srcp=src.data_ptr())] + \
          [op(nh.SIM_COPY, a=9, b=3 + 2 * k, c=5 + 2 * k, k=2, srcp=src.data_ptr()) for k in range(3)] + \
          [op(nh.SIM_ADA_OBS, a=2, c=0, cand=[5, 7, 9], k=3, p=cum + [sigma]),
           op(nh.SIM_NH_OBS, a=2, b=5, c=1, p=[ns, ns * scale, float(nw[0])]),
           op(nh.SIM_NH_RING, a=2, c=11, p=[mix_obs, ns, ns * scale, float(nw[0])])]
    out = nh.simulate_clique(ops, N, 13, 13, 777, DEV).cpu().numpy().astype(np.float64)
    dist = np.stack([np.linalg.norm(c - g["ada_pose"][:, :2], axis=1) for c in cands], 1)
    pick = np.abs(out[:, 0:1] - dist).argmin(1)
    # the reference's block layout -> its per-component residual std
    lo = 0
    for k, cnt in enumerate(g["ada_counts"]):
        ref_r = g["ada_obs"][lo:lo + cnt, 0] - dist[lo:lo + cnt, k]
        lo += cnt
        mine = out[pick == k, 0] - dist[pick == k, k]
        assert abs((pick == k).mean() - w[k]) < 4 * np.sqrt(w[k] * (1 - w[k]) / N), (k, (pick == k).mean())
        assert abs(mine.std() / ref_r.std() - 1) < 0.12 and abs(mine.mean()) < 5 * sigma / np.sqrt(len(mine))
    # null hypothesis: a two-scale mixture around the true range; compare the mixture's quantiles with the reference's draw
    r_dev = out[:, 1] - dist[:, 0]
    r_ref = g["nh_obs"][:, 0] - dist[:, 0]
    for q in (0.05, 0.25, 0.5, 0.75, 0.95):
        assert abs(np.quantile(r_dev, q) - np.quantile(r_ref, q)) < 0.12 * (1 + abs(np.quantile(r_ref, q))), q
    assert abs(r_dev.std() / r_ref.std() - 1) < 0.12
    rad_dev = np.linalg.norm(out[:, 11:13] - g["ada_pose"][:, :2], axis=1) - mix_obs
    rad_ref = np.linalg.norm(g["nh_ring"] - g["ada_pose"][:, :2], axis=1) - mix_obs
    assert abs(rad_dev.std() / rad_ref.std() - 1) < 0.12 and abs(rad_dev.mean() - rad_ref.mean()) < 0.15
    phi = np.arctan2(out[:, 12] - g["ada_pose"][:, 1], out[:, 11] - g["ada_pose"][:, 0])
    assert phi.min() < -3.0 and phi.max() > 3.0 and abs(np.cos(phi).mean()) < 0.08


def _sim_op(nh, code, a=0, b=0, c=0, k=0, p=(), srcp=0):
    o = nh.SimOp()
    o.code, o.a, o.b, o.c, o.k, o.src = code, a, b, c, k, srcp
    for i, v in enumerate(p):
        o.p[i] = float(v)
    return o


def test_fused_simulator_r2_ops_match_reference_bodies():
    """Ops 11-15 of `nfisam_simulate_clique` (the R2 family of the toy range-only examples) driven directly with the inputs
    of the REFERENCE's factor `sample` bodies (tests/golden/se2_factors.npz: src/factors/Factors.py:998-1030 displacement
    factor in its three directions, :2080-2135 R2-R2 range, :362,451 point / ring priors).  The reference vectors carry the
    injected noise draws `r2_noise2` / `f_noise1`: with the ops' noise scale at zero the device must return exactly the
    reference's output minus that draw; the priors are checked by their moments."""
    import nfisam_hip as nh
    g = dict(np.load(os.path.join(GOLDEN, "se2_factors.npz")))
    p1, p2, nz, nz1 = g["r2_p1"], g["r2_p2"], g["r2_noise2"], g["f_noise1"][:, 0]
    n = p1.shape[0]
    src = torch.from_numpy(np.hstack([p1, p2]).astype(np.float32)).to(DEV).contiguous()
    obs = [5.0, -5.0]
    # columns: 0..1 p1 | 2..3 p2 | 4..5 fwd(p1) | 6..7 bwd(p2) | 8..9 measurement | 10..11 ring around p1 | 12 range p1-p2
    ops = [_sim_op(nh, nh.SIM_COPY, a=4, b=0, c=0, k=4, srcp=src.data_ptr()),
           _sim_op(nh, nh.SIM_REL_R2_FWD, a=0, c=4, p=obs + [0, 0, 0]),
           _sim_op(nh, nh.SIM_REL_R2_BWD, a=2, c=6, p=obs + [0, 0, 0]),
           _sim_op(nh, nh.SIM_REL_R2_OBS, a=0, b=2, c=8, p=[0, 0, 0, 0, 0]),
           _sim_op(nh, nh.SIM_RING, a=0, c=10, p=[12.0, 0.0]),
           _sim_op(nh, nh.SIM_RANGE_OBS, a=0, b=2, c=12, p=[0.0])]
    out = nh.simulate_clique(ops, n, 13, 13, 99, DEV).cpu().numpy().astype(np.float64)
    np.testing.assert_allclose(out[:, 4:6], g["r2rel_fwd"] - nz, atol=2e-5)        # var2 = var1 + obs (+ noise)
    np.testing.assert_allclose(out[:, 6:8], g["r2rel_bwd"] + nz, atol=2e-5)        # var1 = var2 - obs (- noise)
    np.testing.assert_allclose(out[:, 8:10], g["r2rel_meas"] - nz, atol=2e-5)      # measurement = var2 - var1 (+ noise)
    np.testing.assert_allclose(np.linalg.norm(out[:, 10:12] - p1, axis=1), np.linalg.norm(g["r2ring"] - p1, axis=1) - nz1, atol=3e-5)
    np.testing.assert_allclose(out[:, 12], g["r2range_meas"][:, 0] - nz1, atol=3e-5)
    # noise of the displacement ops: mu + L z with the lower Cholesky factor as (l00, l10, l11)
    N = 40000
    cov = np.array([[0.5, 0.2], [0.2, 0.3]])
    L = np.linalg.cholesky(cov)
    lp = [L[0, 0], L[1, 0], L[1, 1]]
    big = torch.from_numpy(np.tile(np.array([[1.0, -2.0, 4.0, 0.5]], dtype=np.float32), (N, 1))).to(DEV).contiguous()
    ops = [_sim_op(nh, nh.SIM_COPY, a=4, b=0, c=0, k=4, srcp=big.data_ptr()),
           _sim_op(nh, nh.SIM_REL_R2_FWD, a=0, c=4, p=obs + lp), _sim_op(nh, nh.SIM_REL_R2_BWD, a=2, c=6, p=obs + lp),
           _sim_op(nh, nh.SIM_REL_R2_OBS, a=0, b=2, c=8, p=[0, 0] + lp),
           _sim_op(nh, nh.SIM_PRIOR_R2, c=10, p=[5.0, -3.0] + lp),
           _sim_op(nh, nh.SIM_PRIOR_R2_RING, c=12, p=[1.0, 2.0, 7.0, 0.5])]
    out = nh.simulate_clique(ops, N, 14, 14, 4242, DEV).cpu().numpy().astype(np.float64)
    se = 5.0 / np.sqrt(N)
    for cols, mean in ((slice(4, 6), np.array([6.0, -7.0])), (slice(6, 8), np.array([-1.0, 5.5])),
                       (slice(8, 10), np.array([3.0, 2.5])), (slice(10, 12), np.array([5.0, -3.0]))):
        d = out[:, cols]
        assert np.abs(d.mean(0) - mean).max() < se * np.sqrt(cov.max()), (cols, d.mean(0))
        np.testing.assert_allclose(np.cov(d.T), cov, atol=0.03 * cov.max() + 0.004)
    r = np.linalg.norm(out[:, 12:14] - np.array([1.0, 2.0]), axis=1)           # src/stats/Distributions.py:125-130
    assert abs(r.mean() - 7.0) < 0.5 * se and abs(r.std() - 0.5) < 0.012
    phi = np.arctan2(out[:, 13] - 2.0, out[:, 12] - 1.0)
    assert phi.min() < -3.1 and phi.max() > 3.1 and abs(np.cos(phi).mean()) < 0.03 and abs(np.sin(phi).mean()) < 0.03


def test_fused_simulator_noisy_se2_ops_have_the_reference_noise_model():
    """`PRIOR_SE2`, `REL_FWD`, `REL_BWD`, `REL_OBS` with a NON-TRIVIAL Cholesky factor.  In the reference's draws
    (tests/golden/se2_factors.npz: `prior_out`, `rel_fwd`, `rel_bwd`, src/factors/Factors.py:725-743, 1196-1317) the
    tangent-space residual Log(prior^-1 x) resp. Log(obs^-1 T_i^-1 T_j) IS the injected Gaussian draw -- asserted here on
    the golden vectors with the repo's SE2Pose (itself pinned to the reference's algebra).  The device draws its own
    Gaussians, so the same residuals of 40 000 device samples must have mean 0 and covariance L L^T."""
    import nfisam_hip as nh
    from geometry.TwoDimension import SE2Pose
    g = dict(np.load(os.path.join(GOLDEN, "se2_factors.npz")))
    prior, obs = SE2Pose(*g["prior_pose"]), SE2Pose(*g["rel_obs_value"])
    res = np.array([(prior.inverse() * SE2Pose(*x)).log_map() for x in g["prior_out"]])
    np.testing.assert_allclose(res, g["f_noise3"], atol=1e-12)
    res = np.array([(obs.inverse() * (SE2Pose(*a).inverse() * SE2Pose(*b))).log_map() for a, b in zip(g["f_x1"], g["rel_fwd"])])
    np.testing.assert_allclose(res, g["f_noise3"], atol=1e-12)
    res = np.array([(obs.inverse() * (SE2Pose(*a).inverse() * SE2Pose(*b))).log_map() for a, b in zip(g["rel_bwd"], g["f_x2"])])
    np.testing.assert_allclose(res, g["f_noise3"], atol=1e-12)

    N = 40000
    cov = np.array([[0.09, 0.02, -0.004], [0.02, 0.04, 0.006], [-0.004, 0.006, 0.0025]])
    L = np.linalg.cholesky(cov)
    lp = [L[0, 0], L[1, 0], L[1, 1], L[2, 0], L[2, 1], L[2, 2]]
    Ti, Tj = g["f_x1"][7], g["f_x2"][11]
    src = torch.from_numpy(np.tile(np.hstack([Ti, Tj]).astype(np.float32), (N, 1))).to(DEV).contiguous()
    # columns: 0..2 T_i | 3..5 T_j | 6..8 prior draw | 9..11 T_i * obs * Exp | 12..14 T_j * (obs * Exp)^-1 | 15..17 (T_i^-1 T_j) * Exp
    ops = [_sim_op(nh, nh.SIM_COPY, a=6, b=0, c=0, k=6, srcp=src.data_ptr()),
           _sim_op(nh, nh.SIM_PRIOR_SE2, c=6, p=list(g["prior_pose"]) + lp),
           _sim_op(nh, nh.SIM_REL_FWD, a=0, c=9, p=list(g["rel_obs_value"]) + lp),
           _sim_op(nh, nh.SIM_REL_BWD, a=3, c=12, p=list(g["rel_obs_value"]) + lp),
           _sim_op(nh, nh.SIM_REL_OBS, a=0, b=3, c=15, p=[0, 0, 0] + lp)]
    out = nh.simulate_clique(ops, N, 18, 18, 31337, DEV).cpu().numpy().astype(np.float64)
    pi, pj = SE2Pose(*Ti), SE2Pose(*Tj)
    d_ij = pi.inverse() * pj

    def residuals(rows, f):
        return np.array([f(SE2Pose(*x)).log_map() for x in rows[::4]])           # every 4th sample: 10 000 host log maps
    checks = {"prior": residuals(out[:, 6:9], lambda x: prior.inverse() * x),
              "rel_fwd": residuals(out[:, 9:12], lambda x: obs.inverse() * (pi.inverse() * x)),
              "rel_bwd": residuals(out[:, 12:15], lambda x: obs.inverse() * (x.inverse() * pj)),
              "rel_obs": residuals(out[:, 15:18], lambda x: d_ij.inverse() * x)}
    sd = np.sqrt(np.diag(cov))
    for name, e in checks.items():
        assert np.all(np.abs(e.mean(0)) < 5 * sd / np.sqrt(len(e)) + 2e-5), (name, e.mean(0))
        c = np.cov(e.T)
        assert np.abs(c - cov).max() < 0.06 * cov.max() and np.all(np.abs(np.diag(c) / np.diag(cov) - 1) < 0.06), (name, c)
        # the sign pattern of the off-diagonal terms pins the packing order of the Cholesky factor
        assert np.sign(c[0, 1]) == np.sign(cov[0, 1]) and np.sign(c[1, 2]) == np.sign(cov[1, 2]), (name, c)


def test_non_finite_batch_is_retried_once_then_raises():
    """A non-finite training loss ends the run of THAT clique with NFISAM_ERR_DOMAIN (the reference dies on its
    `Input outside domain` / discriminant checks, src/flows/utils.py:74-76,133); `NFiSAM.train_prepared` retries the clique
    once from fresh parameters and only then gives up.  The other cliques of a batched launch are unaffected."""
    from slam.NFiSAM import NFiSAM, NFiSAMArgs
    from slam.Variables import R2Variable, SE2Variable, VariableType
    rng = np.random.RandomState(2)
    good = ring_clique(600, rng)
    bad = good.copy()
    bad[5, 2] = np.inf
    L0, X0 = R2Variable("L0", VariableType.Landmark), SE2Variable("X0")
    clique = FakeClique(frontal=[X0], separator=[L0])
    solver = NFiSAM(NFiSAMArgs(flow_iterations=100, num_knots=9, learning_rate=0.02, device_simulation=False))
    with pytest.raises(RuntimeError, match="domain"):
        solver.fit_clique_density_model(clique, bad, [L0, X0], None)
    # batched: the healthy clique finishes although its neighbour fails twice
    p_good = solver.prepare_fit(clique, good, [L0, X0])
    p_bad = solver.prepare_fit(clique, good, [L0, X0])
    p_bad["training_data"] = p_bad["training_data"].clone()
    p_bad["training_data"][3, 1] = float("nan")
    with pytest.raises(RuntimeError, match="domain"):
        solver.train_prepared([p_good, p_bad])
    solver.train_prepared([p_good])
    model = solver.finish_fit(p_good)
    xs = model.conditional_sample_given_observation(conditional_dim=6, sample_number=32)
    assert xs.shape == (32, 6) and np.all(np.isfinite(xs)) and 50 <= p_good["iters"] <= 100


def test_small_uploads_through_the_pinned_ring_survive_its_reuse():
    """`nfisam_hip.upload`: small host arrays go to the device as ONE asynchronous copy from a pinned staging ring (one ring
    per thread and stream, 8 chunks of 64 KB).  Thousands of uploads on two streams with kernels queued in front of them --
    so the copies really are pending when the ring comes round -- must all arrive intact: a chunk may only be overwritten
    after the event behind its last copy has fired.  Also: several arrays per call (dtype and shape kept), an array larger
    than a chunk (pageable fallback), an empty call."""
    import nfisam_hip as nh
    dev = torch.device("cuda", 0)
    rng = np.random.RandomState(0)
    streams = [torch.cuda.current_stream(), torch.cuda.Stream()]
    kept = []
    busy = torch.randn(2048, 2048, device=dev)
    for i in range(3000):
        s = streams[i % 2]
        with torch.cuda.stream(s):
            if i % 50 == 0:
                for _ in range(4):
                    busy = (busy @ busy).clamp_(-1, 1)            # work in front of the copies of this stream
            a = rng.randn(rng.randint(1, 3000)).astype(np.float32)
            b = rng.randint(0, 255, size=(rng.randint(1, 40), 3)).astype(np.uint8)
            c = rng.randint(-5, 5, size=rng.randint(0, 9)).astype(np.int32)
            ta, tb, tc = nh.upload(a, b, c, device=dev)
            assert ta.dtype == torch.float32 and tb.dtype == torch.uint8 and tc.dtype == torch.int32
            assert tuple(tb.shape) == b.shape and tuple(tc.shape) == c.shape
            kept.append((s, (a, b, c), (ta, tb, tc)))
    big = rng.randn(40000).astype(np.float32)                       # > one chunk
    tbig, = nh.upload(big, device=dev)
    torch.cuda.synchronize()
    for s, host, devs in kept:
        for h, d in zip(host, devs):
            np.testing.assert_array_equal(d.cpu().numpy(), h)
    np.testing.assert_array_equal(tbig.cpu().numpy(), big)
    assert nh.upload(device=dev) == []
