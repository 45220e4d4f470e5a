#!/usr/bin/env python3
"""Generate golden vectors for the NF-iSAM flow hot path from the *reference itself*.

Runs ONLY in the build container (needs /root/reference).  Nothing from the reference is
copied: the reference's `flows.*` modules are imported read-only and executed on seeded
inputs; inputs and outputs are stored as small float32/float64 fixtures:

    tests/golden/nsf_<case>.npz        flow forward / loss / grads / Adam / inverse   (a1-a8)
    tests/golden/rqs_direct.npz        direct unconstrained_RQS / RQS / searchsorted  (a5, a6)
    tests/golden/normalize.npz         normalize_training_samples + (un)normalize     (a9, a10)
    tests/golden/validation_loop.npz   the training loop with a held-out set (the reference's `for` statement, NFiSAM.py:451-492,
                                       executed via `ast` extraction on the reference's own flow classes)   (a11 rule ii)
    tests/golden/se2_factors.npz       SE2Pose algebra + the `sample` bodies of the factor types of the clique
                                       simulator (f-2): SE(2) prior, odometry (3 directions), range ring / simulated
                                       range, k-way association and null-hypothesis mixtures

Reference entry points exercised (paths relative to /root/reference):
    src/flows/flows.py:43-137   NSF_AR.{forward,inverse,inverse_given_separator}
    src/flows/models.py:4-40    NormalizingFlowModel.forward
    src/flows/prior_dist.py:5-26 CustomMultivariateNormal
    src/flows/utils.py:17-164   searchsorted / unconstrained_RQS / RQS
    src/slam/NFiSAM.py:96-118,515-548  normalisation helpers (function bodies are executed via
        `ast` extraction because `slam.NFiSAM` imports TransportMaps/dynesty which are absent;
        only those three function definitions are compiled, in memory, never written to disk)
    src/geometry/TwoDimension.py:303-541  SE2Pose.{by_exp_map, __mul__, __truediv__, inverse, log_map} (imported)
    src/factors/Factors.py:725-743, 1196-1317, 2575-2649, 3146-3157, 3260-3276, 3300-3380  the `sample*` methods of
        UnarySE2ApproximateGaussianPriorFactor, SE2RelativeGaussianLikelihoodFactor,
        SE2R2RangeGaussianLikelihoodFactor, BinaryFactorMixture, AmbiguousDataAssociationFactor and
        BinaryFactorWithNullHypo -- method bodies executed via `ast` extraction on stand-in `self` objects, because
        `factors.Factors` imports TransportMaps (absent); nothing is written to disk

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
"""
import ast
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/src"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, REF)

from flows.flows import NSF_AR  # noqa: E402
from flows.models import NormalizingFlowModel  # noqa: E402
from flows.prior_dist import CustomMultivariateNormal  # noqa: E402
from flows import utils as ref_utils  # noqa: E402

torch.set_num_threads(1)

H = 8
B = 5.0
CASES = {
    # name: (n, D, K, seed)
    "n64_d2_k5": (64, 2, 5, 0),
    "n128_d6_k9": (128, 6, 9, 1),
    "n256_d11_k9": (256, 11, 9, 2),
    "n128_d16_k12": (128, 16, 12, 3),
    "n96_d1_k9": (96, 1, 9, 4),       # degenerate: only init_param, no conditioner
    # other (num_knots, hidden_dim) than the examples' usual ones: the reference takes any (src/flows/flows.py:51-60)
    "n96_d5_k3_h4": (96, 5, 3, 5, 4),
    "n80_d7_k15_h16": (80, 7, 15, 6, 16),
    "n64_d4_k15_h8": (64, 4, 15, 7, 8),   # num_knots=15: toy_examples/R2RangeGaussian_example/...incremental.py:85
    # hidden widths that are NOT a compiled kernel width (round 5: zero-padded into 8 / 16); the reference's own grid is
    # hidden_dims = [4, 6, 8, 10, 12]: example/slam/manhattan_world_with_range/lawnmower_4x4/run_nfisam.py:5-6
    "n72_d6_k9_h6": (72, 6, 9, 8, 6),
    "n100_d9_k12_h12": (100, 9, 12, 9, 12),
    "n70_d5_k9_h10": (70, 5, 9, 10, 10),
}


def unscramble(z_ref, ld_ref_elem, n, d):
    """Reference forward evaluates the spline on a dim-major flat vector and then reshapes
    it as (n, d) (src/flows/flows.py:88-93).  The mathematically correct layout is the
    (d, n) reshape transposed."""
    zc = z_ref.reshape(-1).reshape(d, n).T.copy()
    return zc


def state_to_np(sd):
    return {k.replace(".", "__"): v.detach().numpy().astype(np.float32) for k, v in sd.items()}


def gen_flow_case(name, n, D, K, seed, H=8):
    torch.manual_seed(seed)
    flow = NSF_AR(dim=D, K=K, B=B, hidden_dim=H)
    prior = CustomMultivariateNormal(dim=D)
    model = NormalizingFlowModel(prior, [flow])
    out = {}
    out["meta"] = np.array([n, D, K, H, seed], dtype=np.int64)
    out["B"] = np.array(B, dtype=np.float64)
    for k, v in state_to_np(flow.state_dict()).items():
        out["p0__" + k] = v

    # inputs: mostly inside the spline domain, ~3 % in the linear tails, plus special rows
    x = 1.8 * torch.randn(n, D)
    special = torch.tensor([B, -B, 0.0, B + 1.0, -(B + 1.0), 4.999, -4.999, 1e-3])
    for r in range(min(len(special), n)):
        x[r, r % D] = special[r]
    out["x"] = x.numpy().copy()

    # ---- forward (raw reference layout + corrected layout) -------------------------------
    z_raw, ld_raw = flow(x)
    out["z_raw"] = z_raw.detach().numpy().copy()
    out["logdet_raw"] = ld_raw.detach().numpy().copy()
    # elementwise log-dets in the reference's flat (dim-major) order
    Ws = torch.zeros((n * D, K)); Hs = torch.zeros_like(Ws); Ds = torch.zeros_like(Ws)[:, :-1]
    for i in range(D):
        if i == 0:
            p = flow.init_param.expand(n, 3 * K - 1)
        else:
            p = flow.layers[i - 1](x[:, :i])
        Ws[i * n:(i + 1) * n], Hs[i * n:(i + 1) * n], Ds[i * n:(i + 1) * n] = torch.split(p, K, dim=1)
    zs, lds = ref_utils.unconstrained_RQS(x.transpose(0, 1).flatten(), Ws, Hs, Ds,
                                          inverse=False, tail_bound=B)
    z_c = zs.reshape(D, n).T.detach()
    ld_c = lds.reshape(D, n).sum(0).detach()
    assert torch.equal(zs.reshape(n, D), z_raw)
    out["z"] = z_c.numpy().copy()
    out["logdet"] = ld_c.numpy().copy()
    out["spline_params"] = torch.cat([Ws, Hs, Ds], 1).reshape(D, n, 3 * K - 1).detach().numpy().copy()
    out["prior_logprob"] = prior.log_prob(z_c).numpy().copy()

    # ---- loss + gradients (loss is invariant to the scramble for a single layer) ---------
    z_m, plp, ld_m = model(x)
    loss = -torch.mean(plp + ld_m)
    out["loss"] = np.array(loss.item(), dtype=np.float64)
    loss_c = -torch.mean(prior.log_prob(z_c) + ld_c)
    out["loss_correct_layout"] = np.array(loss_c.item(), dtype=np.float64)
    grads = torch.autograd.grad(loss, list(flow.parameters()))
    for (k, _), g in zip(flow.named_parameters(), grads):
        out["g0__" + k.replace(".", "__")] = g.numpy().astype(np.float32)

    # ---- Adam trajectory (src/slam/NFiSAM.py:425,469-475) ---------------------------------
    lr = 0.025
    opt = torch.optim.Adam(model.parameters(), lr=lr)
    losses = []
    for it in range(10):
        opt.zero_grad()
        z_m, plp, ld_m = model(x)
        loss = -torch.mean(plp + ld_m)
        losses.append(loss.item())
        loss.backward()
        opt.step()
        if it + 1 in (1, 2, 10):
            for k, v in state_to_np(flow.state_dict()).items():
                out["p%d__%s" % (it + 1, k)] = v
    out["adam_lr"] = np.array(lr, dtype=np.float64)
    out["adam_losses"] = np.array(losses, dtype=np.float64)

    # ---- inverse paths, evaluated with the *initial* parameters ---------------------------
    torch.manual_seed(seed)
    flow0 = NSF_AR(dim=D, K=K, B=B, hidden_dim=H)
    with torch.no_grad():
        zin = z_c.clone()
        x_rec, ld_inv = flow0.inverse(zin)
        out["inv_x"] = x_rec.numpy().copy()
        out["inv_logdet"] = ld_inv.numpy().copy()
        # fresh latent draws (not images of x): exercises tails in the z domain too
        zlat = 1.7 * torch.randn(n, D)
        zlat[0, 0] = B; zlat[1, 0] = -B; zlat[2, D - 1] = B + 0.5
        out["zlat"] = zlat.numpy().copy()
        xl, ldl = flow0.inverse(zlat.clone())
        out["zlat_inv_x"] = xl.numpy().copy()
        out["zlat_inv_logdet"] = ldl.numpy().copy()
        for Ds_ in (0, 1, 3):
            if Ds_ >= D:
                continue
            xs = None if Ds_ == 0 else x[:, :Ds_].clone()
            xf = flow0.inverse_given_separator(zlat[:, Ds_:].clone(), xs)
            out["igs%d_x" % Ds_] = xf.numpy().copy()
    np.savez_compressed(os.path.join(OUT, "nsf_%s.npz" % name), **out)
    print(name, "loss", out["loss"], "loss(correct layout)", out["loss_correct_layout"],
          "max|z_raw-z|", float(np.abs(out["z_raw"] - out["z"]).max()))


def gen_rqs_direct():
    out = {}
    torch.manual_seed(11)
    for tag, (M, K, tb) in {"a": (300, 5, 1.0), "b": (500, 9, 5.0), "c": (200, 12, 5.0), "d": (64, 3, 2.5)}.items():
        W = 1.5 * torch.randn(M, K); Hh = 1.5 * torch.randn(M, K); Dd = 1.5 * torch.randn(M, K - 1)
        inp = 0.6 * tb * torch.randn(M)
        inp[0] = tb; inp[1] = -tb; inp[2] = 0.0; inp[3] = tb * 1.2; inp[4] = -tb * 1.2
        y, ld = ref_utils.unconstrained_RQS(inp.clone(), W.clone(), Hh.clone(), Dd.clone(),
                                            inverse=False, tail_bound=tb)
        xi, ldi = ref_utils.unconstrained_RQS(inp.clone(), W.clone(), Hh.clone(), Dd.clone(),
                                              inverse=True, tail_bound=tb)
        for k, v in dict(W=W, H=Hh, D=Dd, inp=inp, fwd=y, fwd_ld=ld, inv=xi, inv_ld=ldi).items():
            out["%s_%s" % (tag, k)] = v.numpy().copy()
        out["%s_tb" % tag] = np.array(tb)
    # bounded RQS (no tails) on [0,1] + searchsorted
    M, K = 128, 7
    W = torch.randn(M, K); Hh = torch.randn(M, K); Dd = torch.randn(M, K + 1)
    inp = torch.rand(M); inp[0] = 0.0; inp[1] = 1.0
    y, ld = ref_utils.RQS(inp.clone(), W, Hh, Dd, inverse=False)
    xi, ldi = ref_utils.RQS(inp.clone(), W, Hh, Dd, inverse=True)
    for k, v in dict(W=W, H=Hh, D=Dd, inp=inp, fwd=y, fwd_ld=ld, inv=xi, inv_ld=ldi).items():
        out["rqs_%s" % k] = v.numpy().copy()
    bins = torch.sort(torch.rand(40, 6), dim=1)[0]
    q = torch.rand(40)
    q[0] = bins[0, 3]            # exactly on a knot
    q[1] = bins[1, 5]            # exactly on the last knot (eps bump matters)
    b_in = bins.clone()
    idx = ref_utils.searchsorted(b_in, q)
    out["ss_bins"] = bins.numpy().copy(); out["ss_q"] = q.numpy().copy(); out["ss_idx"] = idx.numpy().copy()
    out["ss_bins_after"] = b_in.numpy().copy()   # reference bumps the last knot in place
    np.savez_compressed(os.path.join(OUT, "rqs_direct.npz"), **out)
    print("rqs_direct done")


def _extract_functions(path, names):
    """Compile selected function definitions of a reference file in memory (no import)."""
    with open(path) as f:
        tree = ast.parse(f.read())
    found = {}
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef) and node.name in names:
            found[node.name] = node
    mod = ast.Module(body=[found[n] for n in names], type_ignores=[])
    ns = {}
    from scipy.stats import circmean
    TWO_PI = 2 * np.pi

    def theta_to_pipi(theta):  # src/utils/Functions.py:20-21 (imported symbol in NFiSAM.py:16)
        return (theta + np.pi) % TWO_PI - np.pi
    ns.update(np=np, torch=torch, circmean=circmean, theta_to_pipi=theta_to_pipi)
    exec(compile(mod, path, "exec"), ns)
    return ns


def gen_normalize():
    ns = _extract_functions(os.path.join(REF, "slam/NFiSAM.py"),
                            ["normalize_samples", "unnormalize_samples", "normalize_training_samples"])
    rng = np.random.RandomState(5)
    n, D = 400, 7
    circ = [False, False, True, False, True, False, True]
    s = rng.randn(n, D) * np.array([3.0, 0.5, 0.4, 10.0, 2.5, 1e-7, 0.9]) + \
        np.array([10.0, -3.0, 3.0, 100.0, -1.0, 2.0, -3.1])
    circ_idx = np.where(circ)[0]
    s[:, circ_idx] = (s[:, circ_idx] + np.pi) % (2 * np.pi) - np.pi
    out = {"samples": s.copy(), "circular": np.array(circ)}
    td, mu, sd = ns["normalize_training_samples"](None, s.copy(), circ, "NSF_AR")
    out["train_norm"] = td.numpy().copy(); out["mean"] = mu.numpy().copy(); out["std"] = sd.numpy().copy()
    holder = types.SimpleNamespace(circular_dim_list=circ, samples_mean=mu, samples_std=sd)
    q = rng.randn(50, 4) * 2 + np.array([9.0, -3.0, 2.9, 101.0])
    out["q"] = q.copy()
    out["q_norm_init0"] = ns["normalize_samples"](holder, torch.tensor(np.float32(q)), 0).numpy().copy()
    zz = rng.randn(50, 3).astype(np.float32) * 1.5
    out["zz"] = zz.copy()
    out["zz_unnorm_init4"] = ns["unnormalize_samples"](holder, torch.tensor(zz.copy()), 4).numpy().copy()
    np.savez_compressed(os.path.join(OUT, "normalize.npz"), **out)
    print("normalize done")


def _extract_methods(path, wanted):
    """{(class, method): function} compiled in memory from the reference file (no import of the module)."""
    with open(path) as f:
        tree = ast.parse(f.read())
    defs = []
    for node in tree.body:
        if isinstance(node, ast.ClassDef):
            for item in node.body:
                if isinstance(item, ast.FunctionDef) and (node.name, item.name) in wanted:
                    item.name = "%s__%s" % (node.name, item.name)
                    item.decorator_list = []
                    defs.append(item)
    mod = ast.Module(body=defs, type_ignores=[])
    from typing import Dict, Union
    from geometry.TwoDimension import Point2, Rot2, SE2Pose
    ns = dict(np=np, SE2Pose=SE2Pose, Point2=Point2, Rot2=Rot2, Union=Union, Dict=Dict, Variable=object)
    exec(compile(mod, path, "exec"), ns)
    return {k: ns["%s__%s" % k] for k in wanted}


class _FixedNoise(object):
    """Stand-in for the factors' `_noise_distribution`: hands out a prescribed array."""

    def __init__(self, arr):
        self.arr = arr

    def rvs(self, n):
        assert n == self.arr.shape[0]
        return self.arr.copy()


class _GaussNoise(object):
    """scipy-free N(0, sigma^2) column drawn from numpy's global generator (seeded by the caller)."""

    def __init__(self, sigma):
        self.sigma = sigma

    def rvs(self, n):
        return self.sigma * np.random.standard_normal((n, 1))


def gen_se2_and_factor_samplers():
    from geometry.TwoDimension import SE2Pose
    rng = np.random.RandomState(11)
    out = {}
    # ---- SE2Pose algebra (src/geometry/TwoDimension.py:303-541) ----
    n = 64
    v = rng.randn(n, 3) * np.array([3.0, 2.0, 1.5])
    v[0] = [1.0, -2.0, 0.0]; v[1] = [0.5, 0.25, 1e-12]; v[2] = [1.0, 1.0, np.pi - 1e-9]; v[3] = [-1.0, 2.0, -3.1]
    a = rng.randn(n, 3) * np.array([10.0, 10.0, 2.0]); a[:, 2] = (a[:, 2] + np.pi) % (2 * np.pi) - np.pi
    b = rng.randn(n, 3) * np.array([5.0, 5.0, 2.0]); b[:, 2] = (b[:, 2] + np.pi) % (2 * np.pi) - np.pi
    a[5, 2], b[5, 2] = 3.0, 2.5      # sum wraps
    out["se2_v"] = v
    out["se2_exp"] = np.array([SE2Pose.by_exp_map(x).array for x in v])
    out["se2_a"], out["se2_b"] = a, b
    out["se2_mul"] = np.array([(SE2Pose.by_array(x) * SE2Pose.by_array(y)).array for x, y in zip(a, b)])
    out["se2_div"] = np.array([(SE2Pose.by_array(x) / SE2Pose.by_array(y)).array for x, y in zip(a, b)])
    out["se2_inv"] = np.array([SE2Pose.by_array(x).inverse().array for x in a])
    out["se2_log"] = np.array([SE2Pose.by_array(x).log_map() for x in a])
    out["se2_explog"] = np.array([SE2Pose.by_exp_map(SE2Pose.by_array(x).log_map()).array for x in a])

    # ---- factor sampler bodies on stand-in objects ----
    S = types.SimpleNamespace
    wanted = [("UnarySE2ApproximateGaussianPriorFactor", "sample"), ("SE2RelativeGaussianLikelihoodFactor", "sample"),
              ("SE2R2RangeGaussianLikelihoodFactor", "sample_var2_from_var1"),
              ("SE2R2RangeGaussianLikelihoodFactor", "sample_var1_from_var2"),
              ("SE2R2RangeGaussianLikelihoodFactor", "sample_observations"),
              ("BinaryFactorMixture", "sample_observations"), ("AmbiguousDataAssociationFactor", "sample_observer"),
              ("BinaryFactorWithNullHypo", "sample_var2_from_var1"), ("BinaryFactorWithNullHypo", "sample_observations"),
              ("R2RelativeGaussianLikelihoodFactor", "sample"), ("R2RangeGaussianLikelihoodFactor", "sample_var2_from_var1"),
              ("R2RangeGaussianLikelihoodFactor", "sample_observations")]
    M = _extract_methods(os.path.join(REF, "factors/Factors.py"), wanted)
    m = 48
    noise3 = rng.randn(m, 3) * np.array([0.3, 0.2, 0.1])
    noise1 = rng.randn(m, 1) * 0.7
    x1 = rng.randn(m, 3) * np.array([8.0, 8.0, 1.5]); x1[:, 2] = (x1[:, 2] + np.pi) % (2 * np.pi) - np.pi
    x2 = rng.randn(m, 3) * np.array([8.0, 8.0, 1.5]); x2[:, 2] = (x2[:, 2] + np.pi) % (2 * np.pi) - np.pi
    lm = rng.randn(m, 2) * 15.0
    out.update(f_noise3=noise3, f_noise1=noise1, f_x1=x1, f_x2=x2, f_lm=lm)
    prior_pose = np.array([3.0, -1.0, 2.8])
    prior = S(_noise_distribution=_FixedNoise(noise3), _correlated_R_t=True, _prior_pose=SE2Pose.by_array(prior_pose))
    out["prior_pose"] = prior_pose
    out["prior_out"] = M[wanted[0]](prior, m)
    obs = np.array([4.0, 0.7, -2.9])
    rel = S(_noise_distribution=_FixedNoise(noise3), _correlated_Rt=True, _observation=SE2Pose.by_array(obs), _unary_dim=3)
    out["rel_obs_value"] = obs
    out["rel_fwd"] = M[wanted[1]](rel, var1=x1, var2=None)          # var2 from var1
    out["rel_bwd"] = M[wanted[1]](rel, var1=None, var2=x2)          # var1 from var2
    out["rel_meas"] = M[wanted[1]](rel, var1=x1, var2=x2)           # simulated measurement
    zero3 = np.zeros((m, 3))                                        # noise-free variants (the device ops are checked on these)
    rel0 = S(_noise_distribution=_FixedNoise(zero3), _correlated_Rt=True, _observation=SE2Pose.by_array(obs), _unary_dim=3)
    out["rel_fwd0"] = M[wanted[1]](rel0, var1=x1, var2=None)
    out["rel_bwd0"] = M[wanted[1]](rel0, var1=None, var2=x2)
    out["rel_meas0"] = M[wanted[1]](rel0, var1=x1, var2=x2)
    pose_var = S(dim=3, t_dim_indices=[0, 1])
    lmk_var = S(dim=2, t_dim_indices=[0, 1])
    rng_f = S(_noise_distribution=_FixedNoise(noise1), _observation=np.array([12.0]), var1=pose_var, var2=lmk_var)
    np.random.seed(5)
    out["ring_angles"] = np.random.uniform(-np.pi, np.pi, m)           # what the body draws after the fixed noise
    np.random.seed(5)
    out["ring_from_pose"] = M[wanted[2]](rng_f, x1)
    np.random.seed(5)
    out["ring_from_lmk"] = M[wanted[3]](rng_f, lm)                   # pose xy from the landmark (translation part only)
    out["range_meas"] = M[wanted[4]](rng_f, x1, lm)
    rng_f0 = S(_noise_distribution=_FixedNoise(np.zeros((m, 1))), _observation=np.array([12.0]), var1=pose_var, var2=lmk_var)
    out["range_meas0"] = M[wanted[4]](rng_f0, x1, lm)
    # ---- R2 family of the toy examples (src/factors/Factors.py:998-1030, 2080-2135) ----
    noise2 = rng.randn(m, 2) * np.array([0.3, 0.2])
    p1, p2 = rng.randn(m, 2) * 6.0, rng.randn(m, 2) * 6.0
    r2rel = S(_noise_distribution=_FixedNoise(noise2), _observation=np.array([5.0, -5.0]), _unary_dim=2)
    out.update(r2_noise2=noise2, r2_p1=p1, r2_p2=p2)
    out["r2rel_fwd"] = M[wanted[9]](r2rel, var1=p1, var2=None)
    out["r2rel_bwd"] = M[wanted[9]](r2rel, var1=None, var2=p2)
    out["r2rel_meas"] = M[wanted[9]](r2rel, var1=p1, var2=p2)
    r2rng = S(_noise_distribution=_FixedNoise(noise1), _observation=np.array([12.0]), _unary_dim=2)
    np.random.seed(5)
    out["r2ring"] = M[wanted[10]](r2rng, p1)                          # same angle draws as `ring_angles`
    out["r2range_meas"] = M[wanted[11]](r2rng, p1, p2)
    # ---- mixtures, real random numbers under a seed: frequencies and conditional moments are pinned ----
    N = 3000
    pose = rng.randn(N, 3) * np.array([2.0, 2.0, 0.5])
    cands = [np.array([30.0, 0.0]) + 0.1 * rng.randn(N, 2), np.array([0.0, 45.0]) + 0.1 * rng.randn(N, 2),
             np.array([-60.0, 0.0]) + 0.1 * rng.randn(N, 2)]
    class V(object):                      # hashable stand-in for slam.Variables.Variable
        def __init__(self, name, dim):
            self.name, self.dim, self.t_dim_indices = name, dim, [0, 1]
    Pv, Lv = V("X", 3), [V("L%d" % k, 2) for k in range(3)]

    def range_comp(v1, v2, sigma):
        c = S(_noise_distribution=_GaussNoise(sigma), _observation=np.array([20.0]), var1=v1, var2=v2)
        c.sample_var2_from_var1 = lambda x: M[wanted[2]](c, x)
        c.sample_var1_from_var2 = lambda x: M[wanted[3]](c, x)
        c.sample_observations = lambda p, q: M[wanted[4]](c, p, q)
        c.sample = lambda var1=None, var2=None: (c.sample_var1_from_var2(var2) if var1 is None else
                                                 (c.sample_var2_from_var1(var1) if var2 is None else
                                                  c.sample_observations(var1, var2)))
        return c
    w = np.array([0.5, 0.3, 0.2])
    ada = S(observer_var=Pv, observed_vars=Lv, weights=w, measurement_dim=1,
            components=[range_comp(Pv, Lv[k], 0.5) for k in range(3)])
    np.random.seed(21)
    out["ada_counts"] = np.random.multinomial(N, w)
    np.random.seed(21)
    out["ada_obs"] = M[wanted[5]](ada, {Pv: pose, Lv[0]: cands[0], Lv[1]: cands[1], Lv[2]: cands[2]})
    out.update(ada_pose=pose, ada_c0=cands[0], ada_c1=cands[1], ada_c2=cands[2], ada_weights=w, ada_sigma=np.array(0.5))
    nh_w, nh_sigma, nh_scale = np.array([0.7, 0.3]), 0.5, 6.0
    nh = S(var1=Pv, var2=Lv[0], weights=nh_w, measurement_dim=1,
           components=[range_comp(Pv, Lv[0], nh_sigma), range_comp(Pv, Lv[0], nh_sigma * nh_scale)])
    np.random.seed(22)
    out["nh_counts"] = np.random.multinomial(N, nh_w)
    np.random.seed(22)
    out["nh_obs"] = M[wanted[8]](nh, pose, cands[0])
    np.random.seed(23)
    out["nh_ring_counts"] = np.random.multinomial(N, nh_w)
    np.random.seed(23)
    out["nh_ring"] = M[wanted[7]](nh, pose)
    out.update(nh_weights=nh_w, nh_sigma=np.array(nh_sigma), nh_scale=np.array(nh_scale), mix_obs=np.array(20.0))
    np.savez_compressed(os.path.join(OUT, "se2_factors.npz"), **out)
    print("se2_factors done", {k: np.shape(v) for k, v in out.items() if k.startswith(("prior", "rel", "ring", "range"))})


def gen_validation_loop():
    """The reference's training loop WITH a held-out set (src/slam/NFiSAM.py:451-492): the `for i in range(flow_iterations)`
    statement of `NFiSAM.fit_clique_density_model` is taken out of the reference file by `ast` (the module imports the
    absent TransportMaps), compiled in memory and executed on the reference's own flow / model / prior classes (imported)
    and a seeded batch.  Stored: inputs, initial state_dict, arguments, and what the loop did -- iterations run, the
    loss record, every validation loss, the final parameters.  -> tests/golden/validation_loop.npz"""
    path = os.path.join(REF, "slam/NFiSAM.py")
    with open(path) as f:
        tree = ast.parse(f.read())
    loop = None
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef) and node.name == "fit_clique_density_model":
            for st in ast.walk(node):
                if isinstance(st, ast.For) and isinstance(st.iter, ast.Call) and getattr(st.iter.func, "id", "") == "range" and \
                        st.iter.args and getattr(st.iter.args[0], "id", "") == "flow_iterations":
                    loop = st
    assert loop is not None
    code = compile(ast.Module(body=[loop], type_ignores=[]), path, "exec")
    out = {}
    for case, (n_tr, n_val, D, K, lr, iters, interval, rate, seed) in {
            "overfit": (96, 400, 3, 9, 0.015, 400, 10, 2.0, 11),       # a small training set: the held-out loss turns early
            "interval7": (80, 300, 4, 5, 0.04, 300, 7, 2.0, 12),
            "budget": (400, 400, 3, 9, 0.01, 60, 10, 2.0, 13)}.items():   # the budget runs out before the rule fires
        torch.manual_seed(seed)
        rng = np.random.RandomState(seed)
        def draw(n):
            r = 1.0 + 0.15 * rng.randn(n); ph = rng.uniform(-np.pi, np.pi, n)
            cols = [r * np.cos(ph), r * np.sin(ph)] + [0.6 * rng.randn(n) + 0.4 * np.cos(ph) for _ in range(D - 2)]
            x = np.stack(cols, 1)
            return ((x - x.mean(0)) / x.std(0)).astype(np.float32)       # (each set by its OWN statistics, NFiSAM.py:379-383)
        x_tr, x_va = draw(n_tr), draw(n_val)
        flow = NSF_AR(dim=D, K=K, hidden_dim=H)
        model = NormalizingFlowModel(CustomMultivariateNormal(dim=D), [flow])
        sd0 = {k: v.detach().clone() for k, v in flow.state_dict().items()}
        msgs = []
        logger = types.SimpleNamespace(info=lambda m: msgs.append(str(m)))
        ns = dict(torch=torch, flow_iterations=iters, slower_stop_iter=None, testing_data=torch.tensor(x_va),
                  training_data=torch.tensor(x_tr), validation_interval=interval, last_validation_loss=float("inf"),
                  clique_density_model=model, logger=logger, optimizer=torch.optim.Adam(model.parameters(), lr=lr),
                  iter_loss=torch.zeros(iters), average_window=50, loss_avg=None, loss_delta_tol=1e-2,
                  self=types.SimpleNamespace(_args=types.SimpleNamespace(slower_stop_rate=rate)))
        exec(code, ns)
        il = ns["iter_loss"].numpy().copy()
        vals = [float(m.split("validation loss:")[1]) for m in msgs if "validation loss:" in m]
        out[case + "_x_train"], out[case + "_x_val"] = x_tr, x_va
        out[case + "_args"] = np.array([D, K, H, iters, interval], dtype=np.int64)
        out[case + "_lr_rate"] = np.array([lr, rate])
        for k, v in state_to_np(sd0).items():
            out[case + "_sd0_" + k] = v
        for k, v in state_to_np(flow.state_dict()).items():
            out[case + "_sd1_" + k] = v
        out[case + "_iter_loss"] = il
        out[case + "_iters_run"] = np.array(int(np.count_nonzero(il)))
        out[case + "_val_losses"] = np.array(vals, dtype=np.float64)
        out[case + "_slower_stop_iter"] = np.array(-1 if ns["slower_stop_iter"] is None else int(ns["slower_stop_iter"]))
        print("validation loop", case, "iterations", int(np.count_nonzero(il)), "of", iters, "evaluations", len(vals),
              "slower_stop_iter", ns["slower_stop_iter"])
    np.savez_compressed(os.path.join(OUT, "validation_loop.npz"), **out)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "validation":
        gen_validation_loop()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "flow":          # only the named flow cases
        for name in sys.argv[2:]:
            gen_flow_case(name, *CASES[name])
        sys.exit(0)
    for name, spec in CASES.items():
        gen_flow_case(name, *spec)
    gen_rqs_direct()
    gen_normalize()
    gen_se2_and_factor_samplers()
    gen_validation_loop()
