"""f-2 pinned to the REFERENCE: `geometry/TwoDimension.py` and the host factor samplers
(`factors/Factors.py`) against vectors produced by the reference's own `SE2Pose` algebra
(src/geometry/TwoDimension.py:303-541, imported) and by the `sample*` method bodies of its factor classes
(src/factors/Factors.py:725-743, 1196-1317, 2575-2649, 3146-3157, 3260-3276, 3300-3380; executed via `ast` extraction
in tests/golden/make_golden.py::gen_se2_and_factor_samplers).  Deterministic parts (prescribed noise) are compared to
1e-12; the mixture factors (k-way data association, null hypothesis) by component frequencies and conditional moments."""
import os

import numpy as np
import pytest

import factors.Factors as FF
from geometry.TwoDimension import SE2Pose, se2_compose, se2_exp, se2_inverse, se2_log, wrap_pi
from slam.Variables import R2Variable, SE2Variable, VariableType

G = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "se2_factors.npz")))


def pose_close(got, ref, atol=1e-12):
    np.testing.assert_allclose(got[:, :2], ref[:, :2], atol=atol, rtol=1e-12)
    d = wrap_pi(got[:, 2] - ref[:, 2])
    # an angle that is +-pi up to rounding may be stored as either end of [-pi, pi)
    assert np.all(np.minimum(np.abs(d), 2 * np.pi - np.abs(d)) < 1e-9), np.abs(d).max()


def test_se2_algebra_matches_reference():
    pose_close(se2_exp(G["se2_v"]), G["se2_exp"])
    pose_close(se2_compose(G["se2_a"], G["se2_b"]), G["se2_mul"])
    pose_close(se2_compose(G["se2_a"], se2_inverse(G["se2_b"])), G["se2_div"])
    pose_close(se2_inverse(G["se2_a"]), G["se2_inv"])
    np.testing.assert_allclose(se2_log(G["se2_a"]), G["se2_log"], atol=1e-11, rtol=1e-11)
    pose_close(se2_exp(se2_log(G["se2_a"])), G["se2_explog"], atol=1e-10)
    # the value class
    for k in (0, 5, 17):
        a, b = SE2Pose.by_array(G["se2_a"][k]), SE2Pose.by_array(G["se2_b"][k])
        pose_close((a * b).array[None], G["se2_mul"][k][None])
        pose_close((a / b).array[None], G["se2_div"][k][None])
        pose_close(a.inverse().array[None], G["se2_inv"][k][None])
        np.testing.assert_allclose(a.log_map(), G["se2_log"][k], atol=1e-11)
        pose_close(SE2Pose.by_exp_map(G["se2_v"][k]).array[None], G["se2_exp"][k][None])
    # reference ranges: theta in [-pi, pi)
    assert np.all(G["se2_mul"][:, 2] >= -np.pi) and np.all(G["se2_mul"][:, 2] < np.pi)
    got = se2_compose(G["se2_a"], G["se2_b"])[:, 2]
    assert np.all(got >= -np.pi) and np.all(got < np.pi)


def test_se2_prior_and_odometry_samplers_match_reference_bodies(monkeypatch):
    """Prescribed tangent-space noise -> identical samples (x = prior * Exp(eps); T_j = T_i * (obs * Exp(eps)); T_i =
    T_j / (obs * Exp(eps)); measurement = (T_i^-1 T_j) * Exp(eps))."""
    monkeypatch.setattr(FF, "_gaussian_noise", lambda chol, n: G["f_noise3"].copy())
    X1, X2 = SE2Variable("X1"), SE2Variable("X2")
    cov = np.diag([0.09, 0.04, 0.01])
    prior = FF.UnarySE2ApproximateGaussianPriorFactor(X1, G["prior_pose"], cov)
    pose_close(prior.sample(48), G["prior_out"])
    rel = FF.SE2RelativeGaussianLikelihoodFactor(X1, X2, G["rel_obs_value"], cov)
    pose_close(rel.sample(var1=G["f_x1"], var2=None), G["rel_fwd"])
    pose_close(rel.sample(var1=None, var2=G["f_x2"]), G["rel_bwd"])
    pose_close(rel.sample(var1=G["f_x1"], var2=G["f_x2"]), G["rel_meas"])
    monkeypatch.setattr(FF, "_gaussian_noise", lambda chol, n: np.zeros((n, 3)))
    pose_close(rel.sample(var1=G["f_x1"], var2=None), G["rel_fwd0"])
    pose_close(rel.sample(var1=None, var2=G["f_x2"]), G["rel_bwd0"])
    pose_close(rel.sample(var1=G["f_x1"], var2=G["f_x2"]), G["rel_meas0"])


def test_range_factor_sampler_matches_reference_bodies(monkeypatch):
    """Ring around the sampled end (radius = observation + noise, uniform bearing) and simulated range, with the
    reference's draws (src/factors/Factors.py:2575-2619) injected."""
    X, L = SE2Variable("X1"), R2Variable("L1", VariableType.Landmark)
    sigma = 0.7
    f = FF.SE2R2RangeGaussianLikelihoodFactor(X, L, 12.0, sigma)
    monkeypatch.setattr(np.random, "standard_normal", lambda n: G["f_noise1"][:, 0] / sigma)
    monkeypatch.setattr(np.random, "uniform", lambda lo, hi, n: G["ring_angles"].copy())
    np.testing.assert_allclose(f.sample(var1=G["f_x1"], var2=None), G["ring_from_pose"], atol=1e-11)
    np.testing.assert_allclose(f.sample(var1=None, var2=G["f_lm"]), G["ring_from_lmk"], atol=1e-11)
    np.testing.assert_allclose(f.sample(var1=G["f_x1"], var2=G["f_lm"]), G["range_meas"], atol=1e-11)


def _component_stats(obs, pose, cands, counts):
    """Rows are grouped by component (the reference fills contiguous blocks, src/factors/Factors.py:3146-3157)."""
    out, lo = [], 0
    for k, c in enumerate(counts):
        r = obs[lo:lo + c, 0] - np.linalg.norm(cands[k][lo:lo + c] - pose[lo:lo + c, :2], axis=1)
        out.append((r.mean(), r.std()))
        lo += c
    return out


def test_association_and_null_hypothesis_mixtures_match_reference_statistics():
    """k-way data association (simulated measurement from a multinomially chosen candidate) and the two-component
    null-hypothesis factor (second component: sigma x null_sigma_scale): same component frequencies (within 4 sigma of the
    multinomial) and the same conditional residual moments as the reference's bodies produce under a seed."""
    N = G["ada_pose"].shape[0]
    X = SE2Variable("X")
    Ls = [R2Variable("L%d" % k, VariableType.Landmark) for k in range(3)]
    w, sigma = G["ada_weights"], float(G["ada_sigma"])
    cands = [G["ada_c0"], G["ada_c1"], G["ada_c2"]]
    # what the reference drew
    ref_stats = _component_stats(G["ada_obs"], G["ada_pose"], cands, G["ada_counts"])
    for k, (mu, sd) in enumerate(ref_stats):
        assert abs(G["ada_counts"][k] / N - w[k]) < 4 * np.sqrt(w[k] * (1 - w[k]) / N)
        assert abs(mu) < 4 * sigma / np.sqrt(G["ada_counts"][k]) and abs(sd / sigma - 1) < 0.1
    # ours
    np.random.seed(3)
    ada = FF.AmbiguousDataAssociationFactor(X, Ls, w.copy(), FF.SE2R2RangeGaussianLikelihoodFactor, float(G["mix_obs"]), sigma)
    counts = None
    orig = np.random.multinomial

    def spy(n, p):
        nonlocal counts
        counts = orig(n, p)
        return counts
    np.random.multinomial = spy
    try:
        obs = ada.sample_observations({X: G["ada_pose"], Ls[0]: cands[0], Ls[1]: cands[1], Ls[2]: cands[2]})
    finally:
        np.random.multinomial = orig
    assert obs.shape == (N, 1) and counts is not None and counts.sum() == N
    for k, (mu, sd) in enumerate(_component_stats(obs, G["ada_pose"], cands, counts)):
        assert abs(counts[k] / N - w[k]) < 4 * np.sqrt(w[k] * (1 - w[k]) / N)
        assert abs(mu) < 4 * sigma / np.sqrt(counts[k]) and abs(sd / ref_stats[k][1] - 1) < 0.12, (k, mu, sd)
    # null hypothesis: measurement, and the ring drawn from the pose (radius = observation + noise of the component)
    nw, ns, scale = G["nh_weights"], float(G["nh_sigma"]), float(G["nh_scale"])
    nh = FF.BinaryFactorWithNullHypo(X, Ls[0], nw.copy(), FF.SE2R2RangeGaussianLikelihoodFactor, float(G["mix_obs"]), ns,
                                     null_sigma_scale=scale)
    ref = _component_stats(G["nh_obs"], G["ada_pose"], [cands[0], cands[0]], G["nh_counts"])
    assert abs(ref[0][1] / ns - 1) < 0.1 and abs(ref[1][1] / (ns * scale) - 1) < 0.1
    np.random.multinomial = spy
    try:
        obs = nh.sample(var1=G["ada_pose"], var2=cands[0])
        c_obs = counts
        ring = nh.sample(var1=G["ada_pose"], var2=None)
        c_ring = counts
    finally:
        np.random.multinomial = orig
    ours = _component_stats(obs, G["ada_pose"], [cands[0], cands[0]], c_obs)
    for k in range(2):
        assert abs(c_obs[k] / N - nw[k]) < 4 * np.sqrt(nw[k] * (1 - nw[k]) / N)
        assert abs(ours[k][1] / ref[k][1] - 1) < 0.12 and abs(ours[k][0]) < 5 * ref[k][1] / np.sqrt(c_obs[k])

    def ring_radius_stats(ring, cnt):
        out, lo = [], 0
        for c in cnt:
            r = np.linalg.norm(ring[lo:lo + c] - G["ada_pose"][lo:lo + c, :2], axis=1) - float(G["mix_obs"])
            out.append((r.mean(), r.std()))
            lo += c
        return out
    rr, ro = ring_radius_stats(G["nh_ring"], G["nh_ring_counts"]), ring_radius_stats(ring, c_ring)
    for k in range(2):
        assert abs(ro[k][1] / rr[k][1] - 1) < 0.12 and abs(ro[k][0] - rr[k][0]) < 5 * rr[k][1] / np.sqrt(c_ring[k])


def test_r2_factor_family_matches_reference_bodies(monkeypatch):
    """The toy range-only examples' factor types (BASELINE config[2]): displacement factor in its three directions and the
    R2-R2 range factor, with the reference's draws injected (src/factors/Factors.py:998-1030, 2080-2135)."""
    a, b = R2Variable("x0"), R2Variable("l1", VariableType.Landmark)
    monkeypatch.setattr(FF, "_gaussian_noise", lambda chol, n: G["r2_noise2"].copy())
    rel = FF.R2RelativeGaussianLikelihoodFactor(a, b, np.array([5.0, -5.0]), precision=np.eye(2) * 10)
    np.testing.assert_allclose(rel.sample(var1=G["r2_p1"]), G["r2rel_fwd"], atol=1e-12)
    np.testing.assert_allclose(rel.sample(var2=G["r2_p2"]), G["r2rel_bwd"], atol=1e-12)
    np.testing.assert_allclose(rel.sample(var1=G["r2_p1"], var2=G["r2_p2"]), G["r2rel_meas"], atol=1e-12)
    sigma = 0.7
    rng_f = FF.R2RangeGaussianLikelihoodFactor(a, b, 12.0, sigma)
    monkeypatch.setattr(np.random, "standard_normal", lambda n: G["f_noise1"][:, 0] / sigma)
    monkeypatch.setattr(np.random, "uniform", lambda lo, hi, n: G["ring_angles"].copy())
    np.testing.assert_allclose(rng_f.sample(var1=G["r2_p1"]), G["r2ring"], atol=1e-11)
    np.testing.assert_allclose(rng_f.sample(var1=G["r2_p1"], var2=G["r2_p2"]), G["r2range_meas"], atol=1e-11)
    # text forms round-trip through the `.fg` reader (reference :421-441, :493-507, :975-992, :2063-2079)
    prior = FF.UnaryR2GaussianPriorFactor(b, np.array([5.0, 5.0]), covariance=np.eye(2) * 0.5)
    ring = FF.UnaryR2RangeGaussianPriorFactor(a, np.array([1.0, 2.0]), 7.0, 0.5)
    for f in (prior, ring, rel, rng_f):
        g2 = FF.Factor.construct_from_text(str(f), [a, b])
        assert type(g2) is type(f) and str(g2) == str(f)
    # the ring prior draws radius ~ N(mu, sigma^2) around the centre (src/stats/Distributions.py:125-130)
    monkeypatch.undo()
    np.random.seed(0)
    s = ring.sample(20000)
    r = np.linalg.norm(s - np.array([1.0, 2.0]), axis=1)
    assert abs(r.mean() - 7.0) < 0.02 and abs(r.std() - 0.5) < 0.02
    s = prior.sample(20000)
    assert np.abs(s.mean(0) - 5.0).max() < 0.03 and np.abs(np.cov(s.T) - 0.5 * np.eye(2)).max() < 0.03
