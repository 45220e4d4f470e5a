"""GPU parity tests: the hand-written gfx950 kernels, called through the C ABI
(include/nfisam_hip.h via the ctypes binding), against
  (1) golden vectors produced by the reference itself (tests/golden/*.npz) and
  (2) the oracle (oracle/: C restatement in float64, torch restatement) on seeded inputs.

Tolerances (the kernels use 1-ulp hardware exp2/log2/rcp, fp32 throughout):
  z / x            1e-4 abs
  log-det / loss   2e-4 abs
  gradients        1e-3 rel + 2e-5 abs (of the n-normalised gradient)
  Adam trajectory  after 10 steps: 2e-3 abs on parameters
"""
import glob
import json
import os
import time

import numpy as np
import pytest
import torch

import nfisam_hip as nh
from oracle import c_oracle as CO
from oracle import nsf_torch as O

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
CASES = sorted(glob.glob(os.path.join(GOLDEN, "nsf_*.npz")))
DEV = "cuda:0"
Z_ATOL, LD_ATOL = 1e-4, 2e-4
NOISE_GRAD = 2e-5        # |reference gradient| below the kernels' absolute gradient tolerance: sign of an Adam step undetermined


def load(path):
    g = dict(np.load(path))
    n, D, K, H, seed = [int(v) for v in g["meta"]]
    return g, n, D, K, H, float(g["B"])


def sd_of(g, prefix):
    return {k[len(prefix) + 2:].replace("__", "."): v for k, v in g.items() if k.startswith(prefix + "__")}


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32).to(DEV)


def kpack(blob_np, D, K, H, L=1):
    return nh.pack(dev(blob_np), D, K, H, L)


def grad_close(got, ref, rtol=1e-3, atol=2e-5):
    np.testing.assert_allclose(got, ref, rtol=rtol, atol=atol * max(1.0, float(np.abs(ref).max())))


@pytest.fixture(params=["auto", "wide", "split"])
def family(request):
    """Training-kernel family: chosen per launch by the library ("auto": two lanes per particle for these
    small launches), or forced through NFISAM_TRAIN (read per call)."""
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


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[4:-4] for p in CASES])
class TestAgainstReferenceGolden:
    def test_forward(self, path):
        g, n, D, K, H, B = load(path)
        kp = kpack(O.blob_from_state_dict(sd_of(g, "p0"), D), D, K, H)
        z, ld, lp = nh.forward(dev(g["x"]), kp, K, H, B, want_logprob=True)
        np.testing.assert_allclose(z.cpu().numpy(), g["z"], atol=Z_ATOL)
        np.testing.assert_allclose(ld.cpu().numpy(), g["logdet"], atol=LD_ATOL)
        np.testing.assert_allclose((lp - ld).cpu().numpy(), g["prior_logprob"], atol=LD_ATOL, rtol=1e-5)
        # the reference's own (scrambled) return value is the fixed permutation of ours
        np.testing.assert_allclose(z.cpu().numpy().T.reshape(-1).reshape(n, D), g["z_raw"], atol=Z_ATOL)

    def test_nll_gradients(self, path, family):
        g, n, D, K, H, B = load(path)
        kp = kpack(O.blob_from_state_dict(sd_of(g, "p0"), D), D, K, H)
        kg, _, loss = nh.backward(dev(g["x"]), kp, K, H, B, nll_mode=True)
        loss = loss.item() / n + 0.5 * D * np.log(2 * np.pi)
        assert abs(loss - float(g["loss"])) < 2e-4
        grad = nh.unpack(kg, D, K, H).cpu().numpy() / n
        grad_close(grad, O.blob_from_state_dict(sd_of(g, "g0"), D))
        # padding entries of the kernel layout never receive gradient
        assert float(kg[torch.from_numpy(nh.layout_map(D, K, H) < 0).to(DEV)].abs().sum()) == 0.0

    def test_adam_trajectory(self, path, family):
        g, n, D, K, H, B = load(path)
        kp = kpack(O.blob_from_state_dict(sd_of(g, "p0"), D), D, K, H)
        tb = nh.TrainBatch([dev(g["x"])], [kp], K, H, B, 1, lr=float(g["adam_lr"]), max_iters=10, early_stop=False)
        snaps = {}
        for it in range(10):
            tb.step()
            if it + 1 in (1, 2, 10):
                snaps[it + 1] = nh.unpack(tb.kparams[0], D, K, H).cpu().numpy()
        torch.cuda.synchronize()
        assert tb.state()["step"] == 10 and tb.state()["stop"] == 0
        np.testing.assert_allclose(tb.iter_loss[0].cpu().numpy(), g["adam_losses"], atol=5e-4, rtol=1e-4)
        lr = float(g["adam_lr"])
        # Adam divides every coordinate's step by sqrt(v): a coordinate whose gradient is at fp32-noise level (an empty
        # spline bin -- K = 15 bins on 64 particles) moves by +-lr per step in EITHER implementation, so such coordinates
        # may differ by O(steps x lr).  They are identified from the REFERENCE's own gradient at the start (|g0| below
        # the kernels' absolute gradient tolerance, where the sign of g / sqrt(v) is not determined); every other
        # coordinate is held to the strict tolerance, and nothing may be off by more than the steps taken.
        g0 = np.abs(O.blob_from_state_dict(sd_of(g, "g0"), D))
        noisy = g0 < NOISE_GRAD
        # measured per golden case (fraction of exempt coordinates): 0 (d1_k9, d5_k3_h4) .. 0.0204 (n64_d4_k15: 15 bins on 64
        # particles); the guard is twice the largest, so that a misplaced gradient row cannot hide in the exemption
        print("exempt (noise-level reference gradient) fraction of %s: %.4f" % (os.path.basename(path), noisy.mean()))
        assert noisy.mean() <= 2 * 0.0204, noisy.mean()
        for steps, atol in ((1, 2e-4), (2, 4e-4), (10, 2e-3)):
            ref = O.blob_from_state_dict(sd_of(g, "p%d" % steps), D)
            err = np.abs(snaps[steps] - ref)
            excess = err - 1e-3 * np.abs(ref)
            bad = excess > atol
            assert not np.any(bad & ~noisy), (steps, int((bad & ~noisy).sum()), float(excess[~noisy].max()),
                                               float(g0[bad & ~noisy].min()))
            assert np.quantile(excess, 0.98) <= atol, (steps, float(np.quantile(excess, 0.98)))
            assert err.max() < steps * lr + 1e-4, (steps, err.max())
        tb.step()   # beyond max_iters: must be a no-op
        torch.cuda.synchronize()
        assert tb.state()["step"] == 10
        # padding of the kernel layout (output-column padding; the zero hidden units of a hidden_dim that is not a compiled
        # width: goldens h6 / h10 / h12) never moves: parameters and both moments stay exactly 0 there
        pad = torch.from_numpy(nh.layout_map(D, K, H) < 0).to(DEV)
        for t in (tb.kparams[0], tb.m[0], tb.v[0]):
            assert float(t[pad].abs().sum()) == 0.0

    def test_inverse(self, path):
        g, n, D, K, H, B = load(path)
        kp = kpack(O.blob_from_state_dict(sd_of(g, "p0"), D), D, K, H)
        x, ld = nh.inverse(dev(g["zlat"]), None, kp, K, H, B, want_logdet=True)
        np.testing.assert_allclose(x.cpu().numpy(), g["zlat_inv_x"], atol=2e-4)
        np.testing.assert_allclose(ld.cpu().numpy(), g["zlat_inv_logdet"], atol=3e-4)
        x2 = nh.inverse(dev(g["z"]), None, kp, K, H, B)
        np.testing.assert_allclose(x2.cpu().numpy(), g["inv_x"], atol=2e-4)

    def test_inverse_given_separator(self, path):
        g, n, D, K, H, B = load(path)
        kp = kpack(O.blob_from_state_dict(sd_of(g, "p0"), D), D, K, H)
        for Ds in (1, 3):
            if "igs%d_x" % Ds not in g:
                continue
            xf = nh.inverse(dev(g["zlat"][:, Ds:]), dev(g["x"][:, :Ds]), kp, K, H, B)
            np.testing.assert_allclose(xf.cpu().numpy(), g["igs%d_x" % Ds], atol=2e-4)


def make_problem(n, D, K, H, L, seed, spread=1.6):
    gen = torch.Generator().manual_seed(seed)
    blob = torch.cat([O.init_blob(D, K, H, gen) for _ in range(L)])
    blob = blob + 0.25 * torch.randn(blob.shape, generator=gen)
    x = spread * torch.randn(n, D, generator=gen)
    for r, (c, v) in enumerate([(0, 5.5), (D - 1, -5.0), (D // 2, 5.0), (0, 4.9999)]):
        if r < n:
            x[r, c] = v
    return blob.numpy().astype(np.float32), x.numpy().astype(np.float32)


@pytest.mark.parametrize("n,D,K,L", [(100, 5, 6, 2), (333, 6, 9, 4), (70, 12, 9, 3), (64, 1, 9, 2), (129, 20, 5, 1),
                                     (1, 4, 9, 2), (33, 3, 12, 1), (90, 30, 9, 2)])
def test_multilayer_against_oracle(n, D, K, L, family):
    """No usable multi-layer reference exists: parity is against the oracle.  (1, 4, 9, 2): a single
    particle; (33, 3, 12, 1): one particle past a 32-tile; (90, 30, 9, 2): the parameters of all layers do not fit
    in LDS next to the tiles, so the two-lane kernel reads them from global memory.)"""
    H, B = 8, 5.0
    blob, x = make_problem(n, D, K, H, L, seed=n + D)
    kp = kpack(blob, D, K, H, L)
    zc, ldc = CO.forward(x, blob, K, H, B, L, dtype=np.float64)
    z, ld, _ = nh.forward(dev(x), kp, K, H, B, L)
    np.testing.assert_allclose(z.cpu().numpy(), zc, atol=Z_ATOL * L)
    np.testing.assert_allclose(ld.cpu().numpy(), ldc, atol=LD_ATOL * L)
    # NLL gradients incl. d/dx
    lossc, gradc, lpc, gxc = CO.nll_grad(x, blob, K, H, B, L, dtype=np.float64, want_gx=True)
    kg, gx, loss = nh.backward(dev(x), kp, K, H, B, L, nll_mode=True, want_gx=True)
    assert abs(loss.item() / n + 0.5 * D * np.log(2 * np.pi) - lossc) < 3e-4 * L
    # fp32 rounding grows with the number of chained layers (absolute tolerance relative to the largest entry)
    grad_close(nh.unpack(kg, D, K, H, L).cpu().numpy() / n, gradc, rtol=2e-3, atol=2e-5 * L)
    grad_close(gx.cpu().numpy() / n, gxc, rtol=2e-3, atol=2e-5 * L)
    # generic VJP
    rng = np.random.RandomState(n)
    gz = rng.randn(n, D).astype(np.float32); gl = rng.randn(n).astype(np.float32)
    grad4, gx4 = CO.backward(x, blob, gz, gl, K, H, B, L, dtype=np.float64)
    kg2, gx2, _ = nh.backward(dev(x), kp, K, H, B, L, gz=dev(gz), gl=dev(gl), want_gx=True)
    grad_close(nh.unpack(kg2, D, K, H, L).cpu().numpy(), grad4, rtol=2e-3, atol=5e-5 * max(1, L - 1))
    grad_close(gx2.cpu().numpy(), gx4, rtol=2e-3, atol=5e-5 * max(1, L - 1))
    # inverse of forward
    xb, ldb = nh.inverse(z, None, kp, K, H, B, L, want_logdet=True)
    inside = np.abs(x).max(1) < 4.9
    # autoregressive inversion propagates fp32 error of early dims into later ones (through the
    # conditioners) and divides by the local slope: the round trip is checked at 3e-3 (growing with the
    # length of the chain beyond 10 dims), the direct comparisons above and below at 1e-4
    chain = max(1.0, D / 10.0)
    np.testing.assert_allclose(xb.cpu().numpy()[inside], x[inside], atol=3e-3 * chain)
    xo, _ = CO.inverse(zc, None, blob, K, H, B, L, dtype=np.float64)
    xh = nh.inverse(dev(zc.astype(np.float32)), None, kp, K, H, B, L)
    np.testing.assert_allclose(xh.cpu().numpy()[inside], xo[inside], atol=3e-3 * chain)
    # round trip of L*D chained splines: reconstruction error times |d logdet/dx| accumulates
    np.testing.assert_allclose((ld + ldb).cpu().numpy()[inside], 0, atol=2e-3 * L * chain)


def test_conditional_sampling_with_fused_normalisation():
    """nfisam_nsf_inverse == normalize_samples -> inverse_given_separator -> unnormalize_samples
    (src/slam/NFiSAM.py:96-118,140-155)."""
    n, D, Ds, K, H, B, L = 500, 9, 4, 9, 8, 5.0, 1
    blob, _ = make_problem(8, D, K, H, L, seed=3)
    rng = np.random.RandomState(0)
    circ = np.array([0, 0, 1, 0, 0, 1, 0, 0, 1], dtype=bool)
    mean = rng.randn(D).astype(np.float32) * 2; std = (0.3 + rng.rand(D)).astype(np.float32)
    mean[circ] = np.array([3.0, -2.5, 0.4], dtype=np.float32)
    xs_raw = (rng.randn(n, Ds) * 2 + mean[:Ds]).astype(np.float32)
    z = (1.3 * rng.randn(n, D - Ds)).astype(np.float32)
    xs_n = O.normalize_samples(xs_raw, mean, std, circ, 0)
    xf_n, _ = CO.inverse(z, xs_n, blob, K, H, B, L, dtype=np.float64)
    ref = O.unnormalize_samples(xf_n.astype(np.float32), mean, std, circ, Ds)
    got = nh.inverse(dev(z), dev(xs_raw), kpack(blob, D, K, H, L), K, H, B, L, mean=dev(mean), std=dev(std),
                     circular=torch.from_numpy(circ.astype(np.uint8)).to(DEV)).cpu().numpy()
    d = got - ref
    cidx = np.where(circ[Ds:])[0]
    d[:, cidx] = (d[:, cidx] + np.pi) % (2 * np.pi) - np.pi      # compare angles on the circle
    assert np.abs(d).max() < 5e-4
    assert np.all(np.abs(got[:, cidx]) <= np.pi + 1e-6)


@pytest.mark.parametrize("use_graph", [False, True], ids=["eager", "hipgraph"])
def test_training_loop_early_stop_matches_oracle(use_graph):
    n, D, K, H, B, L = 1000, 3, 5, 8, 5.0, 1
    gen = torch.Generator().manual_seed(5)
    blob = O.init_blob(D, K, H, gen).numpy()
    x = torch.randn(n, D, generator=gen)
    x[:, 1] = x[:, 0] ** 2 - 1 + 0.3 * x[:, 1]
    x = ((x - x.mean(0)) / x.std(0)).numpy().astype(np.float32)
    bc, lc, ic, _, _ = CO.train(x, blob, K, H, B, L, lr=0.03, max_iters=600, average_window=20,
                                loss_delta_tol=5e-3, dtype=np.float32)
    tb = nh.TrainBatch([dev(x)], [kpack(blob, D, K, H)], K, H, B, L, lr=0.03, max_iters=600, average_window=20,
                       loss_delta_tol=5e-3)
    iters = tb.run(use_graph=use_graph)
    il = tb.iter_loss[0].cpu().numpy()
    assert iters[0] % 20 == 0 and 40 <= iters[0] < 600
    assert np.all(il[iters[0]:] == 0) and np.all(il[:iters[0]] != 0)     # zero-padded like the reference
    # trajectories agree while both run; the stop decision may differ by one window at the tolerance edge
    m = min(ic, iters[0])
    np.testing.assert_allclose(il[:m], lc[:m], atol=5e-3)
    assert abs(iters[0] - ic) <= 20
    assert il[iters[0] - 1] < il[0] - 0.05


def test_batched_ragged_cliques_match_single_clique_runs():
    K, H, B, L, iters = 9, 8, 5.0, 1, 7
    shapes = [(200, 6), (64, 8), (129, 3), (1, 5), (333, 12)]
    xs, blobs = [], []
    for c, (n, D) in enumerate(shapes):
        b, x = make_problem(n, D, K, H, L, seed=40 + c, spread=1.0)
        xs.append(x); blobs.append(b)
    tb = nh.TrainBatch([dev(x) for x in xs], [kpack(b, D, K, H) for b, (n, D) in zip(blobs, shapes)], K, H, B, L,
                       lr=0.02, max_iters=iters, early_stop=False)
    for _ in range(iters):
        tb.step()
    torch.cuda.synchronize()
    for c, (n, D) in enumerate(shapes):
        bc, lc, ic, _, _ = CO.train(xs[c], blobs[c], K, H, B, L, lr=0.02, max_iters=iters, early_stop=False,
                                    dtype=np.float32)
        np.testing.assert_allclose(tb.iter_loss[c].cpu().numpy(), lc, atol=5e-4, rtol=2e-4)
        # Adam normalises every coordinate by sqrt(v): a coordinate whose gradient is at fp32 noise
        # level moves by ~lr per step in either implementation, so a handful of such coordinates may
        # differ by O(iters*lr); everything else must agree tightly.
        got = nh.unpack(tb.kparams[c], D, K, H).cpu().numpy()
        err = np.abs(got - bc)
        assert np.quantile(err, 0.99) < 2e-3, (c, np.quantile(err, 0.99))
        assert err.max() < iters * 0.02 + 1e-3


def test_full_size_properties_config_c2():
    """BASELINE config 2 (n=4096, D=6, L=4, K=9): size-independent properties."""
    n, D, K, H, B, L = 4096, 6, 9, 8, 5.0, 4
    blob, x = make_problem(n, D, K, H, L, seed=99, spread=1.0)
    kp = kpack(blob, D, K, H, L)
    xd = dev(x)
    z, ld, lp = nh.forward(xd, kp, K, H, B, L, want_logprob=True)
    xb, ldb = nh.inverse(z, None, kp, K, H, B, L, want_logdet=True)
    inside = (xd.abs().max(1).values < 4.9)
    assert float((xb - xd)[inside].abs().max()) < 1e-2
    assert float((xb - xd)[inside].abs().mean()) < 2e-5
    assert float((ld + ldb)[inside].abs().max()) < 2e-2
    assert float((ld + ldb)[inside].abs().mean()) < 1e-4
    # loss from backward == -mean(logprob) from forward; gradient is deterministic up to atomics order
    kg, _, loss = nh.backward(xd, kp, K, H, B, L, nll_mode=True)
    assert abs(loss.item() / n + 0.5 * D * np.log(2 * np.pi) + lp.mean().item()) < 1e-3
    kg2, _, _ = nh.backward(xd, kp, K, H, B, L, nll_mode=True)
    assert float((kg - kg2).abs().max()) <= 1e-3 * float(kg.abs().max())
    # 30 training iterations reduce the loss
    tb = nh.TrainBatch([xd], [kp.clone()], K, H, B, L, lr=0.02, max_iters=30, early_stop=False)
    assert tb.run(use_graph=True) == [30]
    il = tb.iter_loss[0].cpu().numpy()
    assert np.all(np.isfinite(il)) and il[-1] < il[0]


def test_argument_errors_are_loud():
    kp = torch.zeros(nh.kparam_count(3, 9, 8), device=DEV)
    x = torch.zeros(4, 3, device=DEV)
    with pytest.raises(ValueError):
        nh.forward(x, kp, 7, 8, 5.0)          # unsupported K -> size mismatch / ERR_ARG
    with pytest.raises(ValueError):
        nh.forward(x, kp[:-4], 9, 8, 5.0)
    with pytest.raises(ValueError):
        nh.inverse(x, x, kp, 9, 8, 5.0)       # D would be 6: wrong blob size


def _walk_tree_problem(L, K, H, rng):
    """Three cliques: root {v0,v1} -> child {v2 | v1} -> leaf {v3 | v2, part of v0}; sample-matrix columns permuted."""
    total = 9
    specs = [dict(n_obs=0, sep=[], front=[4, 5, 6, 7, 8]),                 # root: joint of v0, v1
             dict(n_obs=2, sep=[7, 8], front=[1, 2, 3]),                   # obs(2) | v1 -> v2
             dict(n_obs=1, sep=[1, 2, 3, 4, 5], front=[0])]                # obs(1) | v2, part of v0 -> v3
    entries, host = [], []
    for sp in specs:
        D = sp["n_obs"] + len(sp["sep"]) + len(sp["front"]) + (1 if sp is specs[1] else 0)   # one model is larger than used
        blob, _ = make_problem(8, D, K, H, L, seed=int(rng.randint(1000)))
        mean = (rng.randn(D) * 2).astype(np.float32); std = (0.5 + rng.rand(D)).astype(np.float32)
        circ_np = (rng.rand(D) < 0.3)
        obs = rng.randn(sp["n_obs"])
        entries.append(dict(kparams=kpack(blob, D, K, H, L), mean=dev(mean), std=dev(std),
                            circular=torch.from_numpy(circ_np.astype(np.uint8)).to(DEV), D_model=D, obs=obs,
                            sep_cols=sp["sep"], front_cols=sp["front"]))
        host.append((blob, mean, std, circ_np, D, obs))
    return total, specs, entries, host


def _oracle_walk(specs, host, Zt, n, total, K, H, B, L):
    """FactorGraphSolver.sample_posterior (src/slam/FactorGraphSolver.py:497-550) with the float64 oracle: per clique
    normalise the given columns, conditional inverse (truncated flow = first Ds+F dims of the model), un-normalise."""
    ref = np.zeros((n, total))
    zrow = 0
    for sp, (blob, mean, std, circ, D, obs) in zip(specs, host):
        Ds, F = sp["n_obs"] + len(sp["sep"]), len(sp["front"])
        given = np.concatenate([np.tile(obs, (n, 1)), ref[:, sp["sep"]]], 1) if Ds else None
        xs_n = None
        if Ds:
            xs_n = O.normalize_samples(given.astype(np.float32), mean, std, circ, 0).astype(np.float64)
        # the first Ds+F dims of a D-dim autoregressive flow are its (Ds+F)-dim marginal flow: truncate the blob
        Dt = Ds + F
        Pt, P = O.param_count(Dt, K, H), O.param_count(D, K, H)
        tb = np.concatenate([blob[l * P:l * P + Pt] for l in range(L)])
        z = Zt[zrow:zrow + F, :].T
        zrow += F
        xf, _ = CO.inverse(z, xs_n, tb, K, H, B, L, dtype=np.float64)
        ref[:, sp["front"]] = O.unnormalize_samples(xf.astype(np.float32), mean, std, circ, Ds)
    return ref


@pytest.mark.parametrize("L,H", [(1, 8), (2, 8), (1, 4), (1, 16), (2, 16)])
def test_posterior_tree_walk_matches_float64_oracle_chain(L, H):
    """nfisam_nsf_posterior_walk against the ORACLE: a chain of float64 `CO.inverse` calls with the same latent draws
    (FactorGraphSolver.sample_posterior semantics, src/slam/FactorGraphSolver.py:497-550).  Latent rows are consumed
    in walk order; the destination columns are permuted so that the two orders differ.  Both kernels (pipelined
    two-lane walk for L = 1 -- every hidden width --, plain walk) and the per-clique `nfisam_nsf_inverse` path are checked."""
    K, B, n = 9, 5.0, 300
    rng = np.random.RandomState(10 + L + (H if H != 8 else 0))
    total, specs, entries, host = _walk_tree_problem(L, K, H, rng)
    Zt_np = rng.randn(total, n).astype(np.float32)
    Zt = torch.from_numpy(Zt_np).to(DEV)
    ref = _oracle_walk(specs, host, Zt_np.astype(np.float64), n, total, K, H, B, L)
    circ_cols = np.zeros(total, dtype=bool)
    for sp, (_, _, _, circ, D, _) in zip(specs, host):
        Ds = sp["n_obs"] + len(sp["sep"])
        circ_cols[sp["front"]] = circ[Ds:Ds + len(sp["front"])]

    def close(a):
        # a child conditions on its parent's samples: fp32 rounding is amplified by the conditioners' slopes along
        # the chain (autoregressive inversion), so the bulk is checked at 2e-4 and the tail at 5e-3
        d = a - ref
        d[:, circ_cols] = (d[:, circ_cols] + np.pi) % (2 * np.pi) - np.pi
        err = np.abs(d)
        assert np.quantile(err, 0.99) < 2e-4 * L and err.max() < 5e-3 * L, (np.quantile(err, 0.99), err.max())

    S = nh.posterior_walk(entries, total, n, K, H, B, L, DEV, Zt=Zt)
    assert S.shape == (n, total)
    close(S.cpu().numpy().astype(np.float64))
    os.environ["NFISAM_WALK"] = "plain"
    try:
        S1 = nh.posterior_walk(entries, total, n, K, H, B, L, DEV, Zt=Zt)
    finally:
        del os.environ["NFISAM_WALK"]
    close(S1.cpu().numpy().astype(np.float64))
    if L == 1 and H % 8 == 0:
        assert not torch.equal(S1, S)      # two different kernels (rounding differs somewhere)
    # the per-clique path (one nfisam_nsf_inverse call per clique, what FlowsPriorFactor.sample uses)
    per = torch.zeros(n, total, device=DEV)
    zrow = 0
    for sp, e in zip(specs, entries):
        given = []
        if sp["n_obs"]:
            given.append(dev(np.tile(e["obs"], (n, 1))))
        if sp["sep"]:
            given.append(per[:, sp["sep"]])
        xs = torch.cat(given, 1).contiguous() if given else None
        z = Zt[zrow:zrow + len(sp["front"]), :].t().contiguous()
        zrow += len(sp["front"])
        per[:, sp["front"]] = nh.inverse(z, xs, e["kparams"], K, H, B, L, mean=e["mean"], std=e["std"],
                                         circular=e["circular"], model_D=e["D_model"])
    close(per.cpu().numpy().astype(np.float64))


@pytest.mark.parametrize("L,Ds", [(2, 3), (3, 1), (2, 0)])
def test_multilayer_conditional_inverse_round_trips(L, Ds):
    """L > 1 with given columns: layer l is conditioned on the given columns pushed through layers 0..l-1 (DESIGN.md
    §3.3; the reference's literal loop, src/slam/NFiSAM.py:151-152, inverts no composition).  Property:
    forward(cat(x_s, inverse_given_separator(z_f, x_s))) == (., z_f); and equality with the float64 oracle."""
    n, D, K, H, B = 257, 7, 9, 8, 5.0
    blob, x = make_problem(n, D, K, H, L, seed=70 + L, spread=1.0)
    kp = kpack(blob, D, K, H, L)
    rng = np.random.RandomState(L)
    zf = rng.randn(n, D - Ds).astype(np.float32)
    xs = x[:, :Ds].copy() if Ds else None
    xf = nh.inverse(dev(zf), dev(xs) if Ds else None, kp, K, H, B, L)
    full = torch.cat([dev(xs), xf], 1).contiguous() if Ds else xf
    z, _, _ = nh.forward(full, kp, K, H, B, L)
    inside = (full.abs().max(1).values < 4.9).cpu().numpy()
    err = np.abs(z.cpu().numpy()[:, Ds:] - zf)[inside]
    assert inside.sum() > n // 2 and np.quantile(err, 0.99) < 1e-3 and err.max() < 2e-2, (np.quantile(err, 0.99), err.max())
    xo, _ = CO.inverse(zf, xs, blob, K, H, B, L, dtype=np.float64)
    e2 = np.abs(xf.cpu().numpy() - xo)[inside]
    assert np.quantile(e2, 0.99) < 3e-4 * L and e2.max() < 1e-2, (np.quantile(e2, 0.99), e2.max())


def test_training_is_bitwise_reproducible_for_small_single_layer_launches():
    """L = 1 and <= 64 tiles: per-tile gradient slabs (plain stores) + fixed-order reduction in the Adam
    kernel => no float atomics on the gradient path => identical parameters run to run, eager or graph.
    (The loss record goes through 64 atomic slots and may differ in the last bits.  With L > 1 the
    cross-wave dL/dx accumulation uses LDS float atomics: a single backward call then agrees to ~1e-7
    relative run to run, and training trajectories separate at the rate of the problem's own sensitivity,
    scripts/repro_check.py.)"""
    K, H, B = 9, 8, 5.0
    for (n, D, L, exact) in ((2000, 11, 1, True), (700, 15, 1, True)):
        blob, x = make_problem(n, D, K, H, L, seed=5, spread=1.0)
        outs = []
        for rep in range(2):
            tb = nh.TrainBatch([dev(x)], [kpack(blob, D, K, H, L)], K, H, B, L, lr=0.02, max_iters=40,
                               early_stop=False)
            assert tb.run(use_graph=bool(rep)) == [40]
            outs.append(tb.kparams[0].clone())
            il = tb.iter_loss[0].cpu().numpy()
            assert il[-1] < il[0]
        if exact:
            assert torch.equal(outs[0], outs[1]), (n, D, L, float((outs[0] - outs[1]).abs().max()))
        else:
            # Adam turns rounding-level gradient differences into +-lr steps on a few coordinates
            assert float(torch.quantile((outs[0] - outs[1]).abs(), 0.99)) < 2e-3


@pytest.mark.parametrize("n", [64 * 140, 64 * 300], ids=["split-kernel", "wide-kernel"])
def test_large_launch_uses_atomics_and_still_matches_oracle(n):
    """> 128 tiles: single gradient buffer + float atomics (workspace is one copy).  The smaller case still
    runs the two-lanes-per-particle kernel (<= 1280 waves), the larger one the one-lane-per-particle kernel."""
    K, H, B, L, D = 9, 8, 5.0, 1, 4
    ring = 128 * 128 + 64      # per-iteration loss sums behind the gradient copies (+ 64 reserved words)
    assert nh.lib().nfisam_nsf_grad_workspace_count(n, D, K, H, L) == nh.kparam_count(D, K, H) + ring
    # <= 32 tiles of 64 particles: room for the fused-Adam launches' second set of copies and second (theta | m | v), and for
    # the chunk-persistent form's two sets of tagged copies (round 6: up to sixteen 128-particle blocks per group, nsf_half.h)
    assert nh.lib().nfisam_nsf_grad_workspace_count(2000, D, K, H, L) == (63 + 32 + 3 + 64 + 2) * nh.kparam_count(D, K, H) + ring
    assert nh.lib().nfisam_nsf_grad_workspace_count(6000, D, K, H, L) == 94 * nh.kparam_count(D, K, H) + ring
    blob, x = make_problem(n, D, K, H, L, seed=8, spread=1.0)
    tb = nh.TrainBatch([dev(x)], [kpack(blob, D, K, H, L)], K, H, B, L, lr=0.02, max_iters=5, early_stop=False)
    for _ in range(5):
        tb.step()
    torch.cuda.synchronize()
    bc, lc, _, _, _ = CO.train(x, blob, K, H, B, L, lr=0.02, max_iters=5, early_stop=False, dtype=np.float32)
    np.testing.assert_allclose(tb.iter_loss[0].cpu().numpy(), lc, atol=5e-4, rtol=2e-4)
    err = np.abs(nh.unpack(tb.kparams[0], D, K, H).cpu().numpy() - bc)
    assert np.quantile(err, 0.99) < 2e-3


def test_empty_batch_and_domain_error():
    """n = 0 is a no-op (NFISAM_OK); a non-finite training batch ends the run with NFISAM_ERR_DOMAIN
    (reference: `ValueError("Input outside domain")` / the discriminant assert, src/flows/utils.py:74-76,133)."""
    K, H, B, L, D = 9, 8, 5.0, 1, 4
    blob, x = make_problem(64, D, K, H, L, seed=1)
    kp = kpack(blob, D, K, H, L)
    z, ld, _ = nh.forward(torch.zeros(0, D, device=DEV), kp, K, H, B, L)
    assert z.shape == (0, D) and ld.shape == (0,)
    xi = nh.inverse(torch.zeros(0, D, device=DEV), None, kp, K, H, B, L)
    assert xi.shape == (0, D)
    bad = x.copy()
    bad[3, 1] = np.nan
    tb = nh.TrainBatch([dev(bad)], [kp.clone()], K, H, B, L, lr=0.02, max_iters=100, average_window=50)
    with pytest.raises(RuntimeError, match="domain"):
        tb.run(use_graph=True)
    st = tb.state()
    assert st["domain_err"] == 1 and st["stop"] == 1 and st["step"] == 1      # stopped at the first bad iteration


def test_posterior_walk_single_clique_and_wide_root():
    """Edge cases of the pipelined walk: a tree of one clique (no prefetch stage at all) and a root whose parameter
    range exceeds what one wave keeps in flight (8 float4 per lane): the remainder is loaded synchronously."""
    K, H, B, L, n = 9, 8, 5.0, 1, 77
    rng = np.random.RandomState(3)
    for D, n_obs in ((16, 0), (5, 2)):
        blob, _ = make_problem(8, D, K, H, L, seed=D)
        kp = kpack(blob, D, K, H, L)
        mean = dev(rng.randn(D)); std = dev(0.5 + rng.rand(D))
        circ = torch.from_numpy((rng.rand(D) < 0.3).astype(np.uint8)).to(DEV)
        obs = rng.randn(n_obs)
        F = D - n_obs
        cols = list(rng.permutation(F))
        leaf = dict(kparams=kp, mean=mean, std=std, circular=circ, D_model=D, obs=obs, sep_cols=[], front_cols=cols)
        # the second tree puts a small clique in front of the wide one, so that the wide one arrives through the prefetch path
        small_blob, _ = make_problem(8, 2, K, H, L, seed=99)
        small = dict(kparams=kpack(small_blob, 2, K, H, L), mean=dev(np.zeros(2)), std=dev(np.ones(2)),
                     circular=torch.zeros(2, dtype=torch.uint8, device=DEV), D_model=2, obs=np.zeros(0), sep_cols=[],
                     front_cols=[F, F + 1])
        for entries, total, zrow in (([leaf], F, 0), ([small, leaf], F + 2, 2)):
            Zt = torch.from_numpy(rng.randn(total, n).astype(np.float32)).to(DEV)
            S = nh.posterior_walk(entries, total, n, K, H, B, L, DEV, Zt=Zt)
            given = dev(np.tile(obs, (n, 1))) if n_obs else None
            ref = nh.inverse(Zt[zrow:zrow + F].t().contiguous(), given, kp, K, H, B, L, mean=mean, std=std, circular=circ,
                             model_D=D)
            err = np.abs(S[:, cols].cpu().numpy() - ref.cpu().numpy())
            assert np.quantile(err, 0.99) < 2e-5 and err.max() < 1e-3, (D, len(entries), err.max())


def test_kernel_families_agree_on_random_shapes():
    """Sweep of random (n, D, K, L) shapes, slab and atomics sinks: the two-lanes-per-particle and the one-lane-per-particle
    training kernels produce the same loss and gradients (and the same dL/dx), and both match the oracle on the
    smaller cases."""
    rng = np.random.RandomState(2024)
    old = os.environ.get("NFISAM_TRAIN")
    try:
        for case in range(14):
            K = int(rng.choice([5, 6, 9, 12]))
            L = int(rng.choice([1, 1, 2, 3]))
            D = int(rng.randint(1, 19))
            n = int(rng.choice([1, 31, 32, 33, 64, 97, 200, 513, 1000]))
            H, B = 8, 5.0
            blob, x = make_problem(n, D, K, H, L, seed=1000 + case)
            kp = kpack(blob, D, K, H, L)
            res = {}
            for fam in ("split", "pair", "wide"):       # pair = nsf_train3_kernel forced on every shape it can hold (default: D >= 6)
                os.environ["NFISAM_TRAIN"] = "wide" if fam == "wide" else "split"
                os.environ["NFISAM_PAIR"] = "1" if fam == "pair" else "0"
                kg, gx, loss = nh.backward(dev(x), kp, K, H, B, L, nll_mode=True, want_gx=True)
                tb = nh.TrainBatch([dev(x)], [kp.clone()], K, H, B, L, lr=0.01, max_iters=2, early_stop=False)
                tb.step()
                torch.cuda.synchronize()
                res[fam] = (nh.unpack(kg, D, K, H, L).cpu().numpy() / n, gx.cpu().numpy() / n, loss.item() / n,
                            float(tb.iter_loss[0][0]), nh.unpack(tb.kparams[0], D, K, H, L).cpu().numpy())
            for a, b in ((res["split"], res["wide"]), (res["pair"], res["wide"])):
                scale = max(1.0, float(np.abs(b[0]).max()))
                assert np.abs(a[0] - b[0]).max() < 6e-5 * L * scale, (case, n, D, K, L)
                # dL/dx is per particle: a particle whose intermediate z sits on a knot may fall into neighbouring bins in the
                # two kernels (the slope of log dz/dx jumps there), so the bulk is checked tightly and the tail loosely
                gx_err, gx_max = np.abs(a[1] - b[1]), max(1.0, float(np.abs(b[1]).max()))
                assert np.quantile(gx_err, 0.98) < 6e-5 * L * gx_max and gx_err.max() < 0.05 * gx_max, (case, n, D, K, L)
                assert abs(a[2] - b[2]) < 2e-4 * L and abs(a[3] - b[3]) < 3e-4 * L, (case, a[2], b[2], a[3], b[3])
                # one Adam step from identical gradients (up to rounding): parameters move by +-lr at most, equally
                assert np.quantile(np.abs(a[4] - b[4]), 0.98) < 2e-3, (case, n, D, K, L)
                if n * D * L <= 4000:
                    lossc, gradc, _, gxc = CO.nll_grad(x, blob, K, H, B, L, dtype=np.float64, want_gx=True)
                    assert abs(a[2] + 0.5 * D * np.log(2 * np.pi) - lossc) < 3e-4 * L
                    grad_close(a[0], gradc, rtol=2e-3, atol=2e-5 * L)
                    ex = np.abs(a[1] - gxc)
                    assert np.quantile(ex, 0.98) < 2e-3 * np.abs(gxc).max() + 2e-5 * L and ex.max() < 0.05 * max(1.0, np.abs(gxc).max())
    finally:
        os.environ.pop("NFISAM_PAIR", None)
        if old is None:
            os.environ.pop("NFISAM_TRAIN", None)
        else:
            os.environ["NFISAM_TRAIN"] = old


class _Env:
    """Environment switches of the launch-shape helpers (read per call), restored on exit."""
    def __init__(self, **kv):
        self.kv, self.old = kv, {}

    def __enter__(self):
        for k, v in self.kv.items():
            self.old[k] = os.environ.get(k)
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v

    def __exit__(self, *exc):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _train_ragged(shapes, iters, window, use_graph, K=9, H=8, B=5.0, lr=0.01):
    probs = [make_problem(n, D, K, H, 1, seed=300 + c) for c, (n, D) in enumerate(shapes)]
    tb = nh.TrainBatch([dev(x) for _, x in probs], [kpack(b, D, K, H) for (b, _), (_, D) in zip(probs, shapes)], K, H, B, 1,
                       lr=lr, max_iters=iters, average_window=window, loss_delta_tol=0.0, early_stop=True)
    done = tb.run(use_graph=use_graph)
    torch.cuda.synchronize()
    return done, [[t.cpu().numpy().copy() for t in arr] for arr in (tb.kparams, tb.m, tb.v, tb.iter_loss)]


@pytest.mark.parametrize("shapes,iters,window", [([(2000, 15)], 130, 50), ([(1900, 12), (300, 17), (2048, 3), (65, 1)], 130, 50),
                                                 ([(512, 6)], 7, 1), ([(700, 9), (640, 9)], 9, 4)],
                         ids=["plaza-clique", "ragged-batch", "chunks-of-one", "odd-chunks"])
@pytest.mark.parametrize("use_graph", [True, False], ids=["hipgraph", "eager"])
@pytest.mark.parametrize("H", [8, 16], ids=["h8", "h16"])
def test_fused_adam_launches_equal_the_separate_adam_kernel_bit_for_bit(shapes, iters, window, use_graph, H):
    """The Adam update applied at the start of the next gradient launch (nsf_cond_mfma.h; chunk-closing update by
    nsf_adam_kernel) leaves exactly the parameters and moments of gradient kernel + Adam kernel per iteration: full
    chunks, a final partial chunk (130 = 50 + 50 + 30), chunks of one, odd chunk lengths (the two state buffers
    alternate with the iteration's parity), ragged batches with D > 16 and a single-tile clique."""
    with _Env(NFISAM_FUSED_ADAM="0"):
        d0, ref = _train_ragged(shapes, iters, window, use_graph, H=H)
    with _Env(NFISAM_FUSED_ADAM=None):
        d1, got = _train_ragged(shapes, iters, window, use_graph, H=H)
    assert d0 == d1 == [iters] * len(shapes)
    for name, a, b in zip(("theta", "m", "v"), ref[:3], got[:3]):
        for c, (x, y) in enumerate(zip(a, b)):
            assert np.array_equal(x, y), (name, c, np.abs(x - y).max())
    for x, y in zip(ref[3], got[3]):                        # the loss record goes through float atomics (order-dependent rounding)
        np.testing.assert_allclose(x, y, rtol=3e-6, atol=1e-6)


@pytest.mark.parametrize("shape", [[(2000, 15)], [(1900, 12), (300, 17), (700, 9)]], ids=["one-clique", "batch"])
def test_chunks_enqueued_ahead_of_an_early_stop_change_nothing(shape):
    """Training plans enqueue chunk k + 1 before the host has seen the outcome of chunk k (the states arrive in pinned host
    memory; the stop flags are checked on the device).  With an early stop the chunk behind it must be a no-op: same stop
    iteration, parameters, moments and loss record as launch - wait - launch (`NFISAM_RUN_AHEAD=0`); and the plan must be
    re-usable right away (the next run drains what is still in flight before it restarts the chunk sequence)."""
    K, H, B = 9, 8, 5.0
    probs = [make_problem(n, D, K, H, 1, seed=700 + c) for c, (n, D) in enumerate(shape)]

    def run(ahead):
        with _Env(NFISAM_RUN_AHEAD=ahead):
            tb = nh.TrainBatch([dev(x) for _, x in probs], [kpack(b, D, K, H) for (b, _), (_, D) in zip(probs, shape)], K, H, B, 1,
                               lr=0.01, max_iters=600, average_window=50, loss_delta_tol=0.05, early_stop=True)
            first = tb.run(use_graph=True)
            torch.cuda.synchronize()
            snap = [[t.cpu().numpy().copy() for t in arr] for arr in (tb.kparams, tb.m, tb.v, tb.iter_loss)]
            tb.reset([kpack(b, D, K, H) for (b, _), (_, D) in zip(probs, shape)])       # while a chunk may still be draining
            second = tb.run(use_graph=True)
            torch.cuda.synchronize()
            again = [t.cpu().numpy().copy() for t in tb.kparams]
            tb.close()
        return first, snap, second, again
    f0, s0, r0, a0 = run("0")
    f1, s1, r1, a1 = run(None)
    assert f0 == f1 == r0 == r1 and all(100 <= i < 600 and i % 50 == 0 for i in f0), (f0, f1, r0, r1)   # stopped early
    for a, b in zip(s0[:3], s1[:3]):
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
    for x, y in zip(s0[3], s1[3]):
        np.testing.assert_allclose(x, y, rtol=3e-6, atol=1e-6)
        assert np.all(x[f0[0]:] == 0) or len(shape) > 1
    for x, y, z in zip(a0, a1, s1[0]):
        assert np.array_equal(x, y) and np.array_equal(y, z)        # the re-run reproduces the first run


def test_gradient_parts_cover_the_gradient_launch():
    """`nfisam_nsf_train_gradient_part`: the launches of a split iteration write exactly the gradient copies and loss sums
    of the single launch (disjoint (clique, dim) groups), in any order; a launch shape that is not split refuses parts."""
    K, H, B = 9, 8, 5.0
    shapes = [(1900, 12), (300, 17), (700, 9)]
    probs = [make_problem(n, D, K, H, 1, seed=500 + c) for c, (n, D) in enumerate(shapes)]

    def batch():
        return nh.TrainBatch([dev(x) for _, x in probs], [kpack(b, D, K, H) for (b, _), (_, D) in zip(probs, shapes)], K, H, B, 1,
                             lr=0.01, max_iters=5, early_stop=False)
    ref = batch()
    ref.gradient_only()
    got = batch()
    for g in (2, 0, 1):
        got.gradient_part(g, 3)
    torch.cuda.synchronize()
    assert ref.chains() >= 1
    for (n, D), a, b in zip(shapes, ref.g, got.g):
        copies = ((n + 63) // 64 + 3) // 4 * nh.kparam_count(D, K, H)      # one gradient copy per block of four tiles
        assert torch.equal(a[:copies], b[:copies])                          # gradient copies: plain stores, bit for bit
        assert float(a[:copies].abs().sum()) > 0
        torch.testing.assert_close(a[copies:], b[copies:], rtol=1e-5, atol=1e-4)    # loss sums: float atomics (order-dependent rounding)
    lm = nh.TrainBatch([dev(probs[0][1])], [kpack(make_problem(1900, 12, K, H, 2, seed=1)[0], 12, K, H, 2)], K, H, B, 2, lr=0.01,
                       max_iters=5, early_stop=False)
    assert lm.chains() == 1
    with pytest.raises(Exception):
        lm.gradient_part(0, 2)


@pytest.mark.parametrize("chains", [2, 3, 5])
def test_parallel_graph_branches_leave_the_same_bits(chains):
    """`NFISAM_CHAINS=n`: an iteration split into n launches on parallel branches of the chunk's hipGraph (independent
    (clique, dim) groups: nsf_kernels.hip, plan creation) trains exactly the parameters and moments of the single-launch
    iteration -- ragged batch with D > 16, a single-tile clique, full + partial chunks, more chains than group octets."""
    shapes, iters, window = [(1900, 12), (300, 17), (2048, 3), (65, 1), (700, 9)], 130, 50
    with _Env(NFISAM_CHAINS="1"):
        d0, ref = _train_ragged(shapes, iters, window, True)
    with _Env(NFISAM_CHAINS=str(chains)):
        d1, got = _train_ragged(shapes, iters, window, True)
    assert d0 == d1 == [iters] * len(shapes)
    for name, a, b in zip(("theta", "m", "v"), ref[:3], got[:3]):
        for c, (x, y) in enumerate(zip(a, b)):
            assert np.array_equal(x, y), (name, c, np.abs(x - y).max())
    for x, y in zip(ref[3], got[3]):                        # the loss record goes through float atomics (order-dependent rounding)
        np.testing.assert_allclose(x, y, rtol=3e-6, atol=1e-6)


@pytest.mark.parametrize("D,H", [(64, 8), (96, 8), (100, 8), (130, 8), (80, 16), (81, 16)])
def test_wide_cliques_train_against_the_oracle(D, H):
    """Very wide cliques: up to D = 96 (hidden_dim 16: D = 80) the dim-major kernel (its LDS rows grow with D), beyond it the
    tile-major kernels; three training iterations against the float64 oracle either way."""
    K, B, n = 5, 5.0, 130
    blob, x = make_problem(n, D, K, H, 1, seed=D, spread=1.0)
    tb = nh.TrainBatch([dev(x)], [kpack(blob, D, K, H, 1)], K, H, B, 1, lr=0.01, max_iters=3, average_window=3,
                       loss_delta_tol=0.0, early_stop=True)
    assert tb.run(use_graph=False) == [3]
    bc, lc, _, _, _ = CO.train(x, blob, K, H, B, 1, lr=0.01, max_iters=3, early_stop=False, dtype=np.float64)
    np.testing.assert_allclose(tb.iter_loss[0].cpu().numpy()[:3], lc[:3], atol=2e-3, rtol=3e-4)
    err = np.abs(nh.unpack(tb.kparams[0], D, K, H).cpu().numpy() - bc)
    assert np.quantile(err, 0.98) < 2e-3 and err.max() < 0.031, (D, np.quantile(err, 0.98), err.max())
    tb.close()


@pytest.mark.parametrize("H", [8, 16])
def test_mfma_conditioner_matches_the_scalar_path_conditioner(H):
    """Dim-major training kernel (conditioner as v_mfma_f32_4x4x1 chains fed from the LDS weight panel, panel staged
    through the host-built map) against the tile-major wide kernel (`NFISAM_DIM_MAJOR=0`: VALU conditioner with
    scalar-path weights) -- same fp32 products, different summation order; shapes with one tile, partial tiles, D > 16.
    hidden_dim 16: the dim-major kernel's own gradient path (ga2 and ga1 as separate operand tiles, db2 / db1 from chains
    against a constant 1) against the wide kernel's butterfly reduction."""
    K, B = 9, 5.0
    for n, D in ((2000, 15), (333, 17), (64, 2), (1000, 24), (130, 1), (5000, 35)):
        blob, x = make_problem(n, D, K, H, 1, seed=77 + D)
        res = {}
        for mode in ("0", None):
            with _Env(NFISAM_DIM_MAJOR=mode, NFISAM_TRAIN="wide"):
                tb = nh.TrainBatch([dev(x)], [kpack(blob, D, K, H)], K, H, B, 1, lr=0.01, max_iters=3, early_stop=False)
                tb.step()
                torch.cuda.synchronize()
                res[mode] = (float(tb.iter_loss[0][0]), nh.unpack(tb.kparams[0], D, K, H).cpu().numpy(), tb.m[0].cpu().numpy())
        a, b = res["0"], res[None]
        assert abs(a[0] - b[0]) < 2e-5 * max(1.0, abs(a[0])), (n, D, a[0], b[0])
        scale = max(1e-3, float(np.abs(a[2]).max()))
        # m_1 = 0.1 * gradient.  A particle whose coordinate sits on a knot may fall into neighbouring bins in the two kernels
        # (different summation order in theta); that moves every parameter of ONE dim by ~1e-4 of the gradient scale
        # (scripts/exp/h16_err.py: both kernels are equally far from the fp64 oracle there), so the bulk is held tightly and
        # the tail loosely
        d = np.abs(a[2] - b[2])
        assert np.quantile(d, 0.95) < 2e-5 * scale and d.max() < 5e-4 * scale, (n, D, np.quantile(d, 0.95), d.max(), scale)


@pytest.mark.parametrize("H", [4, 8, 16])
def test_every_kernel_instantiation_against_the_oracle(H):
    """All 45 (num_knots, hidden_dim) pairs the library instantiates (2..16 x {4, 8, 16}; include/nfisam_hip.h): forward,
    NLL gradient, inverse round trip and three training iterations (gradient kernel + Adam, fused where the launch
    qualifies) of a small one-layer problem against the float64 C oracle, plus a two-layer forward / gradient."""
    B = 5.0
    for K in range(2, 17):
        assert nh.supported(K, H), (K, H)
        n, D = 70 + K, 4 + (K % 3)
        for L in (1, 2):
            blob, x = make_problem(n, D, K, H, L, seed=17 * K + H + L)
            kp = kpack(blob, D, K, H, L)
            zc, ldc = CO.forward(x, blob, K, H, B, L, dtype=np.float64)
            z, ld, _ = nh.forward(dev(x), kp, K, H, B, L)
            np.testing.assert_allclose(z.cpu().numpy(), zc, atol=Z_ATOL * L, err_msg=str((K, H, L)))
            np.testing.assert_allclose(ld.cpu().numpy(), ldc, atol=LD_ATOL * L, err_msg=str((K, H, L)))
            lossc, gradc, _, _ = CO.nll_grad(x, blob, K, H, B, L, dtype=np.float64, want_gx=True)
            kg, _, loss = nh.backward(dev(x), kp, K, H, B, L, nll_mode=True, want_gx=True)
            assert abs(loss.item() / n + 0.5 * D * np.log(2 * np.pi) - lossc) < 3e-4 * L, (K, H, L)
            grad_close(nh.unpack(kg, D, K, H, L).cpu().numpy() / n, gradc, rtol=2e-3, atol=2e-5 * L)
        # L = 1: inverse of forward, and the training path (the dim-major kernel: every hidden_dim)
        blob1, x1 = make_problem(n, D, K, H, 1, seed=17 * K + H + 1)
        kp1 = kpack(blob1, D, K, H, 1)
        xb = nh.inverse(nh.forward(dev(x1), kp1, K, H, B, 1)[0], None, kp1, K, H, B, 1)
        inside = np.abs(x1).max(1) < 4.9
        np.testing.assert_allclose(xb.cpu().numpy()[inside], x1[inside], atol=3e-3, err_msg=str((K, H)))
        tb = nh.TrainBatch([dev(x1)], [kpack(blob1, D, K, H, 1)], K, H, B, 1, lr=0.01, max_iters=3, average_window=3,
                           loss_delta_tol=0.0, early_stop=True)
        assert tb.run(use_graph=False) == [3]
        bc, lc, _, _, _ = CO.train(x1, blob1, K, H, B, 1, lr=0.01, max_iters=3, early_stop=False, dtype=np.float64)
        np.testing.assert_allclose(tb.iter_loss[0].cpu().numpy()[:3], lc[:3], atol=5e-4, rtol=2e-4, err_msg=str((K, H)))
        err = np.abs(nh.unpack(tb.kparams[0], D, K, H).cpu().numpy() - bc)
        assert np.quantile(err, 0.98) < 2e-3 and err.max() < 0.031, (K, H, np.quantile(err, 0.98), err.max())
        tb.close()


def _widen(blob, D, K, H, Hc, L=1):
    """A width-H reference-order blob written as the width-Hc model with zero rows / columns (numpy)."""
    Po = 3 * K - 1
    P, Pw = nh.param_count(D, K, H), nh.param_count(D, K, Hc)
    out = np.zeros(L * Pw, dtype=blob.dtype)
    for l in range(L):
        src, dst = blob[l * P:(l + 1) * P], out[l * Pw:(l + 1) * Pw]
        dst[:Po] = src[:Po]
        t, tw = Po, Po
        for i in range(1, D):
            for (r, c, rw, cw) in ((H, i, Hc, i), (H, 1, Hc, 1), (H, H, Hc, Hc), (H, 1, Hc, 1), (Po, H, Po, Hc), (Po, 1, Po, 1)):
                blk = np.zeros((rw, cw), dtype=blob.dtype)
                blk[:r, :c] = src[t:t + r * c].reshape(r, c)
                dst[tw:tw + rw * cw] = blk.reshape(-1)
                t += r * c; tw += rw * cw
    return out


@pytest.mark.parametrize("H", [1, 2, 3, 5, 6, 7, 9, 10, 11, 12, 13, 14, 15])
def test_hidden_widths_between_the_compiled_ones_run_zero_padded(H):
    """`hidden_dim` other than 4 / 8 / 16 (the reference takes any: src/flows/flows.py:26-41; its own grid lists 6, 10, 12:
    example/slam/manhattan_world_with_range/lawnmower_4x4/run_nfisam.py:5-6) runs as the next compiled width with zero hidden
    units (ABI 1500).  Against the float64 C oracle AT WIDTH H: forward, NLL gradient, inverse, a 60-iteration training plan
    (chunk-persistent form, early-stop bookkeeping) for L = 1 and forward / gradient / 5 iterations for L = 2; and the padding
    of parameters and both Adam moments is EXACTLY zero after training -- a padded unit's activation is tanh(0) = 0 and its
    out-weights are 0, so every gradient that touches it is a product with an exact zero."""
    B = 5.0
    Hc = 4 if H <= 4 else (8 if H <= 8 else 16)
    assert nh.supported(9, H)
    for K, n, D in ((9, 300, 6), (12, 130, 3)):
        pad = torch.from_numpy(nh.layout_map(D, K, H) < 0).to(DEV)
        for L in (1, 2):
            blob, x = make_problem(n, D, K, H, L, seed=31 * H + K + L)
            kp = kpack(blob, D, K, H, L)
            assert kp.numel() == L * nh.kparam_count(D, K, Hc)
            assert torch.equal(kp, kpack(_widen(blob, D, K, H, Hc, L), D, K, Hc, L))     # the same kernel blob as the explicit wide model
            zc, ldc = CO.forward(x, blob, K, H, B, L, dtype=np.float64)
            z, ld, _ = nh.forward(dev(x), kp, K, H, B, L)
            np.testing.assert_allclose(z.cpu().numpy(), zc, atol=Z_ATOL * L, err_msg=str((K, H, L)))
            np.testing.assert_allclose(ld.cpu().numpy(), ldc, atol=LD_ATOL * L, err_msg=str((K, H, L)))
            lossc, gradc, _, _ = CO.nll_grad(x, blob, K, H, B, L, dtype=np.float64, want_gx=True)
            kg, _, loss = nh.backward(dev(x), kp, K, H, B, L, nll_mode=True, want_gx=True)
            assert abs(loss.item() / n + 0.5 * D * np.log(2 * np.pi) - lossc) < 3e-4 * L, (K, H, L)
            grad_close(nh.unpack(kg, D, K, H, L).cpu().numpy() / n, gradc, rtol=2e-3, atol=2e-5 * L)
            assert float(kg.reshape(L, -1)[:, pad].abs().sum()) == 0.0
            iters = 60 if L == 1 else 5
            tb = nh.TrainBatch([dev(x)], [kp.clone()], K, H, B, L, lr=0.01, max_iters=iters, average_window=20, loss_delta_tol=0.0,
                               early_stop=True)
            assert tb.run(use_graph=True) == [iters]
            bc, lc, _, _, _ = CO.train(x, blob, K, H, B, L, lr=0.01, max_iters=iters, early_stop=False, dtype=np.float64)
            # (float32 against the float64 oracle: tight while the two trajectories are the same trajectory, then within the
            #  rounding-amplified drift of an Adam run on a few hundred particles -- measured up to 0.02 at iteration ~50)
            il = tb.iter_loss[0].cpu().numpy()
            np.testing.assert_allclose(il[:12], lc[:12], atol=1e-3, rtol=2e-4, err_msg=str((K, H, L)))
            # (beyond: two Adam trajectories of a few hundred particles drift apart at rounding-amplified pace -- up to 0.19 of a
            #  loss of 6 seen at iteration ~50 -- and stay the same optimisation: within 5 %, both decreasing)
            np.testing.assert_allclose(il[:iters], lc[:iters], rtol=5e-2, err_msg=str((K, H, L)))
            assert il[iters - 1] < il[0] and np.all(np.isfinite(il[:iters]))
            err = np.abs(nh.unpack(tb.kparams[0], D, K, H, L).cpu().numpy() - bc)
            assert np.median(err) < (5e-2 if L == 1 else 3e-3), (K, H, L, np.median(err), err.max())
            for t in (tb.kparams[0], tb.m[0], tb.v[0]):
                assert float(t.reshape(L, -1)[:, pad].abs().sum()) == 0.0, (K, H, L)
            tb.close()
        # inverse of forward at width H
        blob1, x1 = make_problem(n, D, K, H, 1, seed=77 + H)
        kp1 = kpack(blob1, D, K, H, 1)
        xb = nh.inverse(nh.forward(dev(x1), kp1, K, H, B, 1)[0], None, kp1, K, H, B, 1)
        inside = np.abs(x1).max(1) < 4.9
        np.testing.assert_allclose(xb.cpu().numpy()[inside], x1[inside], atol=3e-3, err_msg=str((K, H)))
        xo, _ = CO.inverse(x1[:, 2:].copy(), x1[:, :2].copy(), blob1, K, H, B, 1, dtype=np.float64)
        xg = nh.inverse(dev(x1[:, 2:].copy()), dev(x1[:, :2].copy()), kp1, K, H, B, 1).cpu().numpy()
        ok = np.abs(xo).max(1) < 4.5
        np.testing.assert_allclose(xg[ok], xo[ok], atol=3e-3)


@pytest.mark.parametrize("H", [8, 16])
def test_throughput_launch_with_wide_dims_matches_the_tile_major_kernels(H):
    """24 cliques of D = 18..20, n = 2000: the dim-major kernel sweeps several tiles per wave (T > 1) and accumulates the
    dW0 rows 16.. in its own LDS rows across them; first Adam moments (0.1 x gradient) and losses against the tile-major
    wide kernel (`NFISAM_DIM_MAJOR=0`), which shares no code with that path."""
    K, B = 9, 5.0
    shapes = [(2000, 18 + (c % 3)) for c in range(24)]
    res = {}
    for mode in ("0", None):
        with _Env(NFISAM_DIM_MAJOR=mode):
            probs = [make_problem(n, D, K, H, 1, seed=900 + c, spread=1.0) for c, (n, D) in enumerate(shapes)]
            tb = nh.TrainBatch([dev(x) for _, x in probs], [kpack(b, D, K, H) for (b, _), (_, D) in zip(probs, shapes)],
                               K, H, B, 1, lr=0.01, max_iters=1, average_window=1, loss_delta_tol=0.0, early_stop=True)
            assert tb.run(use_graph=False) == [1] * len(shapes)
            torch.cuda.synchronize()
            res[mode] = ([m.cpu().numpy() for m in tb.m], [float(l[0]) for l in tb.iter_loss])
            tb.close()
    for c, (a, b) in enumerate(zip(res["0"][0], res[None][0])):
        # (the dim-major kernel stages W0 / W1 pre-multiplied by 2 log2(e) for its tanh: products rounded once more)
        d, sc = np.abs(a - b), max(1e-3, np.abs(a).max())      # (bulk / tail: see test_mfma_conditioner_matches_the_scalar_path_conditioner)
        assert np.quantile(d, 0.95) < 5e-5 * sc and d.max() < 5e-4 * sc, (c, np.quantile(d, 0.95), d.max(), sc)
    np.testing.assert_allclose(res["0"][1], res[None][1], rtol=2e-6, atol=1e-5)


def _train_layers(shapes, L, iters, window, use_graph, K=9, H=8, B=5.0, lr=0.01, early_stop=True):
    probs = [make_problem(n, D, K, H, L, seed=900 + c, spread=1.0) for c, (n, D) in enumerate(shapes)]
    tb = nh.TrainBatch([dev(x) for _, x in probs], [kpack(b, D, K, H, L) for (b, _), (_, D) in zip(probs, shapes)], K, H, B, L,
                       lr=lr, max_iters=iters, average_window=window, loss_delta_tol=0.0, early_stop=early_stop)
    done = tb.run(use_graph=use_graph)
    torch.cuda.synchronize()
    out = [[t.cpu().numpy().copy() for t in arr] for arr in (tb.kparams, tb.m, tb.v, tb.iter_loss)]
    tb.close()
    return done, out, probs


@pytest.mark.parametrize("shapes,L,iters,window", [([(4096, 6)], 4, 23, 10), ([(300, 5), (1000, 7), (200, 2), (64, 1)], 2, 61, 50),
                                                   ([(513, 7)], 3, 12, 5)], ids=["c2", "ragged-batch", "odd-D"])
@pytest.mark.parametrize("use_graph", [True, False], ids=["hipgraph", "eager"])
def test_multilayer_panel_image_equals_staging_from_the_parameters(shapes, L, iters, window, use_graph):
    """Multi-layer training launches (nsf_train3_kernel, two dims per wave): from the second iteration of a chunk on the
    blocks COPY their conditioner panels from the clique's panel image, which the Adam kernel keeps next to the loss ring
    (nsf_cond_mfma.h: build_pair_map); `NFISAM_PAIR_IMAGE=0` makes every launch stage from the parameters instead.  Same
    values by construction -- parameters, moments and the loss record must agree bit for bit (single-writer gradient
    copies and dL/dx rows: nothing in the kernel depends on an order of arrival); full, partial and single chunks, cliques
    of different widths in one launch, D = 1 and odd D."""
    with _Env(NFISAM_PAIR_IMAGE="0"):
        d0, ref, _ = _train_layers(shapes, L, iters, window, use_graph)
    with _Env(NFISAM_PAIR_IMAGE=None):
        d1, got, _ = _train_layers(shapes, L, iters, window, use_graph)
        d2, again, _ = _train_layers(shapes, L, iters, window, use_graph)
    assert d0 == d1 == d2 == [iters] * len(shapes)
    for name, a, b, c2 in zip(("theta", "m", "v"), ref[:3], got[:3], again[:3]):
        for c, (x, y, z) in enumerate(zip(a, b, c2)):
            assert np.array_equal(x, y) and np.array_equal(y, z), (name, c, np.abs(x - y).max())
    for x, y in zip(ref[3], got[3]):                        # the loss record goes through float atomics (order-dependent rounding)
        np.testing.assert_allclose(x, y, rtol=3e-6, atol=1e-6)


@pytest.mark.parametrize("K", [9, 16, 2])
def test_multilayer_hidden_dim_4_on_the_two_dims_per_wave_kernel(K):
    _pair_kernel_of_another_width(K, 4, [(300, 5), (257, 8), (64, 1), (200, 2), (129, 11)])


@pytest.mark.parametrize("K", [9, 16, 2])
def test_multilayer_hidden_dim_16_on_the_two_dims_per_wave_kernel(K):
    """hidden_dim 16 with stacked layers (round 5; VERDICT r4 missing #4): nsf_train3_kernel instantiated for H = 16 -- the
    gradient GEMMs of nsf_train1_kernel's WIDE_H form (ga2 and ga1 one 16-row operand tile each, biases from chains against 1)
    and the panels of ONE layer resident (all layers of C2's shape would be 218 KB): a wave brings its two dims' panels of the
    stage's layer at the top of every stage, from the clique's panel image or -- iteration 0 of a chunk, VJP calls -- from
    the parameters.  Same checks as hidden_dim 4 (below): against the generic kernel, the float64 oracle, image on / off.
    (C2's shape with hidden_dim 16: `regimes.C2_..._H16` of the bench line.)"""
    # (round 6: the widest clique the kernel takes at this hidden width goes first -- one layer's panels of ALL the clique's dims
    #  must fit the CU's LDS next to the tiles, and the launch-shape rule asks that of num_knots 16 whatever K is (`pair_h4_fits`,
    #  csrc/nsf_kernels.hip: a function of the launch shape only): D <= 9.  Wider cliques, PAIR_MAX_D = 16 included, train on the
    #  generic kernel.  So the last wave of an odd clique, five waves and a dW0 row beyond 8 meet the oracle too, not only D <= 8.)
    widest = 9
    _pair_kernel_of_another_width(K, 16, [(120, widest), (300, 5), (257, 8), (64, 1), (200, 2), (129, 6)])


def _pair_kernel_of_another_width(K, H, shapes):
    """hidden_dim 4 with stacked layers: small launches take nsf_train3_kernel (MFMA conditioner, two dims per wave, panel
    image) like hidden_dim 8 -- there is no two-lanes-per-particle kernel of that width, so every clique width 1..16 goes
    there.  Against the tile-major generic kernel (`NFISAM_TRAIN=wide`, shares no code with it): the first Adam moments
    (0.1 x gradient, 0.001 x gradient^2) and the first loss of a ragged batch; against the float64 C oracle: gradient and
    dL/dx of single calls (`nfisam_nsf_backward`), and a six-iteration loss curve; panel image on / off bit for bit.
    (C2's shape with hidden_dim 4: 60.3 -> 29.2 us per iteration, `regimes.C2_..._H04` of the bench line.)"""
    B, L = 5.0, 3                                       # (a small launch; hidden_dim 4: D = 11 x 3 layers -- the panels of num_knots 16 still fit)
    # Bars (round 6, VERDICT r5 weak #1d): SURVEY 8(c)'s -- 1e-4 x scale between kernels, 2e-5 x L absolute + 2e-3 relative against
    # the float64 oracle, 4e-2 on the loss curve -- for EVERY (K, H), plus what float32 itself costs on the problem at hand: three
    # times the distance of the float32 ORACLE from the float64 one (its 99.9 % quantile), as test_multilayer_training_follows_the_oracle
    # does.  Round 5 had loosened the constants 10-30 x for all of hidden_dim 16 because of (K, H) = (16, 16); what is wild is not a
    # (K, H) but single PROBLEMS: three stacked layers can put a particle within rounding of a knot of a later layer, and then every
    # float32 evaluation differs -- scripts/exp/h16_diag2.py, K = 9, H = 16, n = 300, D = 5, seed 901: float32 oracle 7.1e-3 x max|g|
    # off float64 (589 of 9318 entries beyond 1e-4), generic kernel 2.9e-3, this kernel 4.0e-4; seeds 900 / 902: all three 1e-5.
    with _Env(NFISAM_TRAIN="wide"):
        _, wide, _ = _train_layers(shapes, L, 1, 1, False, K=K, H=H)
    _, pair, probs = _train_layers(shapes, L, 1, 1, False, K=K, H=H)
    g64s, owns = [], []
    for c, (n, D) in enumerate(shapes):
        blob, x = probs[c]
        lossc, gradc, _, gxc = CO.nll_grad(x, blob, K, H, B, L, dtype=np.float64, want_gx=True)
        _, g32, _, _ = CO.nll_grad(x, blob, K, H, B, L, dtype=np.float32, want_gx=True)
        g64s.append((lossc, gradc, gxc))
        owns.append(float(np.quantile(np.abs(np.asarray(g32, dtype=np.float64) - gradc), 0.999)))      # float32's own distance from float64
    for c in range(len(shapes)):
        assert not np.array_equal(wide[1][c], pair[1][c])       # two kernels (rounding differs somewhere)
        scale = np.abs(wide[1][c]).max()
        own_m = 0.1 * owns[c]                                   # (the first Adam moment is 0.1 x gradient)
        d = np.abs(pair[1][c] - wide[1][c])
        excess = d - 2e-3 * np.abs(wide[1][c]) - 1e-4 * scale - 6.0 * own_m
        print("K %d H %d clique %d (n %d, D %d): kernel vs kernel %.2e x max|m| (float32 oracle's own distance from float64: %.2e x), over the bar %d of %d"
              % (K, H, c, shapes[c][0], shapes[c][1], d.max() / scale, own_m / scale, int((excess > 0).sum()), excess.size))
        assert (excess > 0).mean() < 3e-4 and excess.max() < 2e-3 * scale, (K, c, int((excess > 0).sum()), float(excess.max() / scale))
        np.testing.assert_allclose(pair[3][c][:1], wide[3][c][:1], rtol=2e-5)
    for c, (n, D) in enumerate(shapes[:3]):
        blob, x = probs[c]
        lossc, gradc, gxc = g64s[c]
        kg, gx, loss = nh.backward(dev(x), kpack(blob, D, K, H, L), K, H, B, L, nll_mode=True, want_gx=True)
        assert abs(loss.item() / n + 0.5 * D * np.log(2 * np.pi) - lossc) < 3e-4 * L, (K, c)
        # (three layers in fp32 against fp64: a particle next to a knot may pick the other bin -- a handful of the ~17 k entries
        #  may sit up to 1 % off; measured at hidden_dim 16: 1 entry, 0.75 %)
        gk, sc = nh.unpack(kg, D, K, H, L).cpu().numpy() / n, max(1.0, float(np.abs(gradc).max()))
        excess = np.abs(gk - gradc) - 2e-3 * np.abs(gradc) - 2e-5 * L * sc - 3.0 * owns[c]
        print("K %d H %d clique %d (D %d): gradient vs float64 oracle: max |err| %.2e x max|g| (float32 oracle's own q99.9: %.2e x), over the bar %d of %d"
              % (K, H, c, D, np.abs(gk - gradc).max() / sc, owns[c] / sc, int((excess > 0).sum()), excess.size))
        assert (excess > 0).mean() < 3e-4 and excess.max() < 2e-3 * sc, (K, c, (excess > 0).sum(), excess.max())
        ex = np.abs(gx.cpu().numpy() / n - gxc)            # (a particle next to a knot may pick the other bin in fp32: dL/dx jumps there)
        assert np.quantile(ex, 0.98) < 2e-3 * np.abs(gxc).max() + 2e-5 * L + 3.0 * owns[c] and ex.max() < 0.05 * max(1.0, np.abs(gxc).max()), (K, c)
    iters = 6
    with _Env(NFISAM_PAIR_IMAGE="0"):
        _, ref, _ = _train_layers(shapes, L, iters, 50, True, K=K, H=H, lr=0.02, early_stop=False)
    done, out, probs = _train_layers(shapes, L, iters, 50, True, K=K, H=H, lr=0.02, early_stop=False)
    assert done == [iters] * len(shapes)
    for c, (n, D) in enumerate(shapes):
        for a, b in zip(ref[:3], out[:3]):
            assert np.array_equal(a[c], b[c]), (K, c)
        if c < 2:
            blob, x = probs[c]
            _, l64, _, _, _ = CO.train(x, blob, K, H, B, L, lr=0.02, max_iters=iters, early_stop=False, dtype=np.float64)
            _, l32, _, _, _ = CO.train(x, blob, K, H, B, L, lr=0.02, max_iters=iters, early_stop=False, dtype=np.float32)
            got, l64 = out[3][c][:iters].astype(np.float64), np.asarray(l64, dtype=np.float64)[:iters]
            own = np.abs(np.asarray(l32, dtype=np.float64)[:iters] - l64)
            print("K %d H %d clique %d: loss curve vs float64 oracle, relative: %s   (float32 oracle's own: %s)"
                  % (K, H, c, " ".join("%.1e" % v for v in np.abs(got / l64 - 1.0)), " ".join("%.1e" % v for v in own / np.abs(l64))))
            # the first three iterations to 2e-4 absolute, the curve to 4e-2 relative (three layers at this step size amplify rounding
            # from the fourth iteration on: test_multilayer_training_follows_the_oracle), each + 3 x the float32 oracle's own distance
            # ((K, H) = (16, 16): the loss RISES at the second iteration -- 25.3, 28.1, 23.4 -- in the oracle too, an unstable start
            #  from which float32 and float64 part ways: 8 % at iteration 6 for the oracle itself)
            assert np.all(np.abs(got[:3] - l64[:3]) <= 2e-4 + 1e-5 * np.abs(l64[:3]) + 3.0 * own[:3]), (K, c, got, l64, own)
            assert np.all(np.abs(got - l64) <= 4e-2 * np.abs(l64) + 3.0 * own), (K, c, got, l64, own)


def test_multilayer_training_follows_the_oracle():
    """Eight Adam iterations of multi-layer flows through the training plan (gradient kernel + Adam kernel + panel image)
    against the oracle's loop.  Three layers at this step size amplify rounding (the oracle in fp32 and in fp64 are 3e-2
    apart in loss after 8 iterations, scripts/exp/ml_traj.py), so the yardstick is the fp64 trajectory and the bound is what
    the fp32 oracle itself achieves against it: the first iterations must agree to 1e-4, every iteration to
    3 x the fp32 oracle's own error + 1e-3."""
    K, H, B, L, iters = 9, 8, 5.0, 3, 8
    shapes = [(257, 4), (600, 6), (90, 7)]
    done, out, probs = _train_layers(shapes, L, iters, 50, True, lr=0.02, early_stop=False)
    assert done == [iters] * len(shapes)
    for c, (n, D) in enumerate(shapes):
        blob, x = probs[c]
        b32, l32, _, _, _ = CO.train(x, blob, K, H, B, L, lr=0.02, max_iters=iters, early_stop=False, dtype=np.float32)
        b64, l64, _, _, _ = CO.train(x, blob, K, H, B, L, lr=0.02, max_iters=iters, early_stop=False, dtype=np.float64)
        got_l = out[3][c][:iters].astype(np.float64)
        np.testing.assert_allclose(got_l[:3], l64[:3], atol=1e-4, rtol=1e-5)
        assert np.all(np.abs(got_l - l64) <= 3.0 * np.abs(l32 - l64) + 1e-3), (c, np.abs(got_l - l64), np.abs(l32 - l64))
        got = nh.unpack(torch.from_numpy(out[0][c]).to(DEV), D, K, H, L).cpu().numpy()
        err, ref = np.abs(got - b64), np.abs(b32 - b64)
        assert np.quantile(err, 0.99) <= 3.0 * np.quantile(ref, 0.99) + 2e-3, (c, np.quantile(err, 0.99), np.quantile(ref, 0.99))
        assert err.max() < iters * 0.02 + 1e-3


@pytest.mark.parametrize("H", [4, 16])
def test_big_batches_of_other_hidden_widths_train_on_both_kernel_families(H):
    """64 cliques of n = 2000: the launch-shape helpers (tiles per gradient copy, copies per workspace) must agree between
    the gradient, Adam and bookkeeping launches for hidden widths 4 and 16 on the dim-major kernel AND on the tile-major
    kernel they fall back to (`NFISAM_DIM_MAJOR=0`; H = 16 with D > 80): first loss and a 20-iteration descent of every
    clique against a single-clique run of the same problem."""
    K, B, n, D = 9, 5.0, 2000, 7
    probs = [make_problem(n, D, K, H, 1, seed=1200 + c, spread=1.0) for c in range(64)]
    single = nh.TrainBatch([dev(probs[5][1])], [kpack(probs[5][0], D, K, H)], K, H, B, 1, lr=0.01, max_iters=20, early_stop=False)
    assert single.run(use_graph=True) == [20]
    ref = single.iter_loss[0].cpu().numpy()[:20]
    single.close()
    for mode in ("0", None):
        with _Env(NFISAM_DIM_MAJOR=mode):
            tb = nh.TrainBatch([dev(x) for _, x in probs], [kpack(b, D, K, H) for b, _ in probs], K, H, B, 1, lr=0.01, max_iters=20,
                               early_stop=False)
            assert tb.run(use_graph=True) == [20] * 64
            il = tb.iter_loss[5].cpu().numpy()[:20]
            tb.close()
        # (two kernel families = two summation orders: the trajectories drift apart by rounding, ~1e-2 after 20 iterations)
        np.testing.assert_allclose(il[:5], ref[:5], rtol=1e-4, atol=5e-4, err_msg=str((H, mode)))
        np.testing.assert_allclose(il, ref, atol=3e-2, err_msg=str((H, mode)))
        assert il[19] < il[0] - 1.0


def test_multilayer_launch_beyond_the_slab_limit_uses_atomics_and_still_trains():
    """n = 5000, D = 6, L = 2: 157 tiles of 32 -- more than the 128 gradient copies a workspace holds, so the multi-layer kernel
    accumulates into ONE copy with float atomics (order-dependent rounding) while the panel image and the parked forward
    state work as in the slab case; ten iterations against the split kernel (`NFISAM_PAIR=0`) and the float64 oracle."""
    K, H, B, L, n, D, iters = 9, 8, 5.0, 2, 5000, 6, 10
    blob, x = make_problem(n, D, K, H, L, seed=4242, spread=1.0)
    res = {}
    for pair in ("0", None):
        with _Env(NFISAM_PAIR=pair):
            tb = nh.TrainBatch([dev(x)], [kpack(blob, D, K, H, L)], K, H, B, L, lr=0.01, max_iters=iters, early_stop=False)
            assert tb.run(use_graph=True) == [iters]
            res[pair] = (tb.iter_loss[0].cpu().numpy()[:iters].astype(np.float64), nh.unpack(tb.kparams[0], D, K, H, L).cpu().numpy())
            tb.close()
    b64, l64, _, _, _ = CO.train(x, blob, K, H, B, L, lr=0.01, max_iters=iters, early_stop=False, dtype=np.float64)
    for pair in ("0", None):
        np.testing.assert_allclose(res[pair][0][:4], l64[:4], atol=2e-4, rtol=1e-5, err_msg=str(pair))
        np.testing.assert_allclose(res[pair][0], l64, atol=2e-2, err_msg=str(pair))       # (fp32 vs fp64 trajectories drift: 6e-3 after 9 iterations)
        assert np.quantile(np.abs(res[pair][1] - b64), 0.98) < 3e-3, pair


@pytest.mark.timeout(120)
def test_plan_as_a_conveyor_of_chunks_trains_every_slot_like_a_plan_of_its_own():
    """`TrainBatch.begin / feed / peek / refill / end` (nfisam_nsf_train_plan_begin ..: the slot scheduler of
    slam.ReplicaNFiSAM): four slots, each training a sequence of problems that enter between two chunks as the previous one
    stops; the library's feeder thread launches the chunks.  Every problem must come out with the iterations, parameters and
    loss record of a plan of its own, bit for bit (a refill that interleaved with a chunk's graph launch once activated a
    slot mid-chunk), over several begin .. end cycles of one plan; empty slots and run-ahead chunks change nothing."""
    K, H, B, L, R, n, D, per_slot, cycles = 9, 8, 5.0, 1, 4, 1000, 7, 3, 3
    # (tolerance 0.015: the 36 problems stop at 150 .. 350 iterations in either kernel family -- with 0.02 all of them stop at 150
    #  on the two-lanes-per-particle family of round 6, scripts/exp/conveyor_stops.py)
    kw = dict(lr=0.02, max_iters=400, average_window=50, loss_delta_tol=0.015, early_stop=True)

    def problem(i):
        blob, x = make_problem(n, D, K, H, L, seed=3000 + i, spread=0.6 + 0.3 * (i % 4))
        return dev(x), kpack(blob, D, K, H, L)
    ref = {}
    for i in range(R * per_slot * cycles):
        x, kp = problem(i)
        tb1 = nh.TrainBatch([x], [kp.clone()], K, H, B, L, **kw)
        it = tb1.run(use_graph=True)
        ref[i] = (it[0], tb1.kparams[0].clone(), tb1.iter_loss[0].clone())
        tb1.close()
    assert len({v[0] for v in ref.values()}) > 1                 # the problems stop at different iterations
    tb = nh.TrainBatch([torch.zeros(n, D, device=DEV) for _ in range(R)], [torch.zeros_like(problem(0)[1]) for _ in range(R)],
                       K, H, B, L, **kw)
    tb.states[:, 1] = 1                                          # empty slots look finished
    nxt = 0
    for cyc in range(cycles):
        tb.begin()
        tb.feed(2)
        owner, seq0, keep, left = [None] * R, [0] * R, [None] * R, [per_slot] * R

        def load(r):
            nonlocal nxt
            keep[r] = problem(nxt)                               # (the sources stay alive while the slot trains them)
            tb.refill(r, *keep[r])
            owner[r], seq0[r] = nxt, tb.enqueued()
            nxt += 1
            left[r] -= 1
        for r in range(R):
            load(r)
        t_last = time.time()
        while any(o is not None for o in owner):
            seq, st = tb.peek()
            for r in range(R):
                if seq >= 0 and owner[r] is not None and seq > seq0[r] and (st[r][1] != 0 or st[r][0] >= kw["max_iters"]):
                    i, owner[r] = owner[r], None
                    assert st[r][0] == ref[i][0], (cyc, r, i, st[r][0], ref[i][0])
                    assert torch.equal(tb.kparams[r], ref[i][1]), (cyc, r, i)
                    np.testing.assert_allclose(tb.iter_loss[r].cpu().numpy(), ref[i][2].cpu().numpy(), rtol=3e-6, atol=1e-6)
                    if left[r] > 0:
                        load(r)
                    t_last = time.time()
            assert time.time() - t_last < 20, "the conveyor stalled"
            time.sleep(2e-5)
        tb.end()
        torch.cuda.synchronize()
        assert tb.enqueued() < 200                               # no chunks launched for an idle conveyor
    tb.close()


PERSIST_WORKER = r'''
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(%(root)r, "nf-isam_amd")); sys.path.insert(0, %(root)r)
import nfisam_hip as nh
dev = torch.device("cuda", 0)
K, B = 9, 5.0
out = {}
for name, shapes, iters, window, tol, H in (("plaza", [(2000, 15)], 230, 50, 0.0, 8), ("two_ragged", [(1500, 11), (777, 16)], 200, 50, 0.0, 8),
                                            ("early_stop", [(2000, 7)], 600, 50, 0.05, 8), ("tiny", [(64, 3)], 120, 40, 0.0, 8),
                                            ("plaza_h16", [(2000, 15)], 130, 50, 0.0, 16), ("h4", [(900, 9)], 100, 50, 0.0, 4),
                                            ("three_wide", [(2000, 16)] * 3, 100, 50, 0.0, 8), ("d24", [(1000, 24), (600, 19)], 100, 50, 0.0, 8),
                                            ("d33", [(700, 33)], 100, 50, 0.0, 8),
                                            ("n4096", [(4096, 6)], 130, 50, 0.0, 8), ("n3000_n2500", [(3000, 9), (2500, 5)], 100, 50, 0.0, 8),
                                            ("n4096_h16", [(4096, 7)], 100, 50, 0.0, 16), ("mixed_16_and_4_blocks", [(4096, 6), (1000, 5)], 100, 50, 0.0, 8),
                                            ("c3", [(2000, D) for D in (6, 8, 8, 10, 10, 12, 12, 12)], 100, 50, 0.0, 8)):
    gen = torch.Generator().manual_seed(len(name))
    xs = [(1.3 * torch.randn(n, D, generator=gen)).clamp_(-4, 4).to(dev) for n, D in shapes]
    kp = [nh.pack((0.2 * torch.randn(nh.param_count(D, K, H), generator=gen)).to(dev), D, K, H, 1) for n, D in shapes]
    tb = nh.TrainBatch(xs, kp, K, H, B, 1, lr=0.01, max_iters=iters, average_window=window, loss_delta_tol=tol, early_stop=True)
    done = tb.run()
    for c in range(len(shapes)):
        out["%%s_%%d_params" %% (name, c)] = tb.kparams[c].cpu().numpy()
        out["%%s_%%d_loss" %% (name, c)] = tb.iter_loss[c].cpu().numpy()
        out["%%s_%%d_iters" %% (name, c)] = np.array(done[c])
    out["span_%%s" %% name] = np.array(tb.xcd_span())
    tb.close()
np.savez(sys.argv[1], **out)
'''


@pytest.mark.timeout(700)
def test_chunk_persistent_kernel_is_bit_identical_to_one_launch_per_iteration(tmp_path):
    """The chunk-persistent form of the dim-major kernel (a chunk's iterations in ONE launch per chain, the blocks of a
    (clique, dim) group meeting at a barrier per iteration; default for launches that are resident at once) against
    NFISAM_PERSIST=0 (one launch per iteration): the same parameters, loss records and early-stop iterations, bit for bit --
    a Plaza-shaped clique (230 iterations: full chunks of 50 through the persistent graph, the last 30 eagerly), two ragged
    cliques in one plan, a run that stops early, a clique of one tile, hidden_dim 16 and 4, three cliques of D = 16 (384 blocks in two
    parallel persistent launches), D = 19 / 24 / 33, the eight C3 cliques (624 blocks: three per CU); round 5: cliques of MORE THAN
    2048 PARTICLES -- n = 4096, D = 6 (BASELINE config[1]'s batch: sixteen blocks per (clique, dim) group), a ragged pair of 12 and 10
    blocks, hidden_dim 16, a plan that mixes groups of sixteen and of four blocks -- whose groups exchange up to sixteen tagged copies in two passes, summed in the lane-partial order
    nsf_adam_kernel uses for that many copies (their one-launch-per-iteration form is gradient kernel + Adam kernel).
    Third run, NFISAM_PERSIST_SCATTER=1 (round 4): the launch's grid is transposed so that the blocks of a group are
    neighbours in dispatch order, i.e. on DIFFERENT XCDs (checked: every plan reports a span > 1 where a group has several
    blocks; the normal launch reports exactly 1) -- still the same bits: the group's exchange goes through agent-scope
    write-through stores and agent-scope loads, the one-XCD placement is for speed only.  (The knobs are read once per process.)"""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "worker.py"
    script.write_text(PERSIST_WORKER % dict(root=root))
    res = {}
    for knob, env in (("1", dict(NFISAM_PERSIST="1")), ("0", dict(NFISAM_PERSIST="0")),
                      ("scatter", dict(NFISAM_PERSIST="1", NFISAM_PERSIST_SCATTER="1")),
                      ("split", dict(NFISAM_PERSIST="1", NFISAM_PERSIST_SPLIT="1")),
                      ("whole", dict(NFISAM_PERSIST="1", NFISAM_PERSIST_SPLIT="0", NFISAM_PERSIST_SCATTER="1"))):
        out = str(tmp_path / ("persist_%s.npz" % knob))
        p = subprocess.run([sys.executable, str(script), out], env=dict(os.environ, **env), capture_output=True,
                           text=True, timeout=300)
        assert p.returncode == 0, p.stderr[-2000:]
        res[knob] = dict(np.load(out))
    assert res["1"].keys() == res["0"].keys() == res["scatter"].keys() == res["split"].keys() == res["whole"].keys()
    for k in res["1"]:
        if k.startswith("span_"):
            continue
        np.testing.assert_array_equal(res["1"][k], res["0"][k], err_msg=k)
        np.testing.assert_array_equal(res["scatter"][k], res["0"][k], err_msg="scattered: " + k)
        # the two ways of staging the update -- every block derives the whole dim's (default for launches of at most one block
        # per CU) / every block derives its slice and publishes it (default beyond) -- forced onto every shape:
        np.testing.assert_array_equal(res["split"][k], res["0"][k], err_msg="update divided among the blocks: " + k)
        np.testing.assert_array_equal(res["whole"][k], res["0"][k], err_msg="whole update per block, scattered: " + k)
    spans = {k[5:]: (int(res["1"][k]), int(res["0"][k]), int(res["scatter"][k])) for k in res["1"] if k.startswith("span_")}
    print("XCDs per (clique, dim) group: persistent / one launch per iteration / scattered", spans)
    for name, (normal, plain, scattered) in spans.items():
        assert plain == 0, (name, plain)                            # no persistent chunk ran
        assert normal == 1, (name, normal)                          # the grid puts a group behind ONE L2
        assert scattered > 1 or name == "tiny", (name, scattered)   # ("tiny": one block per group)
    assert int(res["1"]["early_stop_0_iters"]) < 600 and int(res["1"]["plaza_0_iters"]) == 230
    assert np.all(np.isfinite(res["1"]["plaza_0_loss"][:230])) and res["1"]["plaza_0_loss"][229] < res["1"]["plaza_0_loss"][0]


@pytest.mark.parametrize("K,H", [(10, 8), (12, 8), (15, 8), (16, 8), (14, 4)])
def test_lean_builds_of_wide_splines_agree_with_the_three_wave_builds(K, H):
    """num_knots >= 12 (the chunk-persistent form: >= 10) spill at the three waves per SIMD the dim-major kernels are compiled
    for; launches that are resident at two blocks per CU anyway (single cliques: every fit of a real run) take a second
    instantiation compiled for two waves per SIMD, without scratch (nsf_unit.hip: LEAN; the reference's examples use K = 12 and
    15).  Same source, another register allocation and instruction selection: the first gradient agrees to rounding (the
    parameters after one Adam step to ~1e-6, i.e. they are NOT bit-identical: K = 12 differs by 3e-7), so against
    `NFISAM_LEAN=0`: parameters after 1 and 5 iterations (a coordinate whose gradient is at rounding level may step the other
    way: 2 x lr), the loss record of the first ten iterations to 1e-6, and the loss after 130 iterations through the
    chunk-persistent graph + the eager tail to 3 %; a 20-clique batch (too many blocks for the lean build: the same kernel
    both ways) must not change at all."""
    B, lr = 5.0, 0.01
    def run(shapes, iters, window, use_graph, seed=3):
        gen = torch.Generator().manual_seed(seed)
        xs = [(1.3 * torch.randn(n, D, generator=gen)).clamp_(-4, 4).to(DEV) for n, D in shapes]
        kp = [nh.pack((0.2 * torch.randn(nh.param_count(D, K, H), generator=gen)).to(DEV), D, K, H, 1) for n, D in shapes]
        tb = nh.TrainBatch(xs, kp, K, H, B, 1, lr=lr, max_iters=iters, average_window=window, loss_delta_tol=0.0, early_stop=True)
        done = tb.run(use_graph=use_graph)
        torch.cuda.synchronize()
        out = (done, [t.cpu().numpy().copy() for t in tb.kparams], [t.cpu().numpy().copy()[:iters] for t in tb.iter_loss])
        tb.close()
        return out
    for iters, window, graph in ((1, 1, False), (5, 5, False), (130, 50, True)):
        with _Env(NFISAM_LEAN="0"):
            ref = run([(2000, 15)], iters, window, graph)
        got = run([(2000, 15)], iters, window, graph)
        assert ref[0] == got[0] == [iters]
        np.testing.assert_allclose(got[2][0][:10], ref[2][0][:10], rtol=1e-6)
        if iters <= 5:
            d = np.abs(got[1][0] - ref[1][0])
            assert np.quantile(d, 0.999) < 1e-5 and d.max() <= 2 * lr * iters + 1e-6, (K, iters, np.quantile(d, 0.999), d.max())
        else:
            assert abs(got[2][0][iters - 1] / ref[2][0][iters - 1] - 1.0) < 0.03, (K, got[2][0][iters - 1], ref[2][0][iters - 1])
    with _Env(NFISAM_LEAN="0"):
        ref = run([(2000, 9)] * 20, 12, 6, True)
    got = run([(2000, 9)] * 20, 12, 6, True)
    for a, b in zip(ref[1] + ref[2], got[1] + got[2]):
        assert np.array_equal(a, b), K


STALL_WORKER = r'''
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(%(root)r, "nf-isam_amd")); sys.path.insert(0, %(root)r)
import nfisam_hip as nh
dev = torch.device("cuda", 0)
K, H, B, R = 9, 8, 5.0, 3
gen = torch.Generator().manual_seed(5)
xs = [(1.3 * torch.randn(2000, 15, generator=gen)).clamp_(-4, 4).to(dev) for _ in range(R)]
kp0 = [nh.pack((0.2 * torch.randn(nh.param_count(15, K, H), generator=gen)).to(dev), 15, K, H, 1) for _ in range(R)]
tb = nh.TrainBatch(xs, [k.clone() for k in kp0], K, H, B, 1, lr=0.01, max_iters=100, average_window=50, loss_delta_tol=0.0, early_stop=True)
outcome = "ran"
try:
    done = tb.run()
    sys.stderr.write("no stall; XCDs per group of the persistent launches: %%d (0: the plan did not take that form)\\n" %% tb.xcd_span())
except nh.PersistentStall:
    outcome = "stalled"
    tb.reset(kp0)                    # the same fit again: the library keeps to one launch per iteration now
    done = tb.run()
torch.cuda.synchronize()
np.savez(sys.argv[1], outcome=np.array(outcome), iters=np.array(done), params=torch.stack([k for k in tb.kparams]).cpu().numpy(),
         loss=torch.stack([l for l in tb.iter_loss]).cpu().numpy())
tb.close()
'''


KNOB_WORKER = r'''
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(%(root)r, "nf-isam_amd")); sys.path.insert(0, %(root)r)
import nfisam_hip as nh
dev = torch.device("cuda", 0)
K, B = 9, 5.0
out = {}
for name, shapes, iters, window, tol, H, Kk in (("plaza", [(2000, 15)], 230, 50, 0.0, 8, 9), ("n1500", [(1500, 12)], 120, 40, 0.0, 8, 9),
                                                ("n1000", [(1000, 15)], 130, 50, 0.0, 8, 9), ("early_stop", [(2000, 7)], 600, 50, 0.05, 8, 9),
                                                ("k5", [(2000, 9)], 120, 40, 0.0, 8, 5), ("k12", [(1200, 10)], 120, 40, 0.0, 8, 12),
                                                ("h16", [(2000, 11)], 120, 40, 0.0, 16, 9), ("h4", [(900, 9)], 100, 50, 0.0, 4, 9)):
    gen = torch.Generator().manual_seed(len(name))
    xs = [(1.3 * torch.randn(n, D, generator=gen)).clamp_(-4, 4).to(dev) for n, D in shapes]
    kp = [nh.pack((0.2 * torch.randn(nh.param_count(D, Kk, H), generator=gen)).to(dev), D, Kk, H, 1) for n, D in shapes]
    tb = nh.TrainBatch(xs, kp, Kk, H, B, 1, lr=0.01, max_iters=iters, average_window=window, loss_delta_tol=tol, early_stop=True)
    done = tb.run()
    for c in range(len(shapes)):
        out["%%s_%%d_params" %% (name, c)] = tb.kparams[c].cpu().numpy()
        out["%%s_%%d_loss" %% (name, c)] = tb.iter_loss[c].cpu().numpy()
        out["%%s_%%d_iters" %% (name, c)] = np.array(done[c])
    out["span_%%s" %% name] = np.array(tb.xcd_span())
    tb.close()
np.savez(sys.argv[1], **out)
'''


@pytest.mark.timeout(600)
def test_round6_launch_forms_are_bit_identical_to_one_launch_per_iteration(tmp_path):
    """Round 6's forms of a lone clique's chunk-persistent launch, each against ITS kernel family's one-launch-per-iteration form
    (NFISAM_PERSIST=0; the two families -- 64 particles per wave / two lanes per particle -- differ from each other in rounding, a form
    of one family does not): the default (two lanes per particle up to 1024 particles, helper waves, one kernel closing a chunk),
    NFISAM_HELPERS=0 (four waves per block), NFISAM_FUSED_CLOSE=0 (closing Adam and bookkeeping as two kernels), and NFISAM_HALF=2
    (two lanes per particle with nine to sixteen copies: 240 blocks of one clique, two per slot of the 128-slot loss ring -- with 64
    slots up to four blocks shared a slot and the loss record depended on their order).  Shapes: a Plaza clique, 1500 and 1000
    particles, a run that stops early, num_knots 5 and 12 (two-wave builds added at the round's end), hidden_dim 16 and 4.
    Parameters, loss records and stop iterations equal bit for bit; every persistent run really took that form (XCD span 1)."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "knob_worker.py"
    script.write_text(KNOB_WORKER % dict(root=root))

    def run(tag, **env):
        out = str(tmp_path / ("knob_%s.npz" % tag))
        p = subprocess.run([sys.executable, str(script), out], env=dict(os.environ, **env), capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stderr[-2000:]
        return dict(np.load(out))

    plain = run("plain", NFISAM_PERSIST="0")
    plain_half2 = run("plain_half2", NFISAM_PERSIST="0", NFISAM_HALF="2")
    for tag, env, ref in (("default", dict(NFISAM_PERSIST="1"), plain),
                          ("no_helpers", dict(NFISAM_PERSIST="1", NFISAM_HELPERS="0"), plain),
                          ("two_kernel_close", dict(NFISAM_PERSIST="1", NFISAM_FUSED_CLOSE="0"), plain),
                          ("half2", dict(NFISAM_PERSIST="1", NFISAM_HALF="2"), plain_half2),
                          ("half2_no_helpers", dict(NFISAM_PERSIST="1", NFISAM_HALF="2", NFISAM_HELPERS="0"), plain_half2)):
        got = run(tag, **env)
        assert got.keys() == ref.keys()
        for k in got:
            if k.startswith("span_"):
                assert int(got[k]) == 1, (tag, k, got[k])
                continue
            np.testing.assert_array_equal(got[k], ref[k], err_msg="%s: %s" % (tag, k))


@pytest.mark.timeout(400)
def test_oversubscribed_persistent_launch_stalls_loudly_and_the_rerun_is_exact(tmp_path):
    """A chunk-persistent launch whose blocks do not all arrive must not hang and must not pass for a numerical failure.
    Provoked with a test knob (oversubscribing the machine does not do it: blocks are dispatched in order, 1440 blocks on 768
    places merely ran one behind the other -- measured): NFISAM_PERSIST_DROP=1 makes one block of one (clique, dim) group
    leave at once, like a member whose place is held by a foreign process that never yields; NFISAM_PERSIST_SPINS=12 shortens
    the wait from 2^15 looks (tens of milliseconds; round 4: 2^22, seconds) to 4096.  The starved group's blocks give up, raise the group's abort flag and leave,
    the run ends with NFISAM_ERR_STALL (nfisam_hip.PersistentStall), the process switches to one launch per iteration, and
    the same fit run again gives exactly what NFISAM_PERSIST=0 gives."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "stall_worker.py"
    script.write_text(STALL_WORKER % dict(root=root))
    res = {}
    for name, env in (("forced", dict(NFISAM_PERSIST="1", NFISAM_PERSIST_DROP="1", NFISAM_PERSIST_SPINS="12")), ("plain", dict(NFISAM_PERSIST="0"))):
        out = str(tmp_path / (name + ".npz"))
        p = subprocess.run([sys.executable, str(script), out], env=dict(os.environ, **env), capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stderr[-2000:]
        res[name] = dict(np.load(out))
        if name == "forced":
            assert "stalled" in p.stderr, p.stderr[-500:]            # the library says so once
    assert str(res["forced"]["outcome"]) == "stalled" and str(res["plain"]["outcome"]) == "ran"
    assert list(res["forced"]["iters"]) == list(res["plain"]["iters"]) == [100] * 3
    np.testing.assert_array_equal(res["forced"]["params"], res["plain"]["params"])
    np.testing.assert_array_equal(res["forced"]["loss"], res["plain"]["loss"])


@pytest.mark.timeout(300)
def test_a_busy_device_is_probed_before_a_plan_takes_the_persistent_form():
    """"Probe before persisting" (nsf_kernels.hip: device_is_quiet): the occupancy API answers for this process's kernel alone;
    whether the device really holds all blocks of a chunk-persistent launch AT ONCE right now is asked with one launch of as
    many trivial blocks of the same footprint that must all arrive at a counter within ~200 us of their own start.  On a quiet
    device the plan of the C3 batch (624 blocks, three per CU) takes the persistent form (`xcd_span() >= 1`: a persistent chunk
    ran).  While somebody else's long kernel holds PART of every compute unit -- here the library's diagnostic occupier in a
    SECOND PROCESS: one spinning block per CU with 100 KB of its 160 KB of LDS for 2.5 s -- only one block of the plan's footprint fits a CU: the probe's first 256 blocks wait in vain for the
    other 368, say so, and the NEW plan keeps to one launch per iteration (`xcd_span() == 0`, one line on stderr), with the
    same bits.  (Without the probe that plan would have taken the persistent form, 256 of its blocks would have spun for
    members queued behind the occupier, and the fit would have ended with NFISAM_ERR_STALL.)  NFISAM_PERSIST_PROBE=0: not asked,
    the quiet plan is persistent all the same."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = PROBE_WORKER % dict(root=root)
    outs = {}
    for name, env in (("probe", {}), ("no-probe", dict(NFISAM_PERSIST_PROBE="0", PROBE_TEST_SKIP_BUSY="1"))):
        p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=280)
        assert p.returncode == 0, p.stderr[-2000:]
        outs[name] = (json.loads(p.stdout.strip().splitlines()[-1]), p.stderr)
    r, err = outs["probe"]
    assert r["quiet_span"] >= 1, r                                   # quiet device: the persistent form ran
    assert r["busy_span"] == 0 and "another process is using it" in err, (r, err[-500:])
    assert r["equal"] and r["iters"] == [[100] * 8, [100] * 8]
    r0, _ = outs["no-probe"]
    assert r0["quiet_span"] >= 1, r0


PROBE_WORKER = r"""
import ctypes, json, os, sys, time
sys.path.insert(0, %(root)r + "/nf-isam_amd"); sys.path.insert(0, %(root)r)
import numpy as np, torch
import nfisam_hip as nh
K, H, B = 9, 8, 5.0
dev = torch.device("cuda", 0)
gen = torch.Generator().manual_seed(5)
shapes = [(2000, D) for D in (6, 8, 8, 10, 10, 12, 12, 12)]          # the C3 batch: 624 four-wave blocks, three per CU on 208 CUs
xs = [(1.2 * torch.randn(n, D, generator=gen)).clamp_(-4, 4).to(dev) for n, D in shapes]
kp0 = [nh.pack((0.2 * torch.randn(nh.param_count(D, K, H), generator=gen)).to(dev), D, K, H, 1) for n, D in shapes]
OCCUPIER = (
    "import ctypes, sys, time\n"
    "sys.path.insert(0, %(root)r + '/nf-isam_amd')\n"
    "import torch, nfisam_hip as nh\n"
    "cus = torch.cuda.get_device_properties(0).multi_processor_count\n"
    "s = torch.cuda.Stream()\n"
    "import os\n"
    "lib = ctypes.CDLL(os.path.join(os.path.dirname(nh.LIB_PATH), 'libnfisam_diag.so'))   # the diagnostic library (csrc/nsf_diag.hip), not the product\n"
    "assert lib.nfisam_diag_occupy_device(cus, ctypes.c_size_t(100 * 1024), ctypes.c_float(0.01), ctypes.c_void_p(s.cuda_stream)) == 0\n"
    "s.synchronize()\n"      # (the first launch pays the kernel's code load)
    "assert lib.nfisam_diag_occupy_device(cus, ctypes.c_size_t(100 * 1024), ctypes.c_float(2.5), ctypes.c_void_p(s.cuda_stream)) == 0\n"
    "time.sleep(0.05)\n"
    "print('occupying', flush=True)\n"
    "s.synchronize()\n")
def fit():
    tb = nh.TrainBatch(xs, [k.clone() for k in kp0], K, H, B, 1, lr=0.01, max_iters=100, average_window=50, loss_delta_tol=0.0, early_stop=True)
    tb.prepare(True)                      # <- the probe runs here (plan creation)
    it = tb.run()
    torch.cuda.synchronize()
    out = (tb.xcd_span(), it, [k.cpu().numpy().copy() for k in tb.kparams], [l.cpu().numpy().copy() for l in tb.iter_loss])
    tb.close()
    return out
quiet = fit()
busy = quiet
if os.environ.get("PROBE_TEST_SKIP_BUSY") != "1":
    time.sleep(0.6)                       # (the probe's answer is cached for half a second)
    # a FOREIGN PROCESS holds part of every CU: one spinning block per CU with 100 KB of its 160 KB of LDS for 2.5 s -- room for
    # ONE more block of the plan's footprint per CU.  (A second process, not a side stream of this one: two streams of a
    # process may share a hardware queue, and then the probe simply queues behind the occupier.)
    import subprocess
    occ = subprocess.Popen([sys.executable, "-c", OCCUPIER], stdout=subprocess.PIPE, text=True)
    assert occ.stdout.readline().strip() == "occupying", "the occupier process did not start"
    time.sleep(0.1)
    busy = fit()
    torch.cuda.synchronize()
    occ.wait()
eq = all(np.array_equal(a, b) for a, b in zip(quiet[2], busy[2])) and all(np.array_equal(a, b) for a, b in zip(quiet[3], busy[3]))
print(json.dumps(dict(quiet_span=quiet[0], busy_span=busy[0], iters=[quiet[1], busy[1]], equal=bool(eq))))
"""


@pytest.mark.timeout(120)
def test_concurrent_plan_runs_share_the_persistent_form_safely():
    """Two threads run Plaza-shaped plans at the same time: only ONE run of the process may use the chunk-persistent graph
    (two persistent launches could starve each other's late blocks), the other takes the plain graph -- and both end with
    exactly what each of them gives alone."""
    import threading
    K, H, B = 9, 8, 5.0
    dev = torch.device("cuda", 0)

    def problem(seed):
        gen = torch.Generator().manual_seed(seed)
        x = (1.2 * torch.randn(2000, 15, generator=gen)).clamp_(-4, 4).to(dev)
        kp = nh.pack((0.2 * torch.randn(nh.param_count(15, K, H), generator=gen)).to(dev), 15, K, H, 1)
        return x, kp

    def run(seed, out):
        x, kp = problem(seed)
        tb = nh.TrainBatch([x], [kp], K, H, B, 1, lr=0.01, max_iters=400, average_window=50, loss_delta_tol=0.0, early_stop=True)
        done = tb.run()
        torch.cuda.synchronize()
        out[seed] = (tb.kparams[0].cpu().numpy().copy(), tb.iter_loss[0].cpu().numpy().copy(), done[0])
        tb.close()

    alone = {}
    for seed in (1, 2, 3):
        run(seed, alone)
    for rep in range(3):
        together = {}
        threads = [threading.Thread(target=run, args=(seed, together)) for seed in (1, 2, 3)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        for seed in (1, 2, 3):
            assert together[seed][2] == alone[seed][2] == 400
            np.testing.assert_array_equal(together[seed][0], alone[seed][0])
            np.testing.assert_array_equal(together[seed][1], alone[seed][1])


# ---- hold-out validation on the device (round 4) -----------------------------------------------------------------------------
@pytest.mark.parametrize("use_graph", [True, False], ids=["graph", "eager"])
@pytest.mark.parametrize("case", ["overfit", "interval7", "budget"])
def test_validated_training_plan_against_the_reference_loop(case, use_graph):
    """nfisam_nsf_train_plan_create_validated (the reference's hold-out stop rule, src/slam/NFiSAM.py:452-468, evaluated on the
    device as part of the training plan: one graph replay per validation period, the first interval - 1 iterations of a period
    as one chunk-persistent launch) against the REFERENCE'S OWN LOOP run on the reference's flow classes
    (tests/golden/validation_loop.npz, make_golden.py: gen_validation_loop): iterations run (within one validation interval:
    two float32 implementations may disagree about a marginal comparison; measured: equal), every validation loss, the
    loss record (tight over the first 30 iterations, then within rounding-amplified noise) and the final parameters."""
    from test_oracle_golden import validation_case
    x, xv, b0, b1, D, K, H, iters, interval, lr, rate, g = validation_case(case)
    tb = nh.TrainBatch([dev(x.numpy())], [kpack(b0.numpy(), D, K, H)], K, H, 5.0, 1, lr=lr, max_iters=iters, x_val=[dev(xv.numpy())],
                       validation_interval=interval, slower_stop_rate=rate)
    run = tb.run(use_graph=use_graph)[0]
    ref_run = int(g[case + "_iters_run"])
    assert abs(run - ref_run) <= interval, (run, ref_run)
    il = tb.iter_loss[0].cpu().numpy()
    ref_il, ref_vals = g[case + "_iter_loss"], g[case + "_val_losses"]
    vals = tb.val_loss[0].cpu().numpy()
    n_eval = min(len(ref_vals), int(np.count_nonzero(vals)))
    assert n_eval >= len(ref_vals) - 1
    np.testing.assert_allclose(vals[:n_eval], ref_vals[:n_eval], rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(il[:30], ref_il[:30], rtol=5e-4, atol=5e-4)
    both = min(run, ref_run)
    assert np.median(np.abs(il[:both] - ref_il[:both])) < 5e-3
    assert np.all(il[run:] == 0)
    st = tb.state(0)
    if case == "budget":
        assert run == iters and st["slower_stop_iter"] == 0
    else:
        assert run < iters and st["slower_stop_iter"] == run + 1 and st["stop"] == 1     # the loop breaks in front of iteration slower_stop_iter - 1
        assert run == ref_run, (run, ref_run)
    got = nh.unpack(tb.kparams[0], D, K, H, 1).cpu().numpy()
    # Final parameters: an over-fitting Adam run amplifies float32 rounding, so the bar is the float32 noise of THIS problem --
    # three times the distance between the float64 oracle's run and the reference's float32 run (overfit: 2.6e-3) + 2e-3 --
    # instead of a constant tuned to one kernel family (round 6: the two-lanes-per-particle family measures 7.8e-3 on
    # `overfit`, the 64-particle family 3-4e-3; both well inside)
    from test_oracle_golden import O
    b64 = O.train_with_validation(x.double(), xv.double(), b0.double(), K, H, 5.0, 1, lr=lr, max_iters=iters, validation_interval=interval,
                                  slower_stop_rate=rate)[0]
    noise = float(np.median(np.abs(np.asarray(b64) - b1.numpy())))
    err = float(np.quantile(np.abs(got - b1.numpy()), 0.5))
    print("validated plan %s: median |parameters - reference| %.2e, float64-oracle-to-reference %.2e" % (case, err, noise))
    assert err < 3.0 * noise + 2e-3 and np.all(np.isfinite(got))
    tb.close()


def test_validated_plan_rejects_what_the_device_rule_does_not_cover():
    x = dev(np.random.RandomState(0).randn(128, 3))
    kp = kpack(O.init_blob(3, 9, 8, torch.Generator().manual_seed(0)).numpy(), 3, 9, 8)
    for interval, rate in ((10, 1.5), (0, 2.0), (200, 2.0), (10, 0.5)):
        tb = nh.TrainBatch([x], [kp.clone()], 9, 8, 5.0, 1, lr=0.01, max_iters=50, x_val=[x], validation_interval=interval, slower_stop_rate=rate)
        with pytest.raises(ValueError):
            tb.prepare(True)


def test_timing_plan_reports_the_persistent_launch_and_changes_nothing():
    """`prepare(timing=True)` (bench.py's `roofline.kernel_us`): the persistent chunk becomes two graphs with two timing events on
    the stream around the training launch -- a positive duration below the whole run's, and the same bits as the ordinary plan."""
    K, H, B = 9, 8, 5.0
    gen = torch.Generator().manual_seed(21)
    x = (1.2 * torch.randn(2000, 9, generator=gen)).clamp_(-4, 4).to(DEV)
    kp = nh.pack((0.2 * torch.randn(nh.param_count(9, K, H), generator=gen)).to(DEV), 9, K, H, 1)
    outs = []
    for timing in (False, True):
        tb = nh.TrainBatch([x], [kp.clone()], K, H, B, 1, lr=0.01, max_iters=100, average_window=50, loss_delta_tol=0.0, early_stop=True)
        tb.prepare(use_graph=True, timing=timing)
        t0 = time.perf_counter()
        assert tb.run(use_graph=True) == [100]
        torch.cuda.synchronize()
        wall_ms = 1e3 * (time.perf_counter() - t0)
        if timing:
            assert tb.xcd_span() >= 1                      # the plan took the persistent form
            ms = tb.kernel_ms()
            assert 0.05 < ms < wall_ms, (ms, wall_ms)      # one chunk of 50 iterations: a few hundred microseconds
        else:
            with pytest.raises(ValueError):
                tb.kernel_ms()
        outs.append((tb.kparams[0].clone(), tb.iter_loss[0].clone()))
        tb.close()
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("n, D, iters, window, tol", [(2000, 15, 400, 50, 0.02), (1000, 12, 230, 50, 0.0), (600, 7, 600, 100, 0.15)],
                         ids=["plaza-stops-early", "half-family-partial-last-window", "window-100"])
def test_window_spanning_launch_equals_one_launch_per_chunk(n, D, iters, window, tol):
    """Round 6 (VERDICT r5 next #1b): `NFISAM_SPAN=1` runs a single clique's whole fit as ONE chunk-persistent launch whose blocks
    close every window themselves -- the clique's block that arrives last at a window's end runs the bookkeeping (csrc/nsf_bookkeep.h:
    the code of `nsf_bookkeep_kernel`), everybody reads its decision --, followed by the closing Adam kernel.  Same arithmetic in the
    same order as one launch per window + Adam kernel + bookkeeping kernel: iterations run, parameters, both moments and the loss
    record must agree bit for bit -- with the rule firing at a window's end, with a budget that ends inside a window, with the
    two-lanes-per-particle family (n <= 1024) and the 64-particle family's two-wave build.  (Off by default: it gains 1.5 % on
    Plaza1's fits)"""
    K, H, B, L = 9, 8, 5.0, 1
    blob, x = make_problem(n, D, K, H, L, seed=4242 + D, spread=1.0)
    out = {}
    for span in ("0", "1"):
        with _Env(NFISAM_SPAN=span):
            tb = nh.TrainBatch([dev(x)], [kpack(blob, D, K, H, L)], K, H, B, L, lr=0.02, max_iters=iters, average_window=window, loss_delta_tol=tol,
                               early_stop=True)
            it = tb.run(use_graph=True)
            torch.cuda.synchronize()
            out[span] = (it[0], tb.kparams[0].cpu().numpy().copy(), tb.m[0].cpu().numpy().copy(), tb.v[0].cpu().numpy().copy(),
                         tb.iter_loss[0].cpu().numpy().copy(), dict(tb.state(0)))
            # a second run of the same plan from a reset state: the control words of the workspace are back where a launch expects them
            tb.reset([kpack(blob, D, K, H, L)])
            it2 = tb.run(use_graph=True)
            torch.cuda.synchronize()
            assert it2 == it and np.array_equal(tb.kparams[0].cpu().numpy(), out[span][1])
            tb.close()
    a, b = out["0"], out["1"]
    assert a[0] == b[0] and (tol == 0.0 or a[0] < iters), (a[0], b[0])
    for q in range(1, 5):
        assert np.array_equal(a[q], b[q]), q
    assert a[5]["step"] == b[5]["step"] and a[5]["stop"] == b[5]["stop"]
    assert np.all(np.isfinite(b[1])) and b[4][b[0] - 1] < b[4][0]
