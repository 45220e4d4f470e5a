"""The oracle (oracle/nsf_torch.py, oracle C library) against golden vectors produced by
running the reference itself (tests/golden/make_golden.py).  CPU only."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import nsf_torch as O

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
CASES = sorted(glob.glob(os.path.join(GOLDEN, "nsf_*.npz")))
ATOL = 1e-5   # CPU restatement vs reference, fp32
RTOL = 1e-4


def load(path):
    g = dict(np.load(path))
    n, D, K, H, seed = [int(v) for v in g["meta"]]
    return g, n, D, K, H, float(g["B"])


def sd_of(g, prefix):
    return {k[len(prefix) + 2:].replace("__", "."): v for k, v in g.items() if k.startswith(prefix + "__")}


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[4:-4] for p in CASES])
class TestFlowGolden:
    def test_param_count(self, path):
        g, n, D, K, H, B = load(path)
        blob = O.blob_from_state_dict(sd_of(g, "p0"), D)
        assert blob.size == O.param_count(D, K, H)

    def test_forward_and_loss(self, path):
        g, n, D, K, H, B = load(path)
        blob = torch.tensor(O.blob_from_state_dict(sd_of(g, "p0"), D))
        x = torch.tensor(g["x"])
        theta = O.layer_theta(x, blob, K, H)
        np.testing.assert_allclose(theta.numpy().transpose(1, 0, 2), g["spline_params"], atol=ATOL, rtol=RTOL)
        z, ld = O.forward(x, blob, K, H, B)
        np.testing.assert_allclose(z.numpy(), g["z"], atol=ATOL, rtol=RTOL)
        np.testing.assert_allclose(ld.numpy(), g["logdet"], atol=2e-5, rtol=RTOL)
        # the reference's own (scrambled) return value is the fixed permutation of the correct one
        np.testing.assert_allclose(z.numpy().T.reshape(-1).reshape(n, D), g["z_raw"], atol=ATOL, rtol=RTOL)
        lp = O.log_prob(x, blob, K, H, B)
        np.testing.assert_allclose((lp - ld).numpy(), g["prior_logprob"], atol=2e-5, rtol=RTOL)
        assert abs(O.nll(x, blob, K, H, B).item() - float(g["loss"])) < 2e-5 * max(1, abs(float(g["loss"])))

    def test_gradients(self, path):
        g, n, D, K, H, B = load(path)
        blob = torch.tensor(O.blob_from_state_dict(sd_of(g, "p0"), D))
        _, grad = O.loss_and_grad(torch.tensor(g["x"]), blob, K, H, B)
        gref = O.blob_from_state_dict(sd_of(g, "g0"), D)
        np.testing.assert_allclose(grad.numpy(), gref, atol=ATOL, rtol=1e-3)

    def test_adam_trajectory(self, path):
        g, n, D, K, H, B = load(path)
        blob = torch.tensor(O.blob_from_state_dict(sd_of(g, "p0"), D))
        x = torch.tensor(g["x"])
        for steps in (1, 2, 10):
            b, losses, iters = O.train(x, blob, K, H, B, lr=float(g["adam_lr"]), max_iters=steps, early_stop=False)
            assert iters == steps
            ref = O.blob_from_state_dict(sd_of(g, "p%d" % steps), D)
            np.testing.assert_allclose(b.numpy(), ref, atol=2e-4, rtol=1e-3)
            np.testing.assert_allclose(losses.numpy(), g["adam_losses"][:steps], atol=1e-4, rtol=1e-4)

    def test_inverse(self, path):
        g, n, D, K, H, B = load(path)
        blob = torch.tensor(O.blob_from_state_dict(sd_of(g, "p0"), D))
        xr, ld = O.inverse(torch.tensor(g["z"]), blob, K, H, B)
        np.testing.assert_allclose(xr.numpy(), g["inv_x"], atol=5e-5, rtol=RTOL)
        np.testing.assert_allclose(ld.numpy(), g["inv_logdet"], atol=5e-5, rtol=RTOL)
        xl, ldl = O.inverse(torch.tensor(g["zlat"]), blob, K, H, B)
        np.testing.assert_allclose(xl.numpy(), g["zlat_inv_x"], atol=5e-5, rtol=RTOL)
        np.testing.assert_allclose(ldl.numpy(), g["zlat_inv_logdet"], atol=5e-5, rtol=RTOL)

    def test_inverse_given_separator(self, path):
        g, n, D, K, H, B = load(path)
        blob = torch.tensor(O.blob_from_state_dict(sd_of(g, "p0"), D))
        for Ds in (0, 1, 3):
            key = "igs%d_x" % Ds
            if key not in g:
                continue
            xs = None if Ds == 0 else torch.tensor(g["x"][:, :Ds])
            xf = O.inverse_given_separator(torch.tensor(g["zlat"][:, Ds:]), xs, blob, K, H, B)
            np.testing.assert_allclose(xf.numpy(), g[key], atol=5e-5, rtol=RTOL)


def test_rqs_direct():
    g = dict(np.load(os.path.join(GOLDEN, "rqs_direct.npz")))
    for tag in "abcd":
        W, Hh, Dd, inp = (torch.tensor(g["%s_%s" % (tag, k)]) for k in ("W", "H", "D", "inp"))
        K = W.shape[1]
        tb = float(g["%s_tb" % tag])
        theta = torch.cat([W, Hh, Dd], 1)
        y, ld = O.rqs(inp, theta, K, tb, inverse=False)
        np.testing.assert_allclose(y.numpy(), g[tag + "_fwd"], atol=ATOL, rtol=RTOL)
        np.testing.assert_allclose(ld.numpy(), g[tag + "_fwd_ld"], atol=2e-5, rtol=RTOL)
        xi, ldi = O.rqs(inp, theta, K, tb, inverse=True)
        np.testing.assert_allclose(xi.numpy(), g[tag + "_inv"], atol=2e-5, rtol=RTOL)
        np.testing.assert_allclose(ldi.numpy(), g[tag + "_inv_ld"], atol=5e-5, rtol=RTOL)


def test_searchsorted_semantics():
    g = dict(np.load(os.path.join(GOLDEN, "rqs_direct.npz")))
    idx = O._bin(torch.tensor(g["ss_bins"]), torch.tensor(g["ss_q"]))
    np.testing.assert_array_equal(idx.numpy(), g["ss_idx"])


def test_normalisation():
    g = dict(np.load(os.path.join(GOLDEN, "normalize.npz")))
    circ = [bool(c) for c in g["circular"]]
    td, mu, sd = O.normalize_training_samples(g["samples"], circ)
    np.testing.assert_allclose(mu, g["mean"], atol=1e-6, rtol=1e-6)
    np.testing.assert_allclose(sd, g["std"], atol=1e-7, rtol=1e-6)
    np.testing.assert_allclose(td, g["train_norm"], atol=1e-5, rtol=1e-5)
    qn = O.normalize_samples(g["q"], mu, sd, circ, 0)
    np.testing.assert_allclose(qn, g["q_norm_init0"], atol=1e-5, rtol=1e-5)
    un = O.unnormalize_samples(g["zz"], mu, sd, circ, 4)
    np.testing.assert_allclose(un, g["zz_unnorm_init4"], atol=1e-5, rtol=1e-5)
