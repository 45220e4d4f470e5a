#!/usr/bin/env python3
"""Generate golden vectors for the NF-iSAM flow hot path from the *reference itself*.

Runs ONLY in the build container (needs /root/reference).  Nothing from the reference is
copied: the reference's `flows.*` modules are imported read-only and executed on seeded
inputs; inputs and outputs are stored as small float32/float64 fixtures:

    tests/golden/nsf_<case>.npz        flow forward / loss / grads / Adam / inverse   (a1-a8)
    tests/golden/rqs_direct.npz        direct unconstrained_RQS / RQS / searchsorted  (a5, a6)
    tests/golden/normalize.npz         normalize_training_samples + (un)normalize     (a9, a10)

Reference entry points exercised (paths relative to /root/reference):
    src/flows/flows.py:43-137   NSF_AR.{forward,inverse,inverse_given_separator}
    src/flows/models.py:4-40    NormalizingFlowModel.forward
    src/flows/prior_dist.py:5-26 CustomMultivariateNormal
    src/flows/utils.py:17-164   searchsorted / unconstrained_RQS / RQS
    src/slam/NFiSAM.py:96-118,515-548  normalisation helpers (function bodies are executed via
        `ast` extraction because `slam.NFiSAM` imports TransportMaps/dynesty which are absent;
        only those three function definitions are compiled, in memory, never written to disk)

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
}


def unscramble(z_ref, ld_ref_elem, n, d):
    """Reference forward evaluates the spline on a dim-major flat vector and then reshapes
    it as (n, d) (src/flows/flows.py:88-93).  The mathematically correct layout is the
    (d, n) reshape transposed."""
    zc = z_ref.reshape(-1).reshape(d, n).T.copy()
    return zc


def state_to_np(sd):
    return {k.replace(".", "__"): v.detach().numpy().astype(np.float32) for k, v in sd.items()}


def gen_flow_case(name, n, D, K, seed):
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


if __name__ == "__main__":
    for name, (n, D, K, seed) in CASES.items():
        gen_flow_case(name, n, D, K, seed)
    gen_rqs_direct()
    gen_normalize()
