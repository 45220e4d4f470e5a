"""ORACLE (test infrastructure, NOT product code) — plain-PyTorch CPU restatement of the
NF-iSAM flow hot path, in the mathematically correct layout.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
file.  The product path (`nf-isam_amd/`) never does: it fails loudly when the HIP library is
missing.

Pinned against the reference itself: `tests/test_oracle_golden.py` checks every function here
against `tests/golden/*.npz`, which `tests/golden/make_golden.py` produced by importing and
running `/root/reference/src/flows/*` in the build container.

What is restated (reference file:line, relative to /root/reference):
  conditioner MLP  Linear-tanh-Linear-tanh-Linear      src/flows/flows.py:26-41
  autoregressive parameterisation, dim 0 = init_param   src/flows/flows.py:51-83
  knot construction (softmax/cumsum/pin, softplus)      src/flows/utils.py:85-103, 41-44
  bin search with +1e-6 on the last knot                src/flows/utils.py:17-22, 105-108
  rational-quadratic forward / inverse + log|det|       src/flows/utils.py:123-164
  linear tails outside [-B, B]                          src/flows/utils.py:31-49
  layer chaining + N(0,I) prior log-prob                src/flows/models.py:11-35
  NLL loss, Adam, window early-stop                     src/slam/NFiSAM.py:425,451-491
  the same loop with a held-out set (hold-out stop)     src/slam/NFiSAM.py:452-468   (pinned by tests/golden/validation_loop.npz:
                                                        the reference's own `for` statement run on its own flow classes)
  (un)normalisation with circular dims                  src/slam/NFiSAM.py:96-118,515-548

Deliberate difference from the reference: `forward` returns z and log-det in the correct
(n, D) layout (the reference scrambles them); multi-layer flows therefore
compose correctly here.  Tests compare against the reference after the fixed permutation.

Parameter blob ("torch layout", one per flow layer, float32), identical to the order of
`NSF_AR.parameters()` in the reference:
    init_param[P_o] | for i = 1..D-1:  W0[H,i] b0[H] W1[H,H] b1[H] W2[P_o,H] b2[P_o]
with P_o = 3K-1.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

MIN_W = 1e-3
MIN_H = 1e-3
MIN_D = 1e-3


def param_count(D, K, H):
    Po = 3 * K - 1
    return Po + sum(i * H + H + H * H + H + H * Po + Po for i in range(1, D))


def unpack(blob, D, K, H):
    """Split one layer's torch-layout blob into (init_param, [(W0,b0,W1,b1,W2,b2) for i=1..D-1])."""
    Po = 3 * K - 1
    off = 0

    def take(*shape):
        nonlocal off
        cnt = int(np.prod(shape))
        t = blob[off:off + cnt].reshape(*shape)
        off += cnt
        return t
    init = take(Po)
    nets = []
    for i in range(1, D):
        nets.append((take(H, i), take(H), take(H, H), take(H), take(Po, H), take(Po)))
    assert off == blob.numel()
    return init, nets


def blob_from_state_dict(sd, D, prefix=""):
    """Flatten a reference-style state_dict (keys init_param, layers.j.network.{0,2,4}.{weight,bias})."""
    parts = [np.asarray(sd[prefix + "init_param"], dtype=np.float32).ravel()]
    for j in range(D - 1):
        for m in (0, 2, 4):
            for w in ("weight", "bias"):
                parts.append(np.asarray(sd["%slayers.%d.network.%d.%s" % (prefix, j, m, w)],
                                        dtype=np.float32).ravel())
    return np.concatenate(parts)


def init_blob(D, K, H, generator=None):
    """Reference initialisation: init_param ~ U(-1/2, 1/2) (flows.py:62-63); Linear layers use the
    torch default (kaiming-uniform a=sqrt(5) => U(+-1/sqrt(fan_in)) for weight and bias)."""
    Po = 3 * K - 1
    parts = [torch.rand(Po, generator=generator) - 0.5]
    for i in range(1, D):
        for fan_in, shape in ((i, (H, i)), (i, (H,)), (H, (H, H)), (H, (H,)), (H, (Po, H)), (H, (Po,))):
            bound = 1.0 / math.sqrt(fan_in)
            parts.append(((torch.rand(*shape, generator=generator) * 2 - 1) * bound).reshape(-1))
    return torch.cat(parts).to(torch.float32)


# ------------------------------------------------------------------ spline ---------------
def _knots(unnorm, K, B, min_size):
    w = F.softmax(unnorm, dim=-1)
    w = min_size + (1 - min_size * K) * w
    c = torch.cumsum(w, dim=-1)
    c = F.pad(c, (1, 0))
    c = 2 * B * c - B
    c = torch.cat([torch.full_like(c[..., :1], -B), c[..., 1:-1], torch.full_like(c[..., :1], B)], -1)
    return c, c[..., 1:] - c[..., :-1]


def _bin(knots, v):
    kn = knots.clone()
    kn[..., -1] = kn[..., -1] + 1e-6
    return (v[..., None] >= kn).sum(-1) - 1


def rqs(v, theta, K, B, inverse=False):
    """Elementwise unconstrained rational-quadratic spline.
    v: [...], theta: [..., 3K-1] -> (out [...], logabsdet [...]).  Outside [-B,B] (or NaN):
    identity, log-det 0."""
    inside = (v >= -B) & (v <= B)
    vs = torch.where(inside, v, torch.zeros_like(v))
    uw, uh, ud = theta[..., :K], theta[..., K:2 * K], theta[..., 2 * K:]
    cw, wd = _knots(uw, K, B, MIN_W)
    ch, ht = _knots(uh, K, B, MIN_H)
    const = math.log(math.exp(1 - MIN_D) - 1)
    ud = F.pad(ud, (1, 1), value=const)
    der = MIN_D + F.softplus(ud)
    idx = _bin(ch if inverse else cw, vs).clamp(0, K - 1)[..., None]
    g = lambda t: t.gather(-1, idx)[..., 0]  # noqa: E731
    xk, dx, yk, dy = g(cw), g(wd), g(ch), g(ht)
    s = dy / dx
    d0, d1 = g(der), g(der[..., 1:])
    sig = d0 + d1 - 2 * s
    if inverse:
        dlt = vs - yk
        a = dlt * sig + dy * (s - d0)
        b = dy * d0 - dlt * sig
        c = -s * dlt
        disc = b * b - 4 * a * c
        t = (2 * c) / (-b - torch.sqrt(disc))
        out = t * dx + xk
    else:
        t = (vs - xk) / dx
    q = t * (1 - t)
    den = s + sig * q
    num_d = s * s * (d1 * t * t + 2 * s * q + d0 * (1 - t) * (1 - t))
    lad = torch.log(num_d) - 2 * torch.log(den)
    if inverse:
        lad = -lad
    else:
        out = yk + dy * (s * t * t + d0 * q) / den
    out = torch.where(inside, out, v)
    lad = torch.where(inside, lad, torch.zeros_like(lad))
    return out, lad


# ------------------------------------------------------------------ flow layer ----------
def conditioner(x, init, nets, upto=None):
    """theta[n, D', 3K-1] for dims 0..D'-1 given data x[n, >=D'-1]."""
    n = x.shape[0]
    Dp = (len(nets) + 1) if upto is None else upto
    th = [init.expand(n, -1)]
    for i in range(1, Dp):
        W0, b0, W1, b1, W2, b2 = nets[i - 1]
        h = torch.tanh(F.linear(x[:, :i], W0, b0))
        h = torch.tanh(F.linear(h, W1, b1))
        th.append(F.linear(h, W2, b2))
    return torch.stack(th, 1)


def layer_forward(x, blob, K, H, B):
    D = x.shape[1]
    init, nets = unpack(blob, D, K, H)
    theta = conditioner(x, init, nets)
    z, lad = rqs(x, theta, K, B, inverse=False)
    return z, lad.sum(1)


def layer_theta(x, blob, K, H):
    D = x.shape[1]
    init, nets = unpack(blob, D, K, H)
    return conditioner(x, init, nets)


def layer_inverse(z, blob, K, H, B, x_sep=None, D=None):
    """Sequential inverse.  With x_sep [n, Ds] given, the first Ds columns are fixed and only the
    remaining D-Ds columns are solved from z [n, D-Ds] (flows.py:115-137).  Returns
    (x_free [n, D-Ds], logdet [n])."""
    n = z.shape[0]
    Ds = 0 if x_sep is None else x_sep.shape[1]
    if D is None:
        D = Ds + z.shape[1]
    init, nets = unpack(blob, D, K, H)
    cols = [x_sep[:, k] for k in range(Ds)]
    ld = torch.zeros(n, dtype=z.dtype)
    for i in range(Ds, D):
        if i == 0:
            th = init.expand(n, -1)
        else:
            W0, b0, W1, b1, W2, b2 = nets[i - 1]
            xin = torch.stack(cols[:i], 1)
            h = torch.tanh(F.linear(xin, W0, b0))
            h = torch.tanh(F.linear(h, W1, b1))
            th = F.linear(h, W2, b2)
        xi, l = rqs(z[:, i - Ds], th, K, B, inverse=True)
        cols.append(xi)
        ld = ld + l
    return torch.stack(cols[Ds:], 1), ld


# ------------------------------------------------------------------ model ---------------
def split_layers(blob, D, K, H, L):
    P = param_count(D, K, H)
    assert blob.numel() == P * L
    return [blob[l * P:(l + 1) * P] for l in range(L)]


def forward(x, blob, K, H, B, L=1):
    """x[n,D] -> (z[n,D], logdet[n]) through L stacked layers (models.py:20-22, correct layout)."""
    D = x.shape[1]
    ld = torch.zeros(x.shape[0], dtype=x.dtype)
    for lb in split_layers(blob, D, K, H, L):
        x, l = layer_forward(x, lb, K, H, B)
        ld = ld + l
    return x, ld


def log_prob(x, blob, K, H, B, L=1):
    z, ld = forward(x, blob, K, H, B, L)
    D = x.shape[1]
    return -0.5 * (z * z).sum(1) - 0.5 * D * math.log(2 * math.pi) + ld


def nll(x, blob, K, H, B, L=1):
    return -log_prob(x, blob, K, H, B, L).mean()


def inverse(z, blob, K, H, B, L=1):
    D = z.shape[1]
    ld = torch.zeros(z.shape[0], dtype=z.dtype)
    for lb in reversed(split_layers(blob, D, K, H, L)):
        z, l = layer_inverse(z, lb, K, H, B)
        ld = ld + l
    return z, ld


def inverse_given_separator(z, x_sep, blob, K, H, B, L=1):
    """Normalised-space conditional inverse.  L == 1: NormalizingFlowModelWithSeparator.inverse_given_separator
    (src/slam/NFiSAM.py:140-155).  For L > 1 the reference conditions every layer on the same raw x_sep
    (NFiSAM.py:151-152), which is not the inverse of any composition; here layer l is conditioned on the given
    columns pushed through the marginal flow of layers 0..l-1, so that forward(cat(x_sep, result)) returns z."""
    Ds = 0 if x_sep is None else x_sep.shape[1]
    D = Ds + z.shape[1]
    layers = split_layers(blob, D, K, H, L)
    seps = [x_sep]
    for lb in layers[:-1]:
        if x_sep is None:
            seps.append(None)
            continue
        init, nets = unpack(lb, D, K, H)
        theta = conditioner(seps[-1], init, nets, upto=Ds)
        zs, _ = rqs(seps[-1], theta, K, B, inverse=False)
        seps.append(zs)
    for lb, xs in zip(reversed(layers), reversed(seps)):
        z, _ = layer_inverse(z, lb, K, H, B, x_sep=xs, D=D)
    return z


def loss_and_grad(x, blob, K, H, B, L=1):
    b = blob.detach().clone().requires_grad_(True)
    loss = nll(x, b, K, H, B, L)
    (g,) = torch.autograd.grad(loss, b)
    return loss.detach(), g


def train(x, blob, K, H, B, L=1, lr=0.015, max_iters=10, average_window=50, loss_delta_tol=1e-2,
          early_stop=True):
    """Full-batch Adam loop with the reference's window early-stop (NFiSAM.py:451-491, case
    training_set_frac = 1).  Returns (blob, iter_loss[max_iters] zero-padded, iters_run)."""
    b = blob.detach().clone().requires_grad_(True)
    opt = torch.optim.Adam([b], lr=lr)
    iter_loss = torch.zeros(max_iters, dtype=torch.float32)
    loss_avg = None
    iters = 0
    for i in range(max_iters):
        opt.zero_grad()
        loss = nll(x, b, K, H, B, L)
        iter_loss[i] = loss.detach()
        loss.backward()
        opt.step()
        iters = i + 1
        if early_stop and (i + 1) % average_window == 0:
            new = iter_loss[i - average_window + 1:i + 1].mean()
            if loss_avg is not None and loss_avg != 0.0:
                if abs(1.0 - new / loss_avg) < loss_delta_tol:
                    break
            loss_avg = new
    return b.detach(), iter_loss, iters


def train_with_validation(x, x_val, blob, K, H, B, L=1, lr=0.015, max_iters=10, validation_interval=10, slower_stop_rate=2.0):
    """Full-batch Adam loop with the reference's HOLD-OUT stop rule (NFiSAM.py:451-476, case `testing_data is not None`,
    i.e. training_set_frac < 1): in front of iteration i, if a slower stop is scheduled and i + 1 has reached it, break
    (:453-456); else, when (i + 1) % validation_interval == 0, evaluate the held-out NLL with the current parameters (:458-463)
    and, the first time it exceeds the previous evaluation, schedule the end at int(slower_stop_rate * (i + 1)) (:464-466),
    otherwise remember it (:467-468); then the training step (:469-474).  No window rule in this mode (:479).
    Returns (blob, iter_loss[max_iters] zero-padded, iters_run, [validation losses in evaluation order])."""
    b = blob.detach().clone().requires_grad_(True)
    opt = torch.optim.Adam([b], lr=lr)
    iter_loss = torch.zeros(max_iters, dtype=torch.float32)
    last_validation_loss = float("inf")
    slower_stop_iter = None
    val_losses = []
    iters = 0
    for i in range(max_iters):
        if slower_stop_iter is not None:
            if (i + 1) >= slower_stop_iter:
                break
        elif (i + 1) % validation_interval == 0:
            with torch.no_grad():
                new_loss = nll(x_val, b, K, H, B, L).detach()
            val_losses.append(float(new_loss))
            if new_loss > last_validation_loss:
                slower_stop_iter = int(slower_stop_rate * (i + 1))
            else:
                last_validation_loss = new_loss
        opt.zero_grad()
        loss = nll(x, b, K, H, B, L)
        iter_loss[i] = loss.detach()
        loss.backward()
        opt.step()
        iters = i + 1
    return b.detach(), iter_loss, iters, val_losses


# ------------------------------------------------------------------ normalisation -------
def wrap_pi(t):
    """src/utils/Functions.py:20-21"""
    return (t + np.pi) % (2 * np.pi) - np.pi


def normalize_training_samples(samples, circular):
    """numpy f64 [n,D] -> (f32 [n,D], f32 mean[D], f32 std[D])   (NFiSAM.py:515-548)."""
    from scipy.stats import circmean
    s = np.array(samples, dtype=np.float64, copy=True)
    circ = np.asarray(circular, dtype=bool)
    D = s.shape[1]
    mean = np.zeros(D)
    std = np.zeros(D)
    for c in range(D):
        if circ[c]:
            mean[c] = circmean(s[:, c], high=np.pi, low=-np.pi)
            s[:, c] = wrap_pi(s[:, c] - mean[c])
            std[c] = np.std(s[:, c])
        else:
            mean[c] = np.mean(s[:, c])
            std[c] = np.std(s[:, c])
            s[:, c] = s[:, c] - mean[c]
    std = np.clip(std, 1e-5, None)
    s = s / std
    return s.astype(np.float32), mean.astype(np.float32), std.astype(np.float32)


def normalize_samples(x, mean, std, circular, init_dim=0):
    """float32 semantics of NormalizingFlowModelWithSeparator.normalize_samples (NFiSAM.py:96-106)."""
    x = np.array(x, dtype=np.float32, copy=True)
    for c in range(x.shape[1]):
        m, s = np.float32(mean[c + init_dim]), np.float32(std[c + init_dim])
        if circular[c + init_dim]:
            x[:, c] = wrap_pi(x[:, c] - m).astype(np.float32) / s
        else:
            x[:, c] = (x[:, c] - m) / s
    return x


def unnormalize_samples(xn, mean, std, circular, init_dim=0):
    """NFiSAM.py:108-118"""
    x = np.array(xn, dtype=np.float32, copy=True)
    for c in range(x.shape[1]):
        m, s = np.float32(mean[c + init_dim]), np.float32(std[c + init_dim])
        v = x[:, c] * s + m
        x[:, c] = wrap_pi(v).astype(np.float32) if circular[c + init_dim] else v
    return x
