import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for q in ("tests", "nf-isam_amd", ""): sys.path.insert(0, os.path.join(ROOT, q))
import numpy as np, torch
import test_hip_parity as T
nh, CO = T.nh, T.CO
K, B, H, n, D = 9, 5.0, 8, 600, 19
blob, x0 = T.make_problem(n, D, K, H, 1, seed=77 + D)
def err(x, tag):
    lossc, gradc, _, _ = CO.nll_grad(x, blob, K, H, B, 1, dtype=np.float64, want_gx=True)
    with T._Env(NFISAM_DIM_MAJOR="0", NFISAM_TRAIN="wide"):
        kg, _, loss = nh.backward(T.dev(x), T.kpack(blob, D, K, H), K, H, B, 1, nll_mode=True)
    g = nh.pack(torch.from_numpy(gradc.astype(np.float32)).to(T.DEV), D, K, H, 1).cpu().numpy()
    e = np.abs(kg.cpu().numpy() / n - g)
    j = int(np.argmax(e))
    # which dim block?
    PoP = 32; off = PoP; dim = 0
    for i in range(1, D):
        cnt = i * H + H + H * H + H + H * PoP + PoP
        if off <= j < off + cnt: dim = i; break
        off += cnt
    print(tag, "max err %.2e at kernel index %d (dim %d, offset in block %d) loss dev %.6f oracle %.6f" % (e.max(), j, dim, j - off if dim else j, loss.item() / n + 0.5 * D * np.log(2 * np.pi), lossc))
err(x0, "as planted      ")
for r, c in ((0, 0), (1, D - 1), (2, D // 2), (3, 0)):
    x = x0.copy(); x[r, c] = 0.3
    err(x, "row %d col %2d -> 0.3" % (r, c))
# per-particle: which particle carries the error?  drop particles one block at a time
for lo in range(0, n, 100):
    x = np.delete(x0, slice(lo, lo + 100), axis=0)
    lossc, gradc, _, _ = CO.nll_grad(x, blob, K, H, B, 1, dtype=np.float64, want_gx=True)
    with T._Env(NFISAM_DIM_MAJOR="0", NFISAM_TRAIN="wide"):
        kg, _, loss = nh.backward(T.dev(x), T.kpack(blob, D, K, H), K, H, B, 1, nll_mode=True)
    g = nh.unpack(kg, D, K, H, 1).cpu().numpy() / len(x)
    print("without rows %d..%d: max err %.2e" % (lo, lo + 99, np.abs(g - gradc).max()))
# narrow down to the particle
def maxerr(x):
    lossc, gradc, _, _ = CO.nll_grad(x, blob, K, H, B, 1, dtype=np.float64, want_gx=True)
    with T._Env(NFISAM_DIM_MAJOR="0", NFISAM_TRAIN="wide"):
        kg, _, loss = nh.backward(T.dev(x), T.kpack(blob, D, K, H), K, H, B, 1, nll_mode=True)
    return np.abs(nh.unpack(kg, D, K, H, 1).cpu().numpy() / len(x) - gradc).max()
bad = None
for r in range(400, 500):
    if maxerr(np.delete(x0, r, axis=0)) < 1e-3:
        bad = r; break
print("particle", bad, "x[:8] =", x0[bad, :8])
xs = x0[bad:bad + 1]
lossc, gradc, _, gxc = CO.nll_grad(xs, blob, K, H, B, 1, dtype=np.float64, want_gx=True)
z64, ld64 = CO.forward(xs, blob, K, H, B, 1, dtype=np.float64)
for env in (dict(NFISAM_DIM_MAJOR="0", NFISAM_TRAIN="wide"), dict(NFISAM_TRAIN="split", NFISAM_PAIR="0"), dict(NFISAM_TRAIN="split")):
    with T._Env(**env):
        kg, gx, loss = nh.backward(T.dev(xs), T.kpack(blob, D, K, H), K, H, B, 1, nll_mode=True, want_gx=True)
    g = nh.unpack(kg, D, K, H, 1).cpu().numpy()
    print(env, "single particle: grad max err %.3e (|grad| max %.3e), gx err %.3e" % (np.abs(g - gradc).max(), np.abs(gradc).max(), np.abs(gx.cpu().numpy() - gxc).max()))
z, ld, _ = nh.forward(T.dev(xs), T.kpack(blob, D, K, H), K, H, B, 1)
print("forward z err", np.abs(z.cpu().numpy() - z64).max(), "z64[:8]", z64[0, :8])
