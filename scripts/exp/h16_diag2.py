"""Diagnostic (GPU): one multi-layer problem (argv: K H n D seed L) through the pair kernel and the generic kernel against the float64
AND the float32 oracle: how many gradient entries sit further than 1e-4 x max|g| from float64, and which ones (layer, dim)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "nf-isam_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import nfisam_hip as nh
from oracle import c_oracle as CO
import test_hip_parity as T
K, H, n, D, seed, L = (int(v) for v in sys.argv[1:7])
B = 5.0
blob, x = T.make_problem(n, D, K, H, L, seed=seed, spread=1.0)
_, g64, _, _ = CO.nll_grad(x, blob, K, H, B, L, dtype=np.float64, want_gx=True)
_, g32, _, _ = CO.nll_grad(x, blob, K, H, B, L, dtype=np.float32, want_gx=True)
sc = np.abs(g64).max()
P = g64.size // L
res = {"oracle32": np.asarray(g32, dtype=np.float64)}
for mode in ("pair", "wide"):
    if mode == "wide":
        os.environ["NFISAM_TRAIN"] = "wide"
    else:
        os.environ.pop("NFISAM_TRAIN", None)
    kg, gx, loss = nh.backward(T.dev(x), T.kpack(blob, D, K, H, L), K, H, B, L, nll_mode=True, want_gx=True)
    res[mode] = nh.unpack(kg, D, K, H, L).cpu().numpy().astype(np.float64) / n
os.environ.pop("NFISAM_TRAIN", None)
for name, g in res.items():
    err = np.abs(g - g64) / sc
    print("%-9s max %.2e  q99.9 %.2e  entries > 1e-4: %d of %d  per layer max %s" %
          (name, err.max(), np.quantile(err, 0.999), int((err > 1e-4).sum()), err.size, ["%.1e" % err[l * P:(l + 1) * P].max() for l in range(L)]))
d = np.abs(res["pair"] - res["wide"]) / sc
print("pair vs wide: max %.2e, entries > 1e-4: %d" % (d.max(), int((d > 1e-4).sum())))
