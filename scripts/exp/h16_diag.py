"""Diagnostic (GPU): hidden_dim 16 multi-layer gradient through the pair kernel and through the generic kernel against the float64 oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "nf-isam_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import nfisam_hip as nh
from oracle import c_oracle as CO
import test_hip_parity as T
K, H, B = int(sys.argv[1]), 16, 5.0
for L in (1, 2, 3):
    for (n, D) in ((300, 5), (257, 8), (64, 1)):
        blob, x = T.make_problem(n, D, K, H, L, seed=900)
        lossc, gradc, _, gxc = CO.nll_grad(x, blob, K, H, B, L, dtype=np.float64, want_gx=True)
        res = {}
        for mode in ("pair", "wide"):
            if mode == "wide":
                os.environ["NFISAM_TRAIN"] = "wide"
            else:
                os.environ.pop("NFISAM_TRAIN", None)
            kg, gx, loss = nh.backward(T.dev(x), T.kpack(blob, D, K, H, L), K, H, B, L, nll_mode=True, want_gx=True)
            g = nh.unpack(kg, D, K, H, L).cpu().numpy() / n
            res[mode] = g
            err = np.abs(g - gradc)
            sc = np.abs(gradc).max()
            P = gradc.size // L
            per_layer = [float(err[l * P:(l + 1) * P].max() / sc) for l in range(L)]
            print("K %d L %d n %d D %d %s: loss err %.2e, max|dg|/max|g| per layer %s, gx err %.2e" %
                  (K, L, n, D, mode, abs(loss.item() / n + 0.5 * D * np.log(2 * np.pi) - lossc), ["%.1e" % v for v in per_layer],
                   float(np.abs(gx.cpu().numpy() / n - gxc).max() / max(1e-9, np.abs(gxc).max()))), flush=True)
        os.environ.pop("NFISAM_TRAIN", None)
