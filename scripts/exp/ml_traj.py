import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for q in ("tests", "nf-isam_amd", ""): sys.path.insert(0, os.path.join(ROOT, q))
import numpy as np, torch
import test_hip_parity as T
K, H, B, L, iters = 9, 8, 5.0, 3, 8
shapes = [(257, 4), (600, 6), (90, 7)]
res = {}
for env in ("0", "1"):
    os.environ["NFISAM_PAIR"] = env
    done, out, probs = T._train_layers(shapes, L, iters, 50, True, lr=0.02, early_stop=False)
    res[env] = out
for c, (n, D) in enumerate(shapes):
    blob, x = probs[c]
    bc, lc, ic, _, _ = T.CO.train(x, blob, K, H, B, L, lr=0.02, max_iters=iters, early_stop=False, dtype=np.float32)
    bd, ld, _, _, _ = T.CO.train(x, blob, K, H, B, L, lr=0.02, max_iters=iters, early_stop=False, dtype=np.float64)
    print(c, "old-oracle32", np.abs(res["0"][3][c][:iters] - lc).round(5))
    print(c, "new-oracle32", np.abs(res["1"][3][c][:iters] - lc).round(5))
    print(c, "o32-oracle64", np.abs(ld - lc).round(5))
