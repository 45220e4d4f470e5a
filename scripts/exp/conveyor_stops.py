import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "nf-isam_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import nfisam_hip as nh
from test_hip_parity import make_problem, dev, kpack
K, H, B, L, n, D = 9, 8, 5.0, 1, 1000, 7
kw = dict(lr=0.02, max_iters=400, average_window=50, loss_delta_tol=float(os.environ.get('TOL','0.02')), early_stop=True)
out = []
for i in range(36):
    blob, x = make_problem(n, D, K, H, L, seed=3000 + i, spread=0.6 + 0.3 * (i % 4))
    tb = nh.TrainBatch([dev(x)], [kpack(blob, D, K, H, L)], K, H, B, L, **kw)
    it = tb.run(use_graph=True)
    l = tb.iter_loss[0].cpu().numpy()
    out.append((int(it[0]), float(l[49]), float(l[99]), float(l[149])))
import collections
print(os.environ.get("NFISAM_HALF"), os.environ.get("TOL"), sorted(collections.Counter(o[0] for o in out).items()))
