import os, sys
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/nf-isam_amd")
import numpy as np, torch
import test_configs_gpu as T
nh, BM = T.nh, T.BM
n, iters, lr = 2000, 500, 0.01
probs = [T.clique_problem(sh, n, 100 + c) for c, sh in enumerate(BM.C3_SHAPES)]
xs = [T.dev(x) for x, _, _ in probs]
kps = [nh.pack(T.dev(b), D, T.K, T.H, 1) for _, b, D in probs]
tb = nh.TrainBatch(xs, [k.clone() for k in kps], T.K, T.H, T.B, 1, lr=lr, max_iters=iters, early_stop=False)
tb.run(use_graph=True)
for c in (5, 7):
    il = tb.iter_loss[c].cpu().numpy()
    print(c, np.round(np.diff(il.reshape(10, 50).mean(1)), 3))
