import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for q in ("tests", "nf-isam_amd", ""): sys.path.insert(0, os.path.join(ROOT, q))
import numpy as np, torch
import test_hip_parity as T
nh, CO = T.nh, T.CO
K, B, H = 9, 5.0, 16
for n, D in ((2000, 15), (300, 6), (2000, 6)):
    blob, x = T.make_problem(n, D, K, H, 1, seed=5, spread=1.0)
    bc, lc, _, _, _ = CO.train(x, blob, K, H, B, 1, lr=0.01, max_iters=12, early_stop=False, dtype=np.float64)
    for mode in ("0", None):
        for graph in (True, False):
            with T._Env(NFISAM_DIM_MAJOR=mode):
                tb = nh.TrainBatch([T.dev(x)], [T.kpack(blob, D, K, H)], K, H, B, 1, lr=0.01, max_iters=12, average_window=12, loss_delta_tol=0.0, early_stop=True)
                tb.run(use_graph=graph); torch.cuda.synchronize()
                il = tb.iter_loss[0].cpu().numpy()[:12]
                print(n, D, "DIM_MAJOR=%s graph=%s" % (mode, graph), "max |loss - oracle| %.2e" % np.abs(il - lc).max(), np.round(il[[0, 1, 2, 5, 11]], 4), np.round(lc[[0, 1, 2, 5, 11]], 4))
                tb.close()
