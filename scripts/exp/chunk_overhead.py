"""Diagnostic (GPU): cost of closing a chunk (Adam close, bookkeeping kernel, host read-back, graph relaunch): the same 600
iterations of one Plaza-shaped clique with chunks of 50 (the reference's window) and of 100 iterations."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "nf-isam_amd")); sys.path.insert(0, ROOT)
import nfisam_hip as nh
import bench as BM
dev = torch.device("cuda:0")
n, D, K, H, B = 2000, 15, 9, 8, 5.0
rng = np.random.RandomState(0)
x = torch.from_numpy(rng.randn(n, D).astype(np.float32)).to(dev)
for wnd in (50, 100, 25):
    kp = nh.pack(torch.from_numpy(BM.init_blob_np(D, K, H, 1, 0)).to(dev), D, K, H, 1)
    tb = nh.TrainBatch([x], [kp], K, H, B, 1, lr=0.01, max_iters=600, average_window=wnd, loss_delta_tol=0.0, early_stop=True)
    tb.prepare(use_graph=True)
    best = 1e9
    for rep in range(4):
        tb.reset(kparams=[kp])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tb.run(use_graph=True)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    print("window %3d: %.2f us per iteration (600 iterations, %d chunks)" % (wnd, 1e6 * best / 600, 600 // wnd))
