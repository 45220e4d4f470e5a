"""Diagnostic (GPU): one Plaza-shaped clique, 400 iterations as windows of `w` iterations, early stop armed with tolerance 0 (never fires):
wall clock per run for NFISAM_SPAN=0 / 1 (set by the caller) -> what a window's end costs.  argv: [n D]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "nf-isam_amd")); sys.path.insert(0, ROOT)
import nfisam_hip as nh
import bench as BM
n, D = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2000, 15)
dev = torch.device("cuda:0")
rng = np.random.RandomState(0)
x = torch.from_numpy(rng.randn(n, D).astype(np.float32)).to(dev)
kp0 = nh.pack(torch.from_numpy(BM.init_blob_np(D, BM.K, BM.H, 1, 0)).to(dev), D, BM.K, BM.H, 1)
for w in (25, 50, 100):
    tb = nh.TrainBatch([x], [kp0.clone()], BM.K, BM.H, BM.B, 1, lr=0.01, max_iters=400, average_window=w, loss_delta_tol=0.0, early_stop=True)
    ts = []
    for rep in range(7):
        tb.reset([kp0.clone()]) if hasattr(tb, "reset") and rep else None
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        it = tb.run(use_graph=True)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    print("SPAN=%s window %3d: %d iterations, %.1f us per run (median of the last 5), final loss %.4f" %
          (os.environ.get("NFISAM_SPAN", "default"), w, it[0], 1e6 * float(np.median(ts[2:])), float(tb.iter_loss[0][it[0] - 1])))
    tb.close()
