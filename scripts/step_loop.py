"""Runs a plain training-step loop for one clique shape (for rocprofv3 --kernel-trace): argv n D L iters graph(0/1)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nf-isam_amd")); sys.path.insert(0, ROOT)
import nfisam_hip as nh
import bench as BM
n, D, L, iters, graph = [int(v) for v in sys.argv[1:6]]
K, H, B = 9, 8, 5.0
dev = torch.device("cuda:0")
rng = np.random.RandomState(0)
x = torch.from_numpy(rng.randn(n, D).astype(np.float32)).to(dev)
kp = nh.pack(torch.from_numpy(BM.init_blob_np(D, K, H, L, 0)).to(dev), D, K, H, L)
tb = nh.TrainBatch([x], [kp], K, H, B, L, lr=0.01, max_iters=iters, early_stop=False)
import time
torch.cuda.synchronize()
t0 = time.perf_counter()
if graph:
    tb.run(use_graph=True)
else:
    for _ in range(iters):
        tb.step()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("done", tb.state(), "%.2f us/iteration (host wall clock incl. graph build)" % (dt / iters * 1e6))
if graph:
    tb.reset() if hasattr(tb, "reset") else None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tb.run(use_graph=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("second run (graph cached): %.2f us/iteration" % (dt / iters * 1e6))
