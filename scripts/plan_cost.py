"""Host cost of creating a training plan for a clique shape the process has not seen (graph capture + instantiation), and of
re-running a cached one.   python scripts/plan_cost.py"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nf-isam_amd"))
import nfisam_hip as nh
dev = torch.device("cuda:0")
K, H, B = 9, 8, 5.0
def make(n, D, iters=2000, window=50):
    x = torch.randn(n, D, device=dev)
    kp = (0.1 * torch.randn(nh.kparam_count(D, K, H), device=dev)).contiguous()
    return nh.TrainBatch([x], [kp], K, H, B, 1, lr=0.01, max_iters=iters, average_window=window, loss_delta_tol=0.01, early_stop=True)
tb = make(2000, 6); tb.prepare(use_graph=True); tb.run(); torch.cuda.synchronize(); tb.close()      # pages everything in
for n, D in ((2000, 7), (2000, 11), (2000, 12), (2000, 15), (600, 9), (2000, 16)):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tb = make(n, D)
    t1 = time.perf_counter()
    tb.prepare(use_graph=True)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    done = tb.run()
    torch.cuda.synchronize(); t3 = time.perf_counter()
    print("n=%d D=%2d: TrainBatch() %.2f ms, prepare (capture + instantiate) %.2f ms, run %d iterations %.2f ms" %
          (n, D, (t1 - t0) * 1e3, (t2 - t1) * 1e3, done[0], (t3 - t2) * 1e3))
    tb.close()
