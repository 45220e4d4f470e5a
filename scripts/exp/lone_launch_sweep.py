"""Diagnostic (GPU): one Plaza-shaped clique (n = 2000, D = 15) as plans of K iterations: duration of the ONE chunk-persistent
launch of a replay (two HIP events around it) and of the whole replay -> the launch's fixed cost and the replay's.  argv: [n D]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench as BM
n, D = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2000, 15)
rng = np.random.RandomState(0)
problem = [(rng.randn(n, D).astype(np.float32), BM.init_blob_np(D, BM.K, BM.H, 1, 0))]
rows = []
for K in (5, 10, 25, 50, 100, 125):
    w = BM.Workload(problem, 1, torch.device("cuda:0"))
    r, dt = w.record(K, 5, torch.cuda.synchronize)
    rows.append((K, r["training_launch_us"], 1e6 * dt))
    print("K=%3d  persistent launch %.1f us  whole replay %.1f us  (%s iterations per launch, persistent %s)" %
          (K, r["training_launch_us"], 1e6 * dt, r["iterations_per_launch"], r["chunk_persistent"]), flush=True)
K = np.array([r[0] for r in rows], dtype=float)
for name, col in (("persistent launch", 1), ("whole replay", 2)):
    y = np.array([r[col] for r in rows])
    a, b = np.polyfit(K, y, 1)
    print("%s = %.1f us + %.2f us x K" % (name, b, a))
