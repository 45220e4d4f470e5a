"""Diagnostic (GPU): per-iteration and gradient-kernel times of a synthetic batch through bench.Workload.
argv: n_cliques n D [L]   env: NFISAM_TRAIN / NFISAM_COND / NFISAM_DIM_MAJOR ... select the kernel family."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as BM
nc, n, D = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
L = int(sys.argv[4]) if len(sys.argv) > 4 else 1
rng = np.random.RandomState(0)
Hh = int(os.environ.get("TG_H", BM.H))                  # hidden_dim (TG_H=16 | 4)
if "TG_K" in os.environ:                               # num_knots (TG_K=5 | 12 ...; bench.py's own is 9)
    BM.K = int(os.environ["TG_K"])
problem = [(rng.randn(n, D).astype(np.float32), BM.init_blob_np(D, BM.K, Hh, L, c)) for c in range(nc)]
w = BM.Workload(problem, L, torch.device("cuda:0"), hidden=Hh)
r, _ = w.record(400, 50, torch.cuda.synchronize)
print("H=%d " % Hh, end="")
print("%d x (n=%d, D=%d, L=%d): %.2f us/iteration, gradient kernel %.2f us, loss %.3f -> %.3f" %
      (nc, n, D, L, r["us_per_iteration"], r["gradient_kernel_us"], r["first_loss"], r["final_loss"]))
