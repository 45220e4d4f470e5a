"""Experiment (GPU): hidden_dim 16 through the dim-major kernel (default) and through the tile-major wide kernel with the
butterfly gradient reduction (`NFISAM_DIM_MAJOR=0`, what H = 16 ran before)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "nf-isam_amd")); sys.path.insert(0, ROOT)
import torch
import bench as BM
dev = torch.device("cuda:0")
for name in ("plaza_clique_n2000_D15_H16", "batch64_n2000_D15_H16"):
    prob, L = BM.regime_problem(name, seed0=7)
    for mode in ("0", None):
        if mode is None:
            os.environ.pop("NFISAM_DIM_MAJOR", None)
        else:
            os.environ["NFISAM_DIM_MAJOR"] = mode
        r, _ = BM.Workload(prob, L, dev, 16).record(200, 20, torch.cuda.synchronize)
        print(name, "NFISAM_DIM_MAJOR=%s" % mode, "%.2f us/iteration, kernel %.2f us, frac %.4f" % (r["us_per_iteration"], r["gradient_kernel_us"], r["frac_of_fp32_peak"]))
