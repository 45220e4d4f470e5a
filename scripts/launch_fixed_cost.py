"""Duration of ONE chunk-persistent launch of the C3 batch (or a Plaza clique: argument `plaza`) against its length:
kernel time = a + b x iterations -- `a` is what a short plan (the driver's 20-step line) pays once.
    python scripts/launch_fixed_cost.py [c3|plaza]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as BM
dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "c3"
prob, L = (BM.c3_problem(0), 1) if which == "c3" else BM.regime_problem("plaza_clique_n2000_D15", 7)
wl = BM.Workload(prob, L, dev)
rows = []
for iters in (2, 3, 5, 10, 20, 40, 80, 125):
    ms = wl.time_persistent_kernel(iters)
    rows.append((iters, ms * 1e3))
    print("%4d iterations in one launch: %8.1f us  (%.2f us per iteration)" % (iters, ms * 1e3, ms * 1e3 / iters))
x = np.array([r[0] for r in rows], float); y = np.array([r[1] for r in rows])
b, a = np.polyfit(x[3:], y[3:], 1)
print("fit over >= 10 iterations: %.1f us + %.2f us per iteration" % (a, b))
