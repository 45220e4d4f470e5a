"""One-off (GPU): parallel graph branches vs chunk length.  argv: workload (c3|b64|plaza)
Times 1000 iterations of the plan at chunk lengths 20 / 50 / 125 with NFISAM_CHAINS = 1 and 2 (median of 5 replays)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "nf-isam_amd")); sys.path.insert(0, ROOT)
import nfisam_hip as nh
import bench as BM
dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "c3"
prob, L = (BM.c3_problem(0), 1) if name == "c3" else BM.regime_problem({"b64": "batch64_n2000_D15", "plaza": "plaza_clique_n2000_D15"}[name], 0)
w = BM.Workload(prob, L, dev)
for window, iters in ((20, 20), (20, 1000), (50, 1000), (125, 1000)):
    for chains in ("1", "2"):
        os.environ["NFISAM_CHAINS"] = chains
        tb = nh.TrainBatch(w.xs, [p.clone() for p in w.kp0], BM.K, BM.H, BM.B, L, lr=BM.LR, max_iters=iters, average_window=window,
                           loss_delta_tol=0.0, early_stop=True)
        tb.prepare(True)
        ts = []
        for r in range(7):
            tb.reset(w.kp0)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            tb.run(True)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        tb.close()
        print("%s chunk %3d x %4d iterations, chains %s: %.2f us per iteration" % (name, window, iters, chains, 1e6 * np.median(ts) / iters), flush=True)
