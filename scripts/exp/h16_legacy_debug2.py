import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "nf-isam_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
import bench as BM
import nfisam_hip as nh
dev = torch.device("cuda:0")
for name in ("plaza_clique_n2000_D15_H16", "batch64_n2000_D15_H16"):
    prob, L = BM.regime_problem(name, seed0=7)
    for mode in ("0", None):
        if mode is None: os.environ.pop("NFISAM_DIM_MAJOR", None)
        else: os.environ["NFISAM_DIM_MAJOR"] = mode
        w = BM.Workload(prob, L, dev, 16)
        tb = w.batch(200)
        tb.prepare(use_graph=True)
        out = []
        for r in range(3):
            if r > 0: tb.reset(w.kp0)
            done = tb.run(use_graph=True); torch.cuda.synchronize()
            il = tb.iter_loss[0].cpu().numpy()
            out.append((done[0], round(float(il[0]), 3), round(float(il[99]), 3), round(float(il[100]), 3), round(float(il[199]), 3)))
        print(name, "DIM_MAJOR=%s" % mode, out)
        tb.close()
