import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for q in ("tests", "nf-isam_amd", ""): sys.path.insert(0, os.path.join(ROOT, q))
import numpy as np, torch
import test_hip_parity as T
nh, CO = T.nh, T.CO
K, B, H = 9, 5.0, 8
for n, D in ((2000, 19), (2000, 17), (1000, 19), (2000, 15), (2000, 24), (4000, 19), (600, 19)):
    blob, x = T.make_problem(n, D, K, H, 1, seed=77 + D)
    lossc, gradc, _, _ = CO.nll_grad(x, blob, K, H, B, 1, dtype=np.float64, want_gx=True)
    for env in (dict(NFISAM_DIM_MAJOR="0", NFISAM_TRAIN="wide"), dict(NFISAM_DIM_MAJOR="0", NFISAM_TRAIN="wide", NFISAM_WEIGHTS="scalar"),
                dict(NFISAM_DIM_MAJOR="0", NFISAM_TRAIN="wide", NFISAM_GRAD="butterfly")):
        with T._Env(**env):
            kg, _, loss = nh.backward(T.dev(x), T.kpack(blob, D, K, H), K, H, B, 1, nll_mode=True)
            g = nh.unpack(kg, D, K, H, 1).cpu().numpy() / n
            tb = nh.TrainBatch([T.dev(x)], [T.kpack(blob, D, K, H)], K, H, B, 1, lr=0.01, max_iters=3, early_stop=False)
            tb.step(); torch.cuda.synchronize()
            m = nh.unpack(tb.m[0], D, K, H).cpu().numpy() * 10.0
        print(n, D, {k: v for k, v in env.items() if k not in ("NFISAM_DIM_MAJOR", "NFISAM_TRAIN")},
              "backward() max err %.2e | TrainBatch.step max err %.2e q99 %.2e" % (np.abs(g - gradc).max(), np.abs(m - gradc).max(), np.quantile(np.abs(m - gradc), 0.99)))
