"""Diagnostic (GPU): the two dim-major kernel families against the float64 C oracle -- one-shot gradient and training trajectory.
Question (round 6): Plaza1 ran 11 % more iterations with the two-lanes-per-particle family; is one family further from float64?"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "nf-isam_amd")); sys.path.insert(0, ROOT)
import nfisam_hip as nh
import bench as BM
from oracle import c_oracle as CO
K, H, B = BM.K, BM.H, BM.B
DEV = torch.device("cuda:0")
def dev(a): return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32).to(DEV)
ITERS = int(os.environ.get("ITERS", "400"))
for (n, D, seed) in ((500, 11, 0), (2000, 15, 0), (1000, 15, 1)):
    rng = np.random.RandomState(seed)
    x = rng.randn(n, D).astype(np.float32)
    blob = BM.init_blob_np(D, K, H, 1, seed)
    b64, l64, _, _, _ = CO.train(x, blob, K, H, B, 1, lr=BM.LR, max_iters=ITERS, early_stop=False, dtype=np.float64)
    b32, l32, _, _, _ = CO.train(x, blob, K, H, B, 1, lr=BM.LR, max_iters=ITERS, early_stop=False, dtype=np.float32)
    res = {}
    for fam, env in (("64", "0"), ("half", None)):
        if env is None: os.environ.pop("NFISAM_HALF", None)
        else: os.environ["NFISAM_HALF"] = env
        tb = nh.TrainBatch([dev(x)], [nh.pack(dev(blob), D, K, H, 1)], K, H, B, 1, lr=BM.LR, max_iters=ITERS, average_window=50,
                           loss_delta_tol=0.0, early_stop=True)
        ran = tb.run()
        il = tb.iter_loss[0].cpu().numpy()[:ITERS].copy()
        par = nh.unpack(tb.kparams[0], D, K, H).cpu().numpy().copy()
        tb.close()
        res[fam] = (il, par, ran)
    print("n=%d D=%d: iterations run %s / %s" % (n, D, res["64"][2], res["half"][2]))
    for it in (0, 1, 4, 9, 19, 49, 99, 199, ITERS - 1):
        if it < ITERS:
            print("  iter %3d: float64 %.6f | float32 oracle %+.2e | 64-particle %+.2e | two-lane %+.2e" %
                  (it, l64[it], l32[it] - l64[it], res["64"][0][it] - l64[it], res["half"][0][it] - l64[it]))
    for fam in ("64", "half"):
        e = np.abs(res[fam][1] - b64)
        print("  final parameters vs float64, %s: q50 %.2e q99 %.2e max %.2e   (float32 oracle: q50 %.2e q99 %.2e max %.2e)" %
              (fam, np.quantile(e, .5), np.quantile(e, .99), e.max(), np.quantile(np.abs(b32 - b64), .5), np.quantile(np.abs(b32 - b64), .99), np.abs(b32 - b64).max()))
