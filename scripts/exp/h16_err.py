import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for q in ("tests", "nf-isam_amd", ""): sys.path.insert(0, os.path.join(ROOT, q))
import numpy as np, torch
import test_hip_parity as T
nh, CO = T.nh, T.CO
K, B = 9, 5.0
for H in (8, 16):
    for n, D in ((1000, 24), (2000, 19)):
        blob, x = T.make_problem(n, D, K, H, 1, seed=77 + D)
        lossc, gradc, _, _ = CO.nll_grad(x, blob, K, H, B, 1, dtype=np.float64, want_gx=True)
        out = {}
        for mode in ("0", None):
            with T._Env(NFISAM_DIM_MAJOR=mode, NFISAM_TRAIN="wide"):
                tb = nh.TrainBatch([T.dev(x)], [T.kpack(blob, D, K, H)], K, H, B, 1, lr=0.01, max_iters=3, early_stop=False)
                tb.step(); torch.cuda.synchronize()
                m = nh.unpack(tb.m[0], D, K, H).cpu().numpy() * 10.0        # gradient (mean over n)
                out[mode] = m
        sc = np.abs(gradc).max()
        e0, e1 = np.abs(out["0"] - gradc), np.abs(out[None] - gradc)
        j = int(np.argmax(e1))
        print("H=%d n=%d D=%d: |grad|max %.3f  wide-vs-f64 max %.2e  dim-major-vs-f64 max %.2e (at ref-order index %d of %d, value %.4f)  q99 %.2e / %.2e" %
              (H, n, D, sc, e0.max(), e1.max(), j, len(gradc), gradc[j], np.quantile(e0, 0.99), np.quantile(e1, 0.99)))

for H in (8, 16):
    n, D = 1000, 24
    print('H =', H)
    # where is the H = 16 error?  kernel layout: dim 0 = PoP spline parameters, dim i: W0 [i][H] | b0 [H] | W1 [H][H] | b1 [H] | W2 [H][PoP] | b2 [PoP]
    blob, x = T.make_problem(n, D, K, H, 1, seed=77 + D)
    lossc, gradc, _, _ = CO.nll_grad(x, blob, K, H, B, 1, dtype=np.float64, want_gx=True)
    gk = nh.pack(torch.from_numpy(gradc.astype(np.float32)).to(T.DEV), D, K, H, 1).cpu().numpy()
    with T._Env(NFISAM_DIM_MAJOR=None, NFISAM_TRAIN="wide"):
        tb = nh.TrainBatch([T.dev(x)], [T.kpack(blob, D, K, H)], K, H, B, 1, lr=0.01, max_iters=3, early_stop=False)
        tb.step(); torch.cuda.synchronize()
        mk = tb.m[0].cpu().numpy() * 10.0
    PoP = 32
    off = PoP
    for i in range(1, D):
        secs = [("W0", i * H), ("b0", H), ("W1", H * H), ("b1", H), ("W2", H * PoP), ("b2", PoP)]
        line = []
        for name, cnt in secs:
            e = np.abs(mk[off:off + cnt] - gk[off:off + cnt]).max()
            line.append("%s %.1e" % (name, e))
            off += cnt
        if i in (1, 5, 15, 16, 17, 20, 23):
            print("dim %2d: " % i + "  ".join(line))
