"""Diagnostic (GPU): random batch shapes through the fused-Adam launches and through gradient kernel + Adam kernel; parameters and
moments must be equal bit for bit, and the first iteration's gradient must match the scalar-conditioner kernel."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "nf-isam_amd")); sys.path.insert(0, ROOT)
import nfisam_hip as nh
import bench as BM
dev = torch.device("cuda:0")
rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
K, H, B = 9, int(sys.argv[3]) if len(sys.argv) > 3 else 8, 5.0
bad = 0
for case in range(int(sys.argv[2]) if len(sys.argv) > 2 else 30):
    nc = int(rng.choice([1, 1, 2, 3, 5, 9]))
    shapes = [(int(rng.choice([1, 63, 64, 65, 200, 512, 513, 1000, 2000, 2048])), int(rng.randint(1, 26))) for _ in range(nc)]
    iters = int(rng.choice([3, 7, 12])); wnd = int(rng.choice([1, 2, 3, 4, 5]))
    out = {}
    for mode in ("0", "1"):
        os.environ["NFISAM_FUSED_ADAM"] = mode
        r2 = np.random.RandomState(1000 + case)
        xs = [torch.from_numpy(r2.randn(n, D).astype(np.float32)).to(dev) for n, D in shapes]
        kps = [nh.pack(torch.from_numpy(BM.init_blob_np(D, K, H, 1, c)).to(dev), D, K, H, 1) for c, (n, D) in enumerate(shapes)]
        tb = nh.TrainBatch(xs, kps, K, H, B, 1, lr=0.01, max_iters=iters, average_window=wnd, loss_delta_tol=0.0, early_stop=True)
        tb.run(use_graph=bool(case & 1))
        torch.cuda.synchronize()
        out[mode] = [[t.cpu().numpy().copy() for t in arr] for arr in (tb.kparams, tb.m, tb.v)]
        tb.close()
    ok = all(np.array_equal(a, b) for A, Bb in zip(out["0"], out["1"]) for a, b in zip(A, Bb))
    fin = all(np.all(np.isfinite(a)) for a in out["1"][0])
    if not (ok and fin):
        bad += 1
    print("case %2d: %d cliques %s iters %d window %d -> %s%s" % (case, nc, shapes, iters, wnd, "equal" if ok else "DIFFERENT", "" if fin else " NON-FINITE"))
print("mismatches:", bad)
