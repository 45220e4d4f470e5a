"""Diagnostic (GPU, `make stamps` build): when do the blocks of ONE gradient launch start and end (s_memrealtime, 100 MHz)?
argv: c3 | <n_cliques> <n> <D>"""
import os, sys, ctypes as C
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "nf-isam_amd")); sys.path.insert(0, ROOT)
import nfisam_hip as nh
nh.LIB_PATH = os.path.join(os.path.dirname(nh.LIB_PATH), "libnfisam_hip_stamps.so")
import bench as BM
dev = torch.device("cuda:0")
K, H, B, L = 9, 8, 5.0, 1
if sys.argv[1] == "c3":
    prob = BM.c3_problem(seed0=100)
else:
    nc, n, D = [int(v) for v in sys.argv[1:4]]
    rng = np.random.RandomState(0)
    prob = [(rng.randn(n, D).astype(np.float32), BM.init_blob_np(D, K, H, L, c)) for c in range(nc)]
xs = [torch.from_numpy(x).to(dev) for x, _ in prob]
kps = [nh.pack(torch.from_numpy(b).to(dev), x.shape[1], K, H, L) for x, b in prob]
tb = nh.TrainBatch(xs, kps, K, H, B, L, lr=0.01, max_iters=100000, early_stop=False)
for _ in range(4):
    tb.gradient_only()
torch.cuda.synchronize()
blk = (C.c_ulonglong * (4096 * 2))()
assert nh.lib().nfisam_debug_read_blocks(blk) == 0
bt = np.array(blk[:], dtype=np.int64).reshape(4096, 2)
bt = bt[(bt[:, 0] > 0) & (bt[:, 1] > 0)]
t0 = bt[:, 0].min()
start = (bt[:, 0] - t0) / 100.0
end = (bt[:, 1] - t0) / 100.0
dur = end - start
q = lambda a: " ".join("%.2f" % v for v in np.quantile(a, [0, 0.1, 0.5, 0.9, 1.0]))
print("%d blocks stamped | start offset us (min q10 med q90 max): %s | duration us: %s | end us: %s" % (len(bt), q(start), q(dur), q(end)))
