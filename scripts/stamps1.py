"""Diagnostic (GPU): per-phase cycle stamps of nsf_train1_kernel waves (needs `make -C nf-isam_amd/csrc stamps`).
argv: n_cliques n D   -- block (0, 0, dim) of the launch is stamped: wave w of dim i."""
import os, sys, ctypes as C
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nf-isam_amd")); sys.path.insert(0, ROOT)
import nfisam_hip as nh
nh.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "nf-isam_amd", "csrc", "_diag", os.environ.get("STAMPS_LIB", "libnfisam_hip_stamps.so"))
if not os.path.exists(nh.LIB_PATH):      # built on demand, on this box (not shipped)
    import subprocess
    subprocess.check_call(["make", "-C", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "nf-isam_amd", "csrc"), "stamps"])
import bench as BM
dev = torch.device("cuda:0")
nc, n, D = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
K, H, B, L = 9, 8, 5.0, 1
rng = np.random.RandomState(0)
xs = [torch.from_numpy(rng.randn(n, D).astype(np.float32)).to(dev) for _ in range(nc)]
kps = [nh.pack(torch.from_numpy(BM.init_blob_np(D, K, H, L, c)).to(dev), D, K, H, L) for c in range(nc)]
lib = nh.lib()
if os.environ.get("STAMPS_TRAIN"):     # a whole chunk of training iterations: the stamps are those of its last gradient launch (fused Adam pending)
    tb = nh.TrainBatch(xs, kps, K, H, B, L, lr=0.01, max_iters=8, average_window=4, loss_delta_tol=0.0, early_stop=True)
    tb.run(use_graph=False)
else:
    tb = nh.TrainBatch(xs, kps, K, H, B, L, lr=0.01, max_iters=100000, early_stop=False)
    for _ in range(3):
        tb.gradient_only()
    torch.cuda.synchronize()
    lib.nfisam_debug_write_stamps((C.c_ulonglong * (64 * 32))())
    tb.gradient_only()
torch.cuda.synchronize()
buf = (C.c_ulonglong * (64 * 32))()
assert lib.nfisam_debug_read_stamps(buf) == 0
st = np.array(buf[:]).reshape(64, 32)
names = {1: "loop/top", 2: "load tile", 14: "cond hidden", 3: "cond theta", 4: "spline fwd", 5: "spline bwd", 6: "cond bwd", 7: "phase A", 8: "phase B", 9: "sink"}
print("accumulated cycles per phase (whole wave lifetime), waves of block x=0, clique 0:")
for w in range(min(64, 4 * D)):
    a = st[w][16:32]
    tot = int(st[w][9] - st[w][0]) if st[w][9] > 0 else 0
    if tot:
        print("  slot %2d (dim %2d wave %d) total %6d: " % (w, w // 4, w % 4, tot) + ", ".join("%s=%d" % (names[i], int(a[i])) for i in sorted(names) if a[i] > 0))

if os.environ.get("STAMPS_LIB", "").endswith("stamps2.so"):
    # light stamps (one tile per wave): raw times in program order
    order = [10, 0, 11, 1, 2, 14, 3, 4, 5, 6, 7, 8, 12, 9]
    label = {10: "entry", 0: "setup", 11: "fetch+panel+barrier", 1: "tile stored/top", 2: "tile in LDS", 14: "hidden", 3: "cond fwd",
             4: "spline fwd", 5: "spline bwd", 6: "cond bwd", 7: "phase A", 8: "phase B", 9: "loss", 12: "fragments + block sum + copy"}
    print("light stamps, cycles since kernel entry -> delta per phase:")
    for w in range(min(64, 4 * D)):
        t = {k: int(st[w][k]) for k in order if st[w][k] > 0}
        if 9 not in t or 10 not in t:
            continue
        ks = [k for k in order if k in t]
        parts = ["%s=%d" % (label[b], t[b] - t[a]) for a, b in zip(ks[:-1], ks[1:])]
        print("  slot %2d (dim %2d wave %d) total %6d: " % (w, w // 4, w % 4, t[9] - t[10]) + ", ".join(parts))
