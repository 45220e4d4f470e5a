"""Diagnostic (GPU): per-phase cycle stamps of one train-kernel block (needs `make -C nf-isam_amd/csrc stamps`)."""
import os, sys, ctypes as C
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nf-isam_amd")); sys.path.insert(0, ROOT)
import nfisam_hip as nh
nh.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "nf-isam_amd", "csrc", "_diag", "libnfisam_hip_stamps.so")
if not os.path.exists(nh.LIB_PATH):      # built on demand, on this box (not shipped)
    import subprocess
    subprocess.check_call(["make", "-C", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "nf-isam_amd", "csrc"), "stamps"])
import bench as BM
dev = torch.device("cuda:0")
n, D, L, K, H, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), 9, 8, 5.0
rng = np.random.RandomState(0)
x = torch.from_numpy(rng.randn(n, D).astype(np.float32)).to(dev)
kp = nh.pack(torch.from_numpy(BM.init_blob_np(D, K, H, L, 0)).to(dev), D, K, H, L)
tb = nh.TrainBatch([x], [kp], K, H, B, L, lr=0.01, max_iters=100000, early_stop=False)   # per-tile slab path, as in training
for _ in range(3):
    tb.gradient_only()
torch.cuda.synchronize()
init = (C.c_ulonglong * (64 * 32))()
init[63 * 32 + 20] = 2 ** 63
import ctypes
lib = nh.lib()
sym = ctypes.c_void_p.in_dll(lib, "g_stamps") if False else None
# reset min/max slots through a tiny torch kernel is not possible for a __device__ symbol: run once more after zeroing via hipMemcpyToSymbol
lib.nfisam_debug_write_stamps(init)
tb.gradient_only()
torch.cuda.synchronize()
buf = (C.c_ulonglong * (64 * 32))()
assert nh.lib().nfisam_debug_read_stamps(buf) == 0
st = np.array(buf[:]).reshape(64, 32)
names = ["start", "x+w loaded", "unit start", "theta(MLP fwd)", "spline fwd", "spline bwd", "MLP bwd (VALU)", "phase A (W2 mfma+flush)",
         "phase B (W1,W0 mfma+flush)", "end of layers"]
for w in range(D):
    t = st[w]
    print("wave %d (dim %d):" % (w, w), " ".join("%s=%d" % (names[i].split()[0], t[i] - t[0]) for i in range(10) if t[i] > 0))
    if w > 0:
        seg = [(names[i], int(t[i] - t[i - 1])) for i in range(3, 9)]
        print("    last-unit segments (cycles):", seg, " total kernel:", int(t[9] - t[0]))


acc_names = {1: "prologue", 2: "loop/zero+sync(bwd)", 3: "theta (MLP fwd)", 4: "spline fwd*", 5: "spline fwd+bwd", 6: "MLP bwd", 7: "phase A",
             8: "phase B", 9: "tail", 10: "fwd-only units", 11: "barrier wait (fwd)", 12: "after last unit (bwd)", 13: "barrier wait (bwd)"}
print("accumulated cycles per phase over the whole kernel (block 0):")
for w in range(min(D, 8) if L > 1 else D):
    a = st[w][16:32]
    tot = int(st[w][9] - st[w][0]) if st[w][9] > 0 else 0
    print("  wave %d total %6d: " % (w, tot) + ", ".join("%s=%d" % (acc_names[i], int(a[i])) for i in sorted(acc_names) if a[i] > 0))
blk = (C.c_ulonglong * (4096 * 2))()
assert nh.lib().nfisam_debug_read_blocks(blk) == 0
bt = np.array(blk[:], dtype=np.int64).reshape(4096, 2)
nb = ((n + 31) // 32) * (D if L == 1 else 1)
bt = bt[:nb]
t0 = bt[:, 0].min()
start = (bt[:, 0] - t0) / 100.0
dur = (bt[:, 1] - bt[:, 0]) / 100.0
end = (bt[:, 1] - t0) / 100.0
print("blocks %d: start offset us min/med/max %.2f %.2f %.2f | duration us min/med/max %.2f %.2f %.2f | last end %.2f" % (
    nb, start.min(), np.median(start), start.max(), dur.min(), np.median(dur), dur.max(), end.max()))
order = np.argsort(end)[-5:]
print("  slowest-ending blocks (id, start, dur):", [(int(i), round(float(start[i]), 2), round(float(dur[i]), 2)) for i in order])
