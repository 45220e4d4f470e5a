"""Diagnostic (GPU): per-phase cycle SUMS of nsf_train1_kernel waves, phases pinned by data dependences
(needs `make -C nf-isam_amd/csrc stamps`: libnfisam_hip_stamps3.so).   argv: n_cliques n D [train]
Waves of tile group 0 of clique 0 are stamped; sums run over the wave's T tiles."""
import os, sys, ctypes as C
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nf-isam_amd")); sys.path.insert(0, ROOT)
import nfisam_hip as nh
nh.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "nf-isam_amd", "csrc", "_diag", "libnfisam_hip_stamps3.so")
_shipped = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "nf-isam_amd", "csrc", "_stamps3", "libnfisam_hip_stamps3.so")
if os.path.exists(_shipped):             # (round 6: a build made in the container and shipped with the snapshot)
    nh.LIB_PATH = _shipped
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
per = 1
if len(sys.argv) > 4 and sys.argv[4] == "persist":
    # the chunk-persistent form (when the plan takes it): 2 chunks of 50 iterations through the graph; the sums are those of
    # the LAST launch = 50 iterations ("prologue" = group barrier + staging of the next iteration)
    if nc == 0:                                             # argv: 0 n D persist -> the C3 cliques, widest first
        prob = BM.c3_problem(0)[::-1]
        xs = [torch.from_numpy(x).to(dev) for x, _ in prob]
        kps = [nh.pack(torch.from_numpy(b).to(dev), x.shape[1], K, H, L) for x, b in prob]
        nc, D = len(xs), int(xs[0].shape[1])
    tb = nh.TrainBatch(xs, kps, K, H, B, L, lr=0.01, max_iters=100, average_window=50, loss_delta_tol=0.0, early_stop=True)
    tb.run(use_graph=True)
    print("XCDs per group:", tb.xcd_span(), "(0: the plan did not take the persistent form)")
    per = 50 if tb.xcd_span() > 0 else 1
elif len(sys.argv) > 4:
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
st = np.array(buf[:]).reshape(64, 32)[:, 16:32].astype(np.int64)
names = ["", "prologue", "load tile", "cond fwd", "spline fwd", "spline bwd", "cond bwd", "h2 operand", "grad GEMMs", "epilogue",
         "P:loop top", "P:staging", "P:barrier", "P:looks(count)"]
rows = [w for w in range(min(64, 4 * D)) if st[w].sum() > 0 and w >= 4]
print("%d x (n=%d, D=%d): cycles per phase, summed over the wave's tiles (mean over %d stamped waves of dims >= 1)" % (nc, n, D, len(rows)))
m = st[rows].mean(0) / per
st = st // per
for i in range(1, 14 if per > 1 else 10):
    print("  %-14s %8.0f" % (names[i], m[i]))
print("  %-14s %8.0f" % ("total", m[1:10].sum() + (m[10:13].sum() if per > 1 else 0)))
if hasattr(lib, "nfisam_debug_read_stg"):
    sb = (C.c_ulonglong * 32)()
    if lib.nfisam_debug_read_stg(sb, 0) == 0:
        v = np.array(sb[:]).astype(np.int64)
        print("  staging sub-phases (thread 64 of block (1,0,0), mean cycles):",
              " ".join("%s=%d" % (nm, v[q] // max(1, v[16 + q])) for q, nm in enumerate(["issue+coef", "arrival", "adam", "stores"])), "calls", v[16:20].tolist())
for w in (4, 5, 4 * (D // 2), 4 * (D - 1)):
    if w < 64:
        print("  slot %2d (dim %2d wave %d): " % (w, w // 4, w % 4) + " ".join("%s=%d" % (names[i], st[w][i]) for i in range(1, 10)))
