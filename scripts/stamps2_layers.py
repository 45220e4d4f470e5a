"""Diagnostic (GPU): raw cycle stamps of the multi-layer training kernel (nsf_train3_kernel), light build
(`make -C nf-isam_amd/csrc stamps` -> libnfisam_hip_stamps2.so): start, prologue end, every layer barrier, end."""
import os, sys, ctypes as C
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nf-isam_amd")); sys.path.insert(0, ROOT)
import nfisam_hip as nh
nh.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "nf-isam_amd", "csrc", "_diag", "libnfisam_hip_stamps2.so")
if not os.path.exists(nh.LIB_PATH):      # built on demand, on this box (not shipped)
    import subprocess
    subprocess.check_call(["make", "-C", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "nf-isam_amd", "csrc"), "stamps"])
import bench as BM
dev = torch.device("cuda:0")
n, D, L, K, H, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), 9, 8, 5.0
rng = np.random.RandomState(0)
x = torch.from_numpy(rng.randn(n, D).astype(np.float32)).to(dev)
kp = nh.pack(torch.from_numpy(BM.init_blob_np(D, K, H, L, 0)).to(dev), D, K, H, L)
if len(sys.argv) > 4 and sys.argv[4] == "plan":       # the last launch of a 5-iteration plan: panels copied from the panel image
    tb = nh.TrainBatch([x], [kp], K, H, B, L, lr=0.01, max_iters=5, early_stop=False)
    tb.run(use_graph=True)
else:
    tb = nh.TrainBatch([x], [kp], K, H, B, L, lr=0.01, max_iters=100000, early_stop=False)
    for _ in range(3):
        tb.gradient_only()
torch.cuda.synchronize()
buf = (C.c_ulonglong * (64 * 32))()
assert nh.lib().nfisam_debug_read_stamps(buf) == 0
st = np.array(buf[:], dtype=np.int64).reshape(64, 32)
for w in range((D + 1) // 2):
    t = st[w]
    ev = [("args+issue", t[14]), ("panels", t[3]), ("state", t[15]), ("x tile -> prologue end", t[1])] + [("fwd L%d" % l, t[16 + l]) for l in range(L - 1)] + [("bwd L%d" % l, t[24 + l]) for l in range(L - 1, -1, -1)] + [("end", t[9])]
    prev, out = t[0], []
    for name, v in ev:
        out.append("%s=%d" % (name, v - prev)); prev = v
    print("wave %d: total %d | " % (w, t[9] - t[0]) + " ".join(out))
