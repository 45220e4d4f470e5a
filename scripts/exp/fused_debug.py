import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "nf-isam_amd")); sys.path.insert(0, ROOT)
import nfisam_hip as nh
import bench as BM
dev = torch.device("cuda:0")
n, D = int(sys.argv[1]), int(sys.argv[2])
K, H, B, L = 9, 8, 5.0, 1
os.environ["NFISAM_TRAIN"] = "wide"; os.environ["NFISAM_DIM_MAJOR_MIN"] = "0"
def run(fused, iters, skip_close=False):
    os.environ["NFISAM_FUSED_ADAM"] = "1" if fused else "0"
    if skip_close: os.environ["NFISAM_DEBUG_SKIP_CLOSE"] = "1"
    else: os.environ.pop("NFISAM_DEBUG_SKIP_CLOSE", None)
    rng = np.random.RandomState(0)
    x = torch.from_numpy(rng.randn(n, D).astype(np.float32)).to(dev)
    kp = nh.pack(torch.from_numpy(BM.init_blob_np(D, K, H, L, 0)).to(dev), D, K, H, L)
    tb = nh.TrainBatch([x], [kp], K, H, B, L, lr=0.01, max_iters=iters, average_window=iters, loss_delta_tol=0.0, early_stop=True)
    tb.run(use_graph=False)
    torch.cuda.synchronize()
    return [a[0].cpu().numpy().copy() for a in (tb.kparams, tb.m, tb.v, tb.g)]
P = nh.kparam_count(D, K, H)
copies = (n + 63) // 64
ring = 128 * 64 + 64
def cmp(name, a, b):
    d = np.nonzero(a != b)[0]
    print("%-28s differing %6d of %d, max abs %.3g" % (name, len(d), len(a), np.abs(a - b).max() if len(a) else 0), d[:6])
u1 = run(False, 1); u2 = run(False, 2)
f2 = run(True, 2, skip_close=True)
g = f2[3]
set0, set1 = g[:copies * P], g[copies * P + ring: 2 * copies * P + ring]
alt = g[2 * copies * P + ring: 2 * copies * P + ring + 3 * P]
cmp("alt theta_1 vs unfused", alt[:P], u1[0]); cmp("alt m_1", alt[P:2 * P], u1[1]); cmp("alt v_1", alt[2 * P:], u1[2])
cmp("grad_1 copies (set1 vs set0)", set1, u2[3][:copies * P])
cmp("grad_0 copies", set0, u1[3][:copies * P])
# IEEE emulation of adam_update for theta_1
rng = np.random.RandomState(0)
_ = rng.randn(n, D)
th0 = nh.pack(torch.from_numpy(BM.init_blob_np(D, K, H, L, 0)).to(dev), D, K, H, L).cpu().numpy()
f = np.float32
b1, b2, lr, eps = f(0.9), f(0.999), f(0.01), f(1e-8)
bc1 = f(1.0 - float(b1) ** 1); bc2 = f(1.0 - float(b2) ** 1)
step = f(lr / bc1); inv = f(f(1.0) / np.sqrt(bc2, dtype=f))
m1, v1 = u1[1], u1[2]
denom = (np.sqrt(v1, dtype=f) * inv).astype(f) + eps
upd = ((step * m1).astype(f) / denom).astype(f)
th1 = (th0 - upd).astype(f)
cmp("numpy vs unfused theta_1", th1, u1[0]); cmp("numpy vs fused alt theta_1", th1, alt[:P])
print("coefs: bc1 %r bc2 %r step %r inv %r" % (bc1, bc2, step, inv))
