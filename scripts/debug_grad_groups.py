"""Debug helper (GPU): compare HIP NLL gradients with the C oracle per parameter group."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nf-isam_amd")); sys.path.insert(0, ROOT)
import nfisam_hip as nh
from oracle import c_oracle as CO, nsf_torch as O

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
D = int(sys.argv[2]) if len(sys.argv) > 2 else 4
K = int(sys.argv[3]) if len(sys.argv) > 3 else 9
H, B, L = 8, 5.0, 1
gen = torch.Generator().manual_seed(1)
blob = O.init_blob(D, K, H, gen).numpy()
x = (1.3 * torch.randn(n, D, generator=gen)).numpy().astype(np.float32)
dev = torch.device("cuda:0")
kp = nh.pack(torch.from_numpy(blob).to(dev), D, K, H, L)
kg, _, loss = nh.backward(torch.from_numpy(x).to(dev), kp, K, H, B, L, nll_mode=True)
g = nh.unpack(kg, D, K, H, L).cpu().numpy() / n
lo, go, _, _ = CO.nll_grad(x, blob, K, H, B, L, dtype=np.float64)
Po = 3 * K - 1
off = 0
def rep(name, cnt, shape=None):
    global off
    a, b = g[off:off + cnt], go[off:off + cnt]
    err = np.abs(a - b).max()
    flag = "" if err < 1e-4 * max(1, np.abs(b).max()) else "   <<<<<<"
    if flag or len(sys.argv) <= 4:
        print("%-14s max|ref| %.3e  max err %.3e%s" % (name, np.abs(b).max(), err, flag))
    if flag and shape is not None and cnt <= 300:
        with np.printoptions(precision=4, suppress=True, linewidth=200):
            print("  got", a.reshape(shape)); print("  ref", b.reshape(shape)); print("  bad", (np.abs(a-b).reshape(shape) > 1e-4).astype(int))
    off += cnt
rep("init", Po)
for i in range(1, D):
    rep("d%d W0" % i, H * i, (H, i)); rep("d%d b0" % i, H); rep("d%d W1" % i, H * H, (H, H)); rep("d%d b1" % i, H)
    rep("d%d W2" % i, Po * H, (Po, H)); rep("d%d b2" % i, Po)
