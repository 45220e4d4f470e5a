import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nf-isam_amd")); sys.path.insert(0, ROOT)
import nfisam_hip as nh
import bench as BM
from oracle import c_oracle as CO
dev = torch.device("cuda:0"); K, H, B = 9, 8, 5.0
n, D, L = 1000, 6, 3
rng = np.random.RandomState(5)
x = rng.randn(n, D).astype(np.float32)
blob = BM.init_blob_np(D, K, H, L, 3)
xd = torch.from_numpy(x).to(dev); kp = nh.pack(torch.from_numpy(blob).to(dev), D, K, H, L)
g = [nh.backward(xd, kp, K, H, B, L, nll_mode=True)[0] for _ in range(4)]
ref = g[0]
print("single backward: max rel diff between repeated calls:", [float((q - ref).abs().max() / ref.abs().max()) for q in g[1:]])
lo, go, _, _ = CO.nll_grad(x, blob, K, H, B, L, dtype=np.float64)
print("vs oracle rel:", float(np.abs(nh.unpack(ref, D, K, H, L).cpu().numpy() / n - go).max() / np.abs(go).max()))
def run(graph, iters=40):
    tb = nh.TrainBatch([xd], [kp.clone()], K, H, B, L, lr=0.02, max_iters=iters, early_stop=False)
    tb.run(use_graph=graph); return tb.kparams[0].clone(), tb.iter_loss[0].cpu().numpy()
for it in (1, 2, 5, 10, 40):
    a = [run(False, it) for _ in range(2)] + [run(True, it) for _ in range(2)]
    d = lambda i, j: float((a[i][0] - a[j][0]).abs().max())
    print("iters %2d: eager-eager %.2e graph-graph %.2e eager-graph %.2e | loss last %s" % (it, d(0, 1), d(2, 3), d(0, 2), [round(float(q[1][it-1]), 5) for q in a]))
bo, lo_, _, _, _ = CO.train(x, blob, K, H, B, L, lr=0.02, max_iters=10, early_stop=False, dtype=np.float32)
a = run(True, 10)
print("10 iters vs oracle: param max diff %.2e, 99%% quantile %.2e; losses gpu %s oracle %s" % (
    np.abs(nh.unpack(a[0], D, K, H, L).cpu().numpy() - bo).max(), np.quantile(np.abs(nh.unpack(a[0], D, K, H, L).cpu().numpy() - bo), 0.99), np.round(a[1][:10], 4), np.round(lo_, 4)))
