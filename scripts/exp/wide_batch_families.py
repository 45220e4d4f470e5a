import os, sys
sys.path.insert(0, "/root/repo/nf-isam_amd"); sys.path.insert(0, "/root/repo")
import numpy as np, torch
import nfisam_hip as nh, bench as BM
dev = torch.device("cuda:0"); K, H, B = 9, 8, 5.0
res = {}
for mode in ("0", "1"):
    os.environ["NFISAM_DIM_MAJOR"] = mode
    rng = np.random.RandomState(3)
    shapes = [(2000, 18 + (c % 3)) for c in range(24)]
    xs = [torch.from_numpy(rng.randn(n, D).astype(np.float32)).to(dev) for n, D in shapes]
    kps = [nh.pack(torch.from_numpy(BM.init_blob_np(D, K, H, 1, c)).to(dev), D, K, H, 1) for c, (n, D) in enumerate(shapes)]
    tb = nh.TrainBatch(xs, kps, K, H, B, 1, lr=0.01, max_iters=1, average_window=1, loss_delta_tol=0.0, early_stop=True)
    tb.run(use_graph=False); torch.cuda.synchronize()
    res[mode] = ([m.cpu().numpy() for m in tb.m], [l.cpu().numpy()[:1] for l in tb.iter_loss])
worst = 0
for a, b in zip(res["0"][0], res["1"][0]):
    worst = max(worst, np.abs(a - b).max() / max(1e-6, np.abs(a).max()))
print("max relative difference of the first moments (0.1 x gradient sums) between tile-major and dim-major kernels: %.3g" % worst)
print("losses", res["0"][1][0], res["1"][1][0])
