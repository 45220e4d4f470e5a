"""Diagnostic (GPU): per-iteration losses and final parameters of the fused-Adam launches against the separate Adam kernel."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "nf-isam_amd")); sys.path.insert(0, ROOT)
import nfisam_hip as nh
import bench as BM
dev = torch.device("cuda:0")
nc, n, D, iters, wnd = [int(v) for v in sys.argv[1:6]]
K, H, B, L = 9, 8, 5.0, 1
os.environ["NFISAM_TRAIN"] = "wide"; os.environ["NFISAM_DIM_MAJOR_MIN"] = "0"
out = {}
for mode in ("0", "1", "0b", "1b"):
    os.environ["NFISAM_FUSED_ADAM"] = mode[0]
    rng = np.random.RandomState(0)
    xs = [torch.from_numpy(rng.randn(n - 37 * c, D - (c % 2)).astype(np.float32)).to(dev) for c in range(nc)]
    kps = [nh.pack(torch.from_numpy(BM.init_blob_np(x.shape[1], K, H, L, c)).to(dev), x.shape[1], K, H, L) for c, x in enumerate(xs)]
    tb = nh.TrainBatch(xs, kps, K, H, B, L, lr=0.01, max_iters=iters, average_window=wnd, loss_delta_tol=0.0, early_stop=True)
    done = tb.run(use_graph=True)
    torch.cuda.synchronize()
    out[mode] = ([t.cpu().numpy().copy() for t in tb.iter_loss], [p.cpu().numpy().copy() for p in tb.kparams], done, [p.cpu().numpy().copy() for p in tb.m], [p.cpu().numpy().copy() for p in tb.v])
for c in range(nc):
    la, lb = out["0"][0][c], out["1"][0][c]
    d = np.nonzero(la != lb)[0]
    pa, pb = out["0"][1][c], out["1"][1][c]
    print("clique %d: iterations %s / %s, first differing loss index %s, params equal %s (max |d| %.3g)" %
          (c, out["0"][2][c], out["1"][2][c], d[0] if len(d) else None, np.array_equal(pa, pb), np.abs(pa - pb).max()))
    if len(d):
        k = d[0]
        print("   losses around:", la[max(0, k - 2):k + 3], lb[max(0, k - 2):k + 3])

for a_, b_ in (("0", "0b"), ("1", "1b")):
    print("repeatable", a_, all(np.array_equal(x, y) for x, y in zip(out[a_][1], out[b_][1])))
pa, pb = out["0"][1][0], out["1"][1][0]
bad = np.nonzero(pa != pb)[0]
print("differing parameter indices:", len(bad), "of", len(pa), bad[:40], "...", bad[-10:] if len(bad) else "")

for c in range(nc):
    for name, k in (("theta", 1), ("m", 3), ("v", 4)):
        xa, xb = out["0"][k][c], out["1"][k][c]
        nd = int(np.count_nonzero(xa != xb))
        rel = np.abs(xa - xb) / np.maximum(np.abs(xa), 1e-30)
        print("clique %d %-5s differing %5d of %d, max rel %.3g" % (c, name, nd, len(xa), rel.max()))
