"""Experiment (GPU): R single-clique training plans on R streams, driven from R host threads (the C call releases the GIL),
against ONE batched plan of the same R cliques.  argv: R n D iters"""
import os, sys, time, threading
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "nf-isam_amd")); sys.path.insert(0, ROOT)
import nfisam_hip as nh
import bench as BM
R, n, D, iters = [int(v) for v in sys.argv[1:5]]
K, H, B, L = 9, 8, 5.0, 1
dev = torch.device("cuda:0")
rng = np.random.RandomState(0)
xs = [torch.from_numpy(rng.randn(n, D).astype(np.float32)).to(dev) for _ in range(R)]
kps = [nh.pack(torch.from_numpy(BM.init_blob_np(D, K, H, L, c)).to(dev), D, K, H, L) for c in range(R)]
# batched
tb = nh.TrainBatch(xs, [k.clone() for k in kps], K, H, B, L, lr=0.01, max_iters=iters, early_stop=False)
tb.prepare(use_graph=True); torch.cuda.synchronize()
t0 = time.perf_counter(); tb.run(use_graph=True); torch.cuda.synchronize(); t_batch = time.perf_counter() - t0
# concurrent single-clique plans
tbs = [nh.TrainBatch([xs[c]], [kps[c].clone()], K, H, B, L, lr=0.01, max_iters=iters, early_stop=False) for c in range(R)]
streams = [torch.cuda.Stream() for _ in range(R)]
for c in range(R):
    tbs[c].prepare(use_graph=True)
torch.cuda.synchronize()
def work(c):
    with torch.cuda.stream(streams[c]):
        tbs[c].run(use_graph=True)
t0 = time.perf_counter()
th = [threading.Thread(target=work, args=(c,)) for c in range(R)]
for t in th: t.start()
for t in th: t.join()
torch.cuda.synchronize(); t_conc = time.perf_counter() - t0
# one after the other
t0 = time.perf_counter()
for c in range(R):
    tbs[c].reset([kps[c].clone()]) if hasattr(tbs[c], "reset") else None
torch.cuda.synchronize()
t0 = time.perf_counter()
for c in range(R):
    tbs[c].run(use_graph=True)
torch.cuda.synchronize(); t_seq = time.perf_counter() - t0
print("R=%d n=%d D=%d iters=%d: batched %.2f ms (%.2f us/iter), %d concurrent plans %.2f ms (%.2f us per iteration-round), sequential %.2f ms" %
      (R, n, D, iters, 1e3 * t_batch, 1e6 * t_batch / iters, R, 1e3 * t_conc, 1e6 * t_conc / iters, 1e3 * t_seq))
