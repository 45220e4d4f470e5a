"""Experiment (GPU): what slows the host down while a training plan runs in a worker thread?
Measures, with and without a plan running: a pure-Python loop (interpreter lock), a tiny kernel + .item() round trip on
the default stream, an H2D copy of 64 floats, a hipMemset-like zero_ + synchronize."""
import os, sys, time, threading
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "nf-isam_amd")); sys.path.insert(0, ROOT)
import nfisam_hip as nh
import bench as BM
dev = torch.device("cuda:0")
K, H, B, L = 9, 8, 5.0, 1
rng = np.random.RandomState(0)
xs = [torch.from_numpy(rng.randn(2000, 15).astype(np.float32)).to(dev) for _ in range(4)]
kps = [nh.pack(torch.from_numpy(BM.init_blob_np(15, K, H, L, c)).to(dev), 15, K, H, L) for c in range(4)]
tb = nh.TrainBatch(xs, kps, K, H, B, L, lr=0.01, max_iters=60000, early_stop=False)
tb.prepare(use_graph=True)
a = torch.zeros(64, device=dev)
h = torch.zeros(64)

def probes(tag):
    out = {}
    t0 = time.perf_counter(); s = 0
    for i in range(300000): s += i * i
    out["python loop ms"] = 1e3 * (time.perf_counter() - t0)
    for name, fn in (("kernel + .item() us", lambda: float((a + 1.0).sum().item())),
                     ("H2D 64 floats us", lambda: a.copy_(h, non_blocking=False)),
                     ("zero_ + sync us", lambda: (a.zero_(), torch.cuda.current_stream().synchronize())),
                     ("launch only us", lambda: a.add_(1.0))):
        fn()
        t0 = time.perf_counter()
        for _ in range(300): fn()
        out[name] = 1e6 * (time.perf_counter() - t0) / 300
    torch.cuda.synchronize() if tag == "idle" else None
    print(tag, {k: round(v, 1) for k, v in out.items()})

probes("idle")
th = threading.Thread(target=lambda: tb.run(use_graph=True))
t0 = time.perf_counter(); th.start(); time.sleep(0.05)
probes("next to a running plan")
probes("next to a running plan (again)")
th.join(); print("plan ran %.2f s" % (time.perf_counter() - t0))
probes("idle again")
