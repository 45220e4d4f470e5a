"""One-off (GPU): the chunk-persistent dim-major kernel (NFISAM_PERSIST=1) against the plain one: bitwise equality of the
trained parameters / loss curves, and time per iteration.  usage: persist_check.py [plaza|c3]  (run once per setting: the
knob is read once per process)"""
import os, sys, time, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "nf-isam_amd")); sys.path.insert(0, ROOT)
import nfisam_hip as nh
from flows.flows import init_reference_blob
which = sys.argv[1] if len(sys.argv) > 1 else "plaza"
out = sys.argv[2] if len(sys.argv) > 2 else None
dev = torch.device("cuda", 0)
K, H, B, L = 9, 8, 5.0, 1
Ds = [15] if which == "plaza" else [6, 8, 8, 10, 10, 12, 12, 12]
ns = [2000] * len(Ds) if which != "ragged" else [2000, 1500, 777, 2000, 300, 2000, 1999, 64]
torch.manual_seed(0)
xs = [torch.randn(n, D, device=dev).clamp_(-3, 3) * 0.8 for n, D in zip(ns, Ds)]
kp = [nh.pack(init_reference_blob(D, K, H, dev), D, K, H, 1) for D in Ds]
iters = 300
tb = nh.TrainBatch(xs, [p.clone() for p in kp], K, H, B, L, lr=0.01, max_iters=iters, average_window=50, loss_delta_tol=0.0, early_stop=True)
done = tb.run()
torch.cuda.synchronize()
ts = []
for r in range(5):
    tb.reset(kp)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    done = tb.run()
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print(which, "PERSIST=%s" % os.environ.get("NFISAM_PERSIST"), "iters", done, "us/iter %.2f" % (1e6 * min(ts) / iters),
      "loss0 %.5f lossN %.5f" % (float(tb.iter_loss[0][0]), float(tb.iter_loss[0][iters - 1])))
if out:
    np.savez(out, *[t.cpu().numpy() for t in tb.kparams], *[t.cpu().numpy() for t in tb.iter_loss])
