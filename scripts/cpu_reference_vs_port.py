"""CPU-baseline honesty check: time the TRUE reference (imported read-only from
/root/reference/src, build container only) and this repository's CPU port (oracle/nsf_torch.py) on the same
box, same threads, same clique shapes, with the reference's own training step
(src/slam/NFiSAM.py:425,469-475: zero_grad, model(x), loss = -mean(prior_logprob + log_det), backward, Adam).

Shapes: C2 (n=4096, D=6, L=4), the eight C3 cliques (n=2000, D=6..12, L=1), the Plaza clique (n=2000, D=15, L=1).
Writes a JSON table; BASELINE.md §4 and bench.py's `cpu_baseline.reference_measured` quote it.
usage: python scripts/cpu_reference_vs_port.py [out.json]   (needs /root/reference; not runnable on the GPU box)"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference/src"
import bench as BM                                    # noqa: E402  (puts nf-isam_amd on sys.path: the reference must win below)
from oracle import nsf_torch as O                     # noqa: E402
sys.path.insert(0, REF)
from flows.flows import NSF_AR                        # noqa: E402  (the reference's)
from flows.models import NormalizingFlowModel         # noqa: E402
from flows.prior_dist import CustomMultivariateNormal  # noqa: E402

K, H, B = 9, 8, 5.0
THREADS = min(os.cpu_count() or 1, 8)
torch.set_num_threads(THREADS)


def time_reference(x, L, iters, lr):
    n, D = x.shape
    torch.manual_seed(0)
    flows = [NSF_AR(dim=D, K=K, B=B, hidden_dim=H) for _ in range(L)]
    model = NormalizingFlowModel(CustomMultivariateNormal(dim=D, device="cpu"), flows)
    opt = torch.optim.Adam(model.parameters(), lr=lr)
    xt = torch.tensor(x)

    def step():
        opt.zero_grad()
        z, prior_logprob, log_det = model(xt)
        loss = -torch.mean(prior_logprob + log_det)
        loss.backward()
        opt.step()
    step()
    t0 = time.perf_counter()
    for _ in range(iters):
        step()
    return (time.perf_counter() - t0) / iters


def time_port(x, L, iters, lr):
    n, D = x.shape
    blob = torch.from_numpy(BM.init_blob_np(D, K, H, L, 0))
    xt = torch.tensor(x)
    O.train(xt, blob, K, H, B, L, lr=lr, max_iters=1, early_stop=False)
    t0 = time.perf_counter()
    O.train(xt, blob, K, H, B, L, lr=lr, max_iters=iters, early_stop=False)
    return (time.perf_counter() - t0) / iters


def main():
    rows = []
    s, circ = BM.c2_clique(4096, 0)
    cases = [("C2 n=4096 D=6 L=4", BM.normalize(s, circ)[0], 4)]
    for c, sh in enumerate(BM.C3_SHAPES):
        s, circ = BM.ring_clique(2000, *sh, np.random.RandomState(100 + c))
        x = BM.normalize(s, circ)[0]
        cases.append(("C3[%d] n=2000 D=%d L=1" % (c, x.shape[1]), x, 1))
    s, circ = BM.ring_clique(2000, *BM.PLAZA_SHAPE, np.random.RandomState(200))
    cases.append(("Plaza clique n=2000 D=15 L=1", BM.normalize(s, circ)[0], 1))
    for name, x, L in cases:
        iters = 12 if L == 4 else 25
        tr = time_reference(x, L, iters, 0.01)
        tp = time_port(x, L, iters, 0.01)
        rows.append(dict(case=name, n=int(x.shape[0]), D=int(x.shape[1]), L=L, reference_ms_per_step=1e3 * tr,
                         port_ms_per_step=1e3 * tp, reference_samples_per_s=x.shape[0] / tr,
                         port_samples_per_s=x.shape[0] / tp, port_over_reference=tr / tp))
        print("%-30s reference %7.2f ms  port %7.2f ms  port/reference speed %.2fx" % (name, 1e3 * tr, 1e3 * tp, tr / tp),
              flush=True)
    c3 = [r for r in rows if r["case"].startswith("C3")]
    agg = dict(case="C3 whole batch (8 cliques, sequential on the CPU)",
               reference_samples_per_s=8 * 2000 / sum(r["reference_ms_per_step"] for r in c3) * 1e3,
               port_samples_per_s=8 * 2000 / sum(r["port_ms_per_step"] for r in c3) * 1e3)
    agg["port_over_reference"] = agg["port_samples_per_s"] / agg["reference_samples_per_s"]
    out = dict(threads=THREADS, cpu=open("/proc/cpuinfo").read().split("model name")[1].split("\n")[0].strip(": \t"),
               torch=torch.__version__, rows=rows, c3_batch=agg)
    print(json.dumps(out["c3_batch"]))
    if len(sys.argv) > 1:
        json.dump(out, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
