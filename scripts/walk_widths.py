"""Posterior walk of a Plaza-shaped chain (C cliques: 11 given columns, 4 frontal, D = 15, n samples) per hidden width:
the pipelined two-lanes-per-sample kernel against the plain one-lane walk (NFISAM_WALK=plain).
    python scripts/walk_widths.py [C=160] [n=1000]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nf-isam_amd"))
import nfisam_hip as nh
C = int(sys.argv[1]) if len(sys.argv) > 1 else 160
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
DEV = torch.device("cuda:0")
K, B, D, F = 9, 5.0, 15, 4
for H in (4, 8, 16):
    torch.manual_seed(H)
    entries = []
    for c in range(C):
        kp = (0.05 * torch.randn(nh.kparam_count(D, K, H), device=DEV)).contiguous()
        first = c == 0
        f0 = 0 if first else 15 + F * (c - 1)                  # root: 15 frontal columns; then 4 per clique, 10 separator columns before them
        sep = [] if first else list(range(f0 - 10, f0))
        entries.append(dict(kparams=kp, mean=torch.zeros(D, device=DEV), std=torch.ones(D, device=DEV),
                            circular=torch.zeros(D, dtype=torch.uint8, device=DEV), D_model=D,
                            obs=np.zeros(0 if first else 1), sep_cols=sep,
                            front_cols=list(range(0, 15)) if first else list(range(f0, f0 + F))))
    total = max(max(e["front_cols"]) for e in entries) + 1
    Zt = torch.randn(total, n, device=DEV)
    out = {}
    for mode in ("", "plain"):
        if mode: os.environ["NFISAM_WALK"] = mode
        else: os.environ.pop("NFISAM_WALK", None)
        S = nh.posterior_walk(entries, total, n, K, H, B, 1, DEV, Zt=Zt)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for _ in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            S = nh.posterior_walk(entries, total, n, K, H, B, 1, DEV, Zt=Zt)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        S2 = nh.posterior_walk(entries, total, n, K, H, B, 1, DEV, Zt=Zt)
        assert torch.equal(S, S2), "the walk is not reproducible"
        out[mode or "pipelined"] = (min(ts) * 1e3, S)
    os.environ.pop("NFISAM_WALK", None)
    d = (out["pipelined"][1] - out["plain"][1]).abs().max().item()
    print("H = %2d  %d cliques, n = %d: pipelined-capable path %.2f ms, plain walk %.2f ms per call (host table build included); max |difference| %.2e"
          % (H, C, n, out["pipelined"][0], out["plain"][0], d))
