"""Diagnostic (GPU): A/B of library builds on ONE device, interleaved rounds (devices differ by several per cent, so
timings of different gpurun calls do not compare).
    python scripts/ab.py [rounds] variant ...      variant = lib.so[,ENV=value,...]  (file names inside nf-isam_amd/nfisam_hip/;
                                                   e.g. libnfisam_hip.so,NFISAM_PERSIST=0)
Each round runs every library once in its own process: C3, one Plaza clique, the 64-clique batch (bench.Workload)."""
import json, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

if os.environ.get("AB_LIB"):
    sys.path.insert(0, os.path.join(ROOT, "nf-isam_amd")); sys.path.insert(0, ROOT)
    import torch
    import nfisam_hip as nh
    nh.LIB_PATH = os.path.join(os.path.dirname(nh.LIB_PATH), os.environ["AB_LIB"])
    if os.environ.get("AB_OLD_ABI"):       # a library of an earlier round: only the entry points it has (plans without the newer diagnostics)
        import ctypes
        _old = ctypes.CDLL(nh.LIB_PATH)
        nh.EXPORTS[:] = [e for e in nh.EXPORTS if hasattr(_old, e)]
        nh.TrainBatch.xcd_span = lambda self: 0
    import bench as BM
    if os.environ.get("AB_K"):             # another num_knots than the bench's 9
        BM.K = int(os.environ["AB_K"])
    dev = torch.device("cuda:0")
    out = {}
    which = os.environ.get("AB_REGIMES", "c3,plaza,b64").split(",")
    for name in which:
        if name == "c3":
            prob, L = BM.c3_problem(0), 1
        else:
            if name == "r8":                # eight Plaza-shaped cliques: the replica conveyor's launch
                prob, L = BM.regime_problem("batch64_n2000_D15", 0)
                prob = prob[:8]
            else:
                prob, L = BM.regime_problem({"plaza": "plaza_clique_n2000_D15", "b64": "batch64_n2000_D15",
                                             "c2": "C2_single_clique_n4096_D6_L4"}[name], 0)
        w = BM.Workload(prob, L, dev)
        r, _ = w.record(300, 30, torch.cuda.synchronize)
        out[name] = (r["gradient_kernel_us"], r["us_per_iteration"])
    print("AB_RESULT " + json.dumps(out))
    sys.exit(0)

args = sys.argv[1:]
rounds = int(args.pop(0)) if args and args[0].isdigit() else 3
libs = args
res = {l: [] for l in libs}
for r in range(rounds):
    for l in libs:
        parts = l.split(",")
        env = dict(os.environ, AB_LIB=parts[0], **dict(kv.split("=", 1) for kv in parts[1:]))
        o = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True)
        line = [x for x in o.stdout.split("\n") if x.startswith("AB_RESULT ")]
        if not line:
            print(l, "FAILED", o.stderr[-400:]); continue
        res[l].append(json.loads(line[0][10:]))
for l in libs:
    if not res[l]:
        continue
    s = "%-60s" % l
    for name in res[l][0]:
        k = np.median([x[name][0] for x in res[l]]); st = np.median([x[name][1] for x in res[l]])
        s += "  %s kernel %7.2f step %7.2f |" % (name, k, st)
    print(s)
