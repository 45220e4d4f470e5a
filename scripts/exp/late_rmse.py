"""Diagnostic (GPU): trajectory RMSE (posterior mean of pose xy vs the .fg ground truth) of the complete Manhattan-136 run at
updates 20 / 60 / 135, N seeds of this repository's solver next to the reference's three seeds (fixture).
    python scripts/exp/late_rmse.py [seeds=10] [env: DEVICE_SIM=0 -> host simulator]"""
import json, os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "nf-isam_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from slam.NFiSAM import NFiSAM, NFiSAMArgs
from slam.RunBatch import graph_file_parser, group_nodes_factors_incrementally
fx = np.load(os.path.join(ROOT, "tests", "golden", "pipeline_manhattan136_full.npz"))
kwargs = json.loads(str(fx["arguments"])); kwargs["cuda_training"] = True
if os.environ.get("DEVICE_SIM") == "0":
    kwargs["device_simulation"] = False
path = os.path.join(ROOT, "tests", "data", "ManhattanPlaza136", "factor_graph.fg")
LATE = (20, 60, 135)
n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
def rmse_of(order_vars, A, truth):
    off, out = 0, []
    for v in order_vars:
        if str(v.name).startswith("X"):
            out.append(A[:, off:off + 2].mean(0) - truth[v][:2])
        off += v.dim
    return float(np.sqrt((np.array(out) ** 2).sum(1).mean()))
ours = {i: [] for i in LATE}
for seed in range(n_seeds):
    random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)
    nodes, truth, factors = graph_file_parser(path, "fg", prior_cov_scale=0.1)
    steps = group_nodes_factors_incrementally(nodes, factors, incremental_step=1)
    solver = NFiSAM(NFiSAMArgs(**kwargs))
    for i, (vs, fs) in enumerate(steps):
        for v in vs: solver.add_node(v)
        for f in fs: solver.add_factor(f)
        solver.update_physical_and_working_graphs()
        res = solver.incremental_inference()
        if i in LATE:
            ours[i].append(rmse_of(solver.elimination_ordering, np.hstack([res[v] for v in solver.elimination_ordering]), truth))
            if seed == 0:
                byname = {str(v.name): v for v in solver.elimination_ordering}
                order = [byname[str(n)] for n in fx["seed0_step%d_ordering" % i]]
                print("update", i, "reference seeds:", [round(rmse_of(order, fx["seed%d_step%d_samples" % (s, i)].astype(np.float64), truth), 2) for s in range(3)], flush=True)
for i in LATE:
    print("update", i, "ours:", [round(v, 2) for v in ours[i]], "median %.2f" % np.median(ours[i]), flush=True)
