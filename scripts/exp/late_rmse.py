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
for k_, cast in (("flow_iterations", int), ("learning_rate", float), ("local_sample_num", int), ("posterior_sample_num", int), ("num_knots", int)):
    if os.environ.get("ARG_" + k_.upper()):                     # sensitivity runs: ARG_FLOW_ITERATIONS=1000 ...
        kwargs[k_] = cast(os.environ["ARG_" + k_.upper()])
path = os.path.join(ROOT, "tests", "data", "ManhattanPlaza136", "factor_graph.fg")
if os.environ.get("ORACLE_FIT") == "1":
    # DIAGNOSTIC ONLY: every clique fit through the CPU oracle's loop (oracle/nsf_torch.py: the reference's statements, torch
    # autograd + torch.optim.Adam) instead of the training kernels -- everything else (simulation, message passing, posterior walk)
    # unchanged.  Separates "the kernels' fits differ from the reference-equivalent fits" from "the pipeline around them differs".
    import nfisam_hip as nh
    from oracle import nsf_torch as O
    torch.set_num_threads(int(os.environ.get("ORACLE_THREADS", "4")))
    def _oracle_train(self, preps, retry=True):
        a = self._args
        for p in preps:
            K, H, B, L = p["cfg"]
            x = p["training_data"].detach().float().cpu()
            blob0 = nh.unpack(p["kp0"], p["D"], K, H, L).cpu()
            b, il, iters = O.train(x, blob0, K, H, B, L, lr=a.learning_rate, max_iters=a.flow_iterations,
                                   average_window=a.average_window, loss_delta_tol=a.loss_delta_tol, early_stop=True)
            p["trained"] = nh.pack(b.to(p["device"]), p["D"], K, H, L)
            p["iters"], p["iter_loss"] = iters, il.to(p["device"])
    NFiSAM.train_prepared = _oracle_train
if os.environ.get("NO_REUSE") == "1":
    # DIAGNOSTIC ONLY: last update's root is never re-used as a leaf (FactorGraphSolver._recycle_models: the reference re-wraps its
    # model, src/slam/FactorGraphSolver.py:306-340) -- the vanished cliques are forgotten and the new leaf is trained afresh
    from slam.FactorGraphSolver import FactorGraphSolver as _FGS
    def _forget_only(self, previous_ordering, device):
        alive = self._physical_bayes_tree.clique_nodes
        for old in [c for c in list(self._clique_density_model) if c not in alive]:
            for store in (self._clique_density_model, self._clique_true_obs, self._clique_variable_pattern, self._clique_samples):
                store.pop(old, None)
    _FGS._recycle_models = _forget_only
LATE = (20, 60, 135)
n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
seed0 = int(os.environ.get("SEED0", "0"))
def rmse_of(order_vars, A, truth):
    off, out = 0, []
    for v in order_vars:
        if str(v.name).startswith("X"):
            out.append(A[:, off:off + 2].mean(0) - truth[v][:2])
        off += v.dim
    return float(np.sqrt((np.array(out) ** 2).sum(1).mean()))
ours = {i: [] for i in LATE}
per_pose = {i: [] for i in LATE}            # DUMP=<file.npz>: per seed the posterior-mean error of every pose (name order X0, X1, ...)
def pose_errors(order_vars, A, truth):
    off, out = 0, {}
    for v in order_vars:
        if str(v.name).startswith("X"):
            out[int(str(v.name)[1:])] = np.concatenate([A[:, off:off + 2].mean(0) - truth[v][:2], A[:, off:off + 2].std(0)])
        off += v.dim
    return np.array([out[k] for k in sorted(out)])
for seed in range(seed0, seed0 + n_seeds):
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
            per_pose[i].append(pose_errors(solver.elimination_ordering, np.hstack([res[v] for v in solver.elimination_ordering]), truth))
            if seed == seed0:
                byname = {str(v.name): v for v in solver.elimination_ordering}
                order = [byname[str(n)] for n in fx["seed0_step%d_ordering" % i]]
                print("update", i, "reference seeds:", [round(rmse_of(order, fx["seed%d_step%d_samples" % (s, i)].astype(np.float64), truth), 2) for s in range(3)], flush=True)
for i in LATE:
    print("update", i, "ours:", [round(v, 2) for v in ours[i]], "median %.2f" % np.median(ours[i]), flush=True)
if os.environ.get("DUMP"):
    np.savez_compressed(os.environ["DUMP"], **{"update%d" % i: np.array(per_pose[i]) for i in LATE})
