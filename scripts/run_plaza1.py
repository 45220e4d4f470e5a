"""BASELINE config 5 end to end: Plaza1 range-only dataset (778 poses, 4 landmarks), arguments of
example/slam/plaza_dataset/run_nfisam.py:5-21 (K=9, n=2000, <=2000 it, lr .01, window 50, tol .01,
incremental_step=5 -> 156 updates), on the MI355X back end.  Reports wall-clock per incremental update
with the reference's sub-timers, flow-training samples/s, and trajectory RMSE against the ground truth
stored in the .fg file.   usage: run_plaza1.py [max_updates] [out.json]"""
import json, os, sys, tempfile, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nf-isam_amd")); sys.path.insert(0, ROOT)
from slam.NFiSAM import NFiSAM, NFiSAMArgs
from slam.RunBatch import graph_file_parser, group_nodes_factors_incrementally

max_updates = int(sys.argv[1]) if len(sys.argv) > 1 else 10 ** 9
out_json = sys.argv[2] if len(sys.argv) > 2 else None
seed = int(os.environ.get('SEED', '0'))
np.random.seed(seed); torch.manual_seed(seed)
dataset = os.environ.get("DATASET", "Plaza1EFG")
nodes, truth, factors = graph_file_parser(os.path.join(ROOT, "tests", "data", dataset, "factor_graph.fg"), "fg")
steps = group_nodes_factors_incrementally(nodes, factors, incremental_step=int(os.environ.get("STEP", "5")))
args = NFiSAMArgs(num_knots=9, flow_iterations=int(os.environ.get("ITERS", "2000")), local_sample_num=2000,
                  learning_rate=.01, hidden_dim=8, cuda_training=True, elimination_method="pose_first",
                  training_set_frac=1.0, loss_delta_tol=float(os.environ.get("TOL", ".01")), average_window=50,
                  device_simulation=os.environ.get("DEVSIM", "1") != "0",
                  lazy_posterior=os.environ.get("LAZY", "0") == "1",      # LAZY=1: the samples of update i are read under update i + 1
                  async_fits=os.environ.get("ASYNC", "0") == "1")         # ASYNC=1: the fits of an update are enqueued, their outcomes read once
replicas = int(os.environ.get("REPLICAS", "1"))
if replicas > 1:
    # R independent runs (seeds SEED .. SEED+R-1) on one GPU, their cliques in the R slots of one batched training plan
    # (slam.ReplicaNFiSAM; the reference loops over its dataset variants one after the other, run_nfisam.py:11-21)
    from slam.ReplicaNFiSAM import ReplicaNFiSAM
    rep = ReplicaNFiSAM(args, [seed + r for r in range(replicas)])
    walls, fit_iters = [], 0
    t_all = time.time()
    if os.environ.get("FREE", "1") != "0":
        # every replica at its own pace (ReplicaNFiSAM.run_incrementally): no barrier per update
        per = [[] for _ in range(replicas)]
        last = [None] * replicas

        def on_update(r, k, samples, seconds):
            per[r].append(seconds)
            last[r] = samples
        rep.run_incrementally(steps[:max_updates], on_update=on_update)
        total = time.time() - t_all
        fit_iters = sum(rep.fit_iterations)
        rm = []
        for s, o in zip(rep.solvers, last):
            poses = [v for v in s.physical_vars if str(v.name).startswith("X")]
            err = np.array([o[v][:, :2].mean(0) - truth[v][:2] for v in poses])
            rm.append(float(np.sqrt((err ** 2).sum(1).mean())))
        n_up = len(per[0])
        print("free-running replicas: %.2f s for %d replicas x %d updates = %.2f ms per replica-update, traj RMSE %s" %
              (total, replicas, n_up, 1e3 * total / (replicas * n_up), " ".join("%.2f" % r for r in rm)), flush=True)
        summary = dict(replicas=replicas, updates=n_up, total_s=total, schedule="free-running (run_incrementally)",
                       wall_per_replica_update_mean=float(total / (replicas * n_up)),
                       wall_per_replica_update_median=float(total / (replicas * n_up)),
                       own_seconds_per_update_median=float(np.median(np.concatenate([np.array(p) for p in per]))),
                       training_sample_iters=float(2000 * fit_iters), final_rmse=rm, seconds_by_phase=getattr(rep, 'profile', None))
        print(json.dumps(summary))
        if out_json:
            json.dump(dict(summary=summary, per_replica_seconds=per), open(out_json, "w"))
        sys.exit(0)
    for i, (vs, fs) in enumerate(steps[:max_updates]):
        for v in vs: rep.add_node(v)
        for f in fs: rep.add_factor(f)
        t0 = time.time()
        outs = rep.update()
        walls.append(time.time() - t0)
        fit_iters += sum(int(np.count_nonzero(v)) for s in rep.solvers for v in s._temp_training_loss.values())
        if i % int(os.environ.get('EVERY', '10')) == 0 or i == min(len(steps), max_updates) - 1:
            rm = []
            for s, o in zip(rep.solvers, outs):
                poses = [v for v in s.physical_vars if str(v.name).startswith("X")]
                err = np.array([o[v][:, :2].mean(0) - truth[v][:2] for v in poses])
                rm.append(float(np.sqrt((err ** 2).sum(1).mean())))
            print("update %3d: %.3f s for %d replicas = %.1f ms per replica-update, batches %s, traj RMSE %s" %
                  (i, walls[-1], replicas, 1e3 * walls[-1] / replicas, rep.last_batches, " ".join("%.2f" % r for r in rm)), flush=True)
    total = time.time() - t_all
    w = np.array(walls)
    summary = dict(replicas=replicas, updates=len(walls), total_s=total, wall_per_update_all_replicas_median=float(np.median(w)),
                   wall_per_replica_update_mean=float(w.mean() / replicas), wall_per_replica_update_median=float(np.median(w) / replicas),
                   training_sample_iters=float(2000 * fit_iters), final_rmse=rm, seconds_by_phase=getattr(rep, 'profile', None))
    print(json.dumps(summary))
    if out_json:
        json.dump(dict(summary=summary, walls=walls), open(out_json, "w"))
    sys.exit(0)
solver = NFiSAM(args)
rows = []
t_all = time.time()
lazy = bool(getattr(args, "lazy_posterior", False))
pending = None                                               # LAZY=1: (row, samples, poses) of the previous update
lazy_wait = 0.0


rmse_seconds = [0.0]


def rmse_of(samples, poses_):
    t_ = time.time()
    try:
        return _rmse_of(samples, poses_)
    finally:
        rmse_seconds[0] += time.time() - t_


def _rmse_of(samples, poses_):
    # (one mean over the stacked xy columns: a numpy call per variable costs 0.7 s over the run)
    err_ = np.hstack([samples[v][:, :2] for v in poses_]).mean(0).reshape(-1, 2) - np.array([truth[v][:2] for v in poses_])
    return float(np.sqrt((err_ ** 2).sum(1).mean()))


for i, (vs, fs) in enumerate(steps[:max_updates]):
    for v in vs: solver.add_node(v)
    for f in fs: solver.add_factor(f)
    timer = []
    t0 = time.time()
    solver.update_physical_and_working_graphs(timer=timer)
    samples = solver.incremental_inference(timer=timer)
    dt = time.time() - t0
    if pending is not None:                                  # the previous update's samples: its walk ran under this update's host work
        t_r = time.time()
        pending[1][pending[2][0]]                            # (first access: waits for the walk's event if it is still running)
        lazy_wait += time.time() - t_r
        pending[0]["rmse"] = rmse_of(pending[1], pending[2])
        pending = None
    loss = solver._temp_training_loss
    iters = [int(np.count_nonzero(v)) for v in loss.values()]
    fit = sum(timer[2:-1:2]) if len(timer) > 2 else 0.0      # [graph, (sample, fit)*, posterior]
    samp = sum(timer[1:-1:2])
    poses_ = [v for v in solver.physical_vars if str(v.name).startswith("X")]
    last_ = i == min(len(steps), max_updates) - 1
    rmse_ = float("nan") if (lazy and not last_) else rmse_of(samples, poses_)
    rows.append(dict(update=i, rmse=rmse_, wall=dt, graph=timer[0], sampling=samp, fitting=fit, posterior=timer[-1],
                     cliques_trained=len(iters), iterations=sum(iters), n_vars=len(solver.physical_vars)))
    if lazy and not last_:
        pending = (rows[-1], samples, poses_)
    if i % int(os.environ.get('EVERY', '10')) == 0 or i == min(len(steps), max_updates) - 1:
        print("update %3d: %.3f s (graph %.3f, sampling %.3f, fit %.3f [%d cliques, %d it], posterior %.3f) vars %d "
              "traj RMSE %.2f m" % (i, dt, timer[0], samp, fit, len(iters), sum(iters), timer[-1],
                                    len(solver.physical_vars), rmse_), flush=True)
total = time.time() - t_all
w = np.array([r["wall"] for r in rows]); f = np.array([r["fitting"] for r in rows]); it = np.array([r["iterations"] for r in rows])
summary = dict(updates=len(rows), total_s=total, wall_per_update_mean=float(w.mean()), wall_per_update_median=float(np.median(w)),
               wall_per_update_max=float(w.max()), fitting_total_s=float(f.sum()), training_sample_iters=float(2000 * it.sum()),
               flow_training_samples_per_s=float(2000 * it.sum() / max(f.sum(), 1e-9)),
               sampling_total_s=float(sum(r["sampling"] for r in rows)), posterior_total_s=float(sum(r["posterior"] for r in rows)),
               graph_total_s=float(sum(r["graph"] for r in rows)), lazy_posterior=lazy, lazy_first_access_wait_s=float(lazy_wait),
               rmse_bookkeeping_of_this_script_s=float(rmse_seconds[0]))
print(json.dumps(summary))
if out_json:
    json.dump(dict(summary=summary, rows=rows), open(out_json, "w"))
