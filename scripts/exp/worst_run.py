"""Diagnostic (GPU): one seed of the complete Manhattan-136 problem, first UPDATES updates, with the trajectory RMSE after every
update and, per clique fit, its shape / iterations / final loss; DUMP=<file.npz> keeps every fit's normalised batch, initial and
trained kernel-layout parameters so that the same fits can be repeated through the CPU oracle off the box.
    SEED=53 ARG_LOCAL_SAMPLE_NUM=4000 UPDATES=21 DUMP=gpurun_out/worst.npz python scripts/exp/worst_run.py"""
import json, os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "nf-isam_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from slam.NFiSAM import NFiSAM, NFiSAMArgs
from slam.RunBatch import graph_file_parser, group_nodes_factors_incrementally
fx = np.load(os.path.join(ROOT, "tests", "golden", "pipeline_manhattan136_full.npz"))
kwargs = json.loads(str(fx["arguments"])); kwargs["cuda_training"] = True
if os.environ.get("ARG_LOCAL_SAMPLE_NUM"):
    kwargs["local_sample_num"] = int(os.environ["ARG_LOCAL_SAMPLE_NUM"])
seed, updates = int(os.environ.get("SEED", "0")), int(os.environ.get("UPDATES", "21"))
path = os.path.join(ROOT, "tests", "data", "ManhattanPlaza136", "factor_graph.fg")
fits, upd = [], [0]
orig = NFiSAM._train_prepared_locked
def traced(self, preps, retry):
    r = orig(self, preps, retry)
    for p in preps:
        il = p["iter_loss"].detach().float().cpu().numpy()
        fits.append(dict(update=upd[0], n=p["n"], D=p["D"], sep=p["sep_dim"], iters=int(p["iters"]), loss=il[:int(p["iters"])].copy(),
                         x=p["training_data"].detach().float().cpu().numpy(), kp0=p["kp0"].detach().cpu().numpy(),
                         kp=p["trained"].detach().cpu().numpy(), vars=[str(v.name) for v in p["clique"].vars]))
    return r
NFiSAM._train_prepared_locked = traced
random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)
nodes, truth, factors = graph_file_parser(path, "fg", prior_cov_scale=0.1)
steps = group_nodes_factors_incrementally(nodes, factors, incremental_step=1)[:updates]
solver = NFiSAM(NFiSAMArgs(**kwargs))
for i, (vs, fs) in enumerate(steps):
    upd[0] = i
    for v in vs: solver.add_node(v)
    for f in fs: solver.add_factor(f)
    solver.update_physical_and_working_graphs()
    res = solver.incremental_inference()
    err = {str(v.name): float(np.linalg.norm(res[v][:, :2].mean(0) - truth[v][:2])) for v in solver.elimination_ordering if str(v.name).startswith("X")}
    last = [f for f in fits if f["update"] == i]
    print("update %3d rmse %.2f last-pose err %.2f | fits: %s" % (i, float(np.sqrt(np.mean(np.square(list(err.values()))))), err["X%d" % max(int(k[1:]) for k in err)],
          "; ".join("%s n%d D%d sep%d it%d loss %.3f->%.3f (min %.3f)" % ("".join(f["vars"]), f["n"], f["D"], f["sep"], f["iters"], f["loss"][0], f["loss"][-1], f["loss"].min()) for f in last)), flush=True)
if os.environ.get("DUMP"):
    out = {}
    for j, f in enumerate(fits):
        out["fit%d_x" % j], out["fit%d_kp0" % j], out["fit%d_kp" % j], out["fit%d_loss" % j] = f["x"], f["kp0"], f["kp"], f["loss"]
        out["fit%d_meta" % j] = np.array(json.dumps(dict(update=f["update"], n=f["n"], D=f["D"], sep=f["sep"], vars=f["vars"])))
    out["arguments"] = np.array(json.dumps(kwargs))
    np.savez_compressed(os.environ["DUMP"], **out)
