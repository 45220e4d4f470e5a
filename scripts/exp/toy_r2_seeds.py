"""Diagnostic (GPU): the toy R2 range-only graph of tests/test_configs_gpu.py over several solver seeds; per landmark the
fraction of posterior samples within 6 m of the truth (the true mode's weight)."""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "nf-isam_amd"))
import numpy as np, torch
import test_configs_gpu as T
from slam.NFiSAM import NFiSAM, NFiSAMArgs
from slam.RunBatch import group_nodes_factors_incrementally
nodes, truth, factors = T._toy_r2_graph()
steps = group_nodes_factors_incrementally(nodes, factors, incremental_step=5)
for seed in range(int(sys.argv[1])):
    random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)
    solver = NFiSAM(NFiSAMArgs(num_knots=9, flow_iterations=1500, local_sample_num=2000, learning_rate=.02, hidden_dim=8,
                               cuda_training=True, elimination_method="pose_first", training_set_frac=1.0, loss_delta_tol=.01,
                               posterior_sample_num=500))
    for vs, fs in steps:
        for v in vs: solver.add_node(v)
        for f in fs: solver.add_factor(f)
        solver.update_physical_and_working_graphs()
        res = solver.incremental_inference()
    name = {str(v.name): v for v in solver.physical_vars}
    tn = {str(k.name): v for k, v in truth.items()}
    err = np.array([res[name["X%d" % i]].mean(0) - tn["X%d" % i] for i in range(20)])
    w = [float(np.mean(np.linalg.norm(res[name["L%d" % j]] - tn["L%d" % j], axis=1) < 6.0)) for j in range(4)]
    print("seed %d: pose rmse %.2f, true-mode weight per landmark %s" % (seed, np.sqrt((err ** 2).sum(1).mean()), np.round(w, 2)))
