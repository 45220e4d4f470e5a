"""cProfile of the host side of incremental updates (Plaza1, first N updates)."""
import cProfile, pstats, os, sys, io
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nf-isam_amd")); sys.path.insert(0, ROOT)
from slam.NFiSAM import NFiSAM, NFiSAMArgs
from slam.RunBatch import graph_file_parser, group_nodes_factors_incrementally
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
np.random.seed(0); torch.manual_seed(0)
nodes, truth, factors = graph_file_parser(os.path.join(ROOT, "tests", "data", "Plaza1EFG", "factor_graph.fg"), "fg")
steps = group_nodes_factors_incrementally(nodes, factors, incremental_step=5)
solver = NFiSAM(NFiSAMArgs(num_knots=9, flow_iterations=2000, local_sample_num=2000, learning_rate=.01, hidden_dim=8,
                           elimination_method="pose_first", loss_delta_tol=.01, average_window=50))
def run(lo, hi):
    for vs, fs in steps[lo:hi]:
        for v in vs: solver.add_node(v)
        for f in fs: solver.add_factor(f)
        solver.update_physical_and_working_graphs()
        solver.incremental_inference()
run(0, 5)     # warm-up
pr = cProfile.Profile(); pr.enable(); run(5, N); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats(os.environ.get("SORT", "cumulative")).print_stats(int(os.environ.get("TOP", "45"))); print(s.getvalue()[:9000])
