"""Host/GPU split of sample_posterior on the full Plaza1 tree (782 variables): cProfile of 10 calls + kernel-only timing."""
import cProfile, pstats, os, sys, io, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nf-isam_amd")); sys.path.insert(0, ROOT)
from slam.NFiSAM import NFiSAM, NFiSAMArgs
from slam.RunBatch import graph_file_parser, group_nodes_factors_incrementally
N = int(sys.argv[1]) if len(sys.argv) > 1 else 156
np.random.seed(0); torch.manual_seed(0)
nodes, truth, factors = graph_file_parser(os.path.join(ROOT, "tests", "data", "Plaza1EFG", "factor_graph.fg"), "fg")
steps = group_nodes_factors_incrementally(nodes, factors, incremental_step=5)
solver = NFiSAM(NFiSAMArgs(num_knots=9, flow_iterations=2000, local_sample_num=2000, learning_rate=.01, hidden_dim=8,
                           elimination_method="pose_first", loss_delta_tol=.01, average_window=50))
for vs, fs in steps[:N]:
    for v in vs: solver.add_node(v)
    for f in fs: solver.add_factor(f)
    solver.update_physical_and_working_graphs()
    solver.incremental_inference()
torch.cuda.synchronize()
for mode in ("", "plain"):
    if mode: os.environ["NFISAM_WALK"] = mode
    solver.sample_posterior()
    t0 = time.perf_counter()
    for _ in range(10): solver.sample_posterior()
    print("sample_posterior [%s]: %.2f ms per call, %d variables" % (mode or "pipelined", (time.perf_counter() - t0) * 100,
                                                                     len(solver.physical_vars)))
os.environ.pop("NFISAM_WALK", None)
pr = cProfile.Profile(); pr.enable()
for _ in range(10): solver.sample_posterior()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14); print(s.getvalue()[:4000])
