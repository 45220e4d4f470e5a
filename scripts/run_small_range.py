"""End-to-end run of BASELINE config[0] (small_range_gaussian_problem, case1) through the MI355X back end;
prints per-step timing and MMDb against the reference's stored NF-iSAM run and nested-sampling posteriors."""
import os, sys, tempfile, time, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nf-isam_amd")); sys.path.insert(0, ROOT)
from slam.NFiSAM import NFiSAM_empirial_study
from utils.Statistics import MMDb

g = np.load(os.path.join(ROOT, "tests", "golden", "small_range_case1.npz"))
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0


def xy(order, arr):
    cols, off = {}, 0
    for v in order:
        d = 3 if v.startswith("X") else 2
        cols[v] = arr[:, off:off + 2]
        off += d
    return np.hstack([cols[v] for v in sorted(order)])


with tempfile.TemporaryDirectory() as td:
    open(os.path.join(td, "factor_graph.fg"), "w").write(str(g["factor_graph_fg"]))
    import random
    random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)
    t0 = time.time()
    run_dirs = NFiSAM_empirial_study([9], [2000], [2000], [.025], [8], td, "factor_graph.fg", "fg", incremental_step=1,
                                     cuda_training=True, elimination_method="pose_first", training_set_frac=1.0,
                                     loss_delta_tol=.01, posterior_sample_num=1000)
    wall = time.time() - t0
    rd = run_dirs[0]
    st = [float(t) for t in open(os.path.join(rd, "step_timing")).read().split()]
    ft = [float(t) for t in open(os.path.join(rd, "fitting_timer")).read().split()]
    pt = [float(t) for t in open(os.path.join(rd, "posterior_sampling_timer")).read().split()]
    print("total wall %.2f s" % wall)
    for i in range(6):
        ours = np.loadtxt(os.path.join(rd, "step%d" % i))
        order = open(os.path.join(rd, "step%d_ordering" % i)).read().split()
        loss = json.load(open(os.path.join(rd, "step%d_step_training_loss" % i)))
        iters = [int(np.count_nonzero(v)) for v in loss.values()]
        m_run = MMDb(xy(order, ours), xy(str(g["run1_step%d_ordering" % i]).split(), g["run1_step%d" % i]))
        line = "step %d: update %.3f s (fit %.3f, posterior %.4f) iters %s | ref GPU run: %.2f s | MMDb vs ref-NF %.3f" % (
            i, st[i], ft[i], pt[i], iters, g["run1_step_timing"][i], m_run)
        if i < 4:
            m_dyn = MMDb(xy(order, ours), xy(str(g["dyn1_step%d_ordering" % i]).split(), g["dyn1_step%d" % i]))
            r_dyn = MMDb(xy(str(g["run1_step%d_ordering" % i]).split(), g["run1_step%d" % i]),
                         xy(str(g["dyn1_step%d_ordering" % i]).split(), g["dyn1_step%d" % i]))
            line += " | vs nested %.3f (ref-NF vs nested %.3f)" % (m_dyn, r_dyn)
        print(line)
    if len(sys.argv) > 2:
        for i in (3, 4, 5):
            ours = np.loadtxt(os.path.join(rd, "step%d" % i))
            order = open(os.path.join(rd, "step%d_ordering" % i)).read().split()
            ref = g["run1_step%d" % i]
            off = 0
            print("--- step", i)
            for v in order:
                d = 3 if v.startswith("X") else 2
                o, r = ours[:, off:off + 2], ref[:, off:off + 2]
                print("  %s ours mean %s std %s | ref mean %s std %s" % (v, np.round(o.mean(0), 1), np.round(o.std(0), 1),
                                                                      np.round(r.mean(0), 1), np.round(r.std(0), 1)))
                off += d
