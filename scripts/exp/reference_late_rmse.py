"""Host-only: trajectory RMSE (posterior mean of pose xy vs the .fg ground truth) of REFERENCE runs of the complete Manhattan-136
problem at updates 20 / 60 / 135, from worker outputs of tests/golden/make_pipeline_fixture.py (manhattan136_full_seed*.npz).
    python scripts/exp/reference_late_rmse.py <dir with manhattan136_full_seed*.npz> ..."""
import glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "nf-isam_amd")); sys.path.insert(0, ROOT)
import numpy as np
from slam.RunBatch import graph_file_parser
nodes, truth, factors = graph_file_parser(os.path.join(ROOT, "tests", "data", "ManhattanPlaza136", "factor_graph.fg"), "fg", prior_cov_scale=0.1)
tr = {str(v.name): np.asarray(truth[v], dtype=float)[:2] for v in nodes}
out = {}
for d in sys.argv[1:]:
    for f in sorted(glob.glob(os.path.join(d, "manhattan136_full_seed*.npz"))):
        z = np.load(f)
        seed = int(os.path.basename(f).split("seed")[1].split(".")[0])
        row = {}
        for i in (20, 60, 135):
            order = [str(v) for v in z["step%d_ordering" % i]]
            A = z["step%d_samples" % i].astype(np.float64)
            off, err = 0, []
            for n in order:
                w = 3 if n.startswith("X") else 2
                if n.startswith("X"):
                    err.append(A[:, off:off + 2].mean(0) - tr[n])
                off += w
            row[i] = float(np.sqrt((np.array(err) ** 2).sum(1).mean()))
        out[seed] = row
for i in (20, 60, 135):
    v = [out[s][i] for s in sorted(out)]
    print("update", i, "reference seeds", sorted(out), [round(x, 2) for x in v], "median %.2f" % np.median(v))
print(json.dumps(out))
