"""Diagnostic (GPU): the numbers behind tests/test_pipeline_gpu.py as JSON (python scripts/pipeline_report.py out.json [case ...]):
per dataset every row `compare_case` evaluates (clique fits, per-step joint / block-wise / marginal MMD: this repository's
runs (3 seeds) against the reference's seed band, with the bound) and the wall-clock per update next to the reference's CPU timing."""
import json, os, sys, tempfile, pathlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nf-isam_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_pipeline_gpu as T

out = {}
for case in (sys.argv[2:] or list(T.CASES)):
    fx = np.load(os.path.join(T.GOLDEN, "pipeline_%s.npz" % case))
    ref_seeds = [int(s) for s in fx["seeds"]]
    with tempfile.TemporaryDirectory() as td:
        rows, failures, timing = T.compare_case(pathlib.Path(td), case)
    out[case] = {"arguments": json.loads(str(fx["arguments"])), "reference_seeds": len(ref_seeds), "rows": rows, "failures": failures,
                 "seconds_per_update_here": np.round(np.median(np.array(timing), 0), 4).tolist(),
                 "seconds_per_update_reference_cpu": np.round(np.median(np.array([fx["seed%d_timing" % s] for s in ref_seeds]), 0), 2).tolist()}
    print(case, "done:", len(rows), "rows,", len(failures), "over their bound", flush=True)
json.dump(out, open(sys.argv[1] if len(sys.argv) > 1 else "pipeline_report.json", "w"), indent=1)
