"""Diagnostic (GPU): the numbers behind tests/test_pipeline_gpu.py as JSON (python scripts/pipeline_report.py out.json):
per dataset, per clique fit / per step: median MMD of this repository's run (3 seeds) to the reference's 5 seeds next to
the reference's own median pairwise MMD, and the wall-clock per update next to the reference's CPU timing."""
import json, os, sys, tempfile, pathlib, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nf-isam_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_pipeline_gpu as T

out = {}
for case in T.CASES:
    fx = np.load(os.path.join(T.GOLDEN, "pipeline_%s.npz" % case))
    ref_seeds = [int(s) for s in fx["seeds"]]
    rows = {"fit": {}, "step": {}}
    timing = []
    for seed in range(3):
        with tempfile.TemporaryDirectory() as td:
            run_dir, fits, n_steps = T._run(pathlib.Path(td), case, fx, seed)
            timing.append([float(t) for t in open(os.path.join(run_dir, "step_timing")).read().split()])
            import json as _j
            for j, f in enumerate(fits):
                meta = _j.loads(str(fx["seed0_fit%d_meta" % j]))
                obs_names = [v for v in meta["vars"] if v.startswith("O")]
                dims = f["dims"] if f["vars"] == meta["vars"] else [1] * len(obs_names) + f["dims"]
                refs = [T._xy_block(meta["vars"], meta["dims"], fx["seed%d_fit%d_batch" % (s, j)]) for s in ref_seeds]
                scale = np.maximum(np.vstack(refs).std(0), 1e-3)
                r = np.random.RandomState(j).permutation(f["batch"].shape[0])[:refs[0].shape[0]]
                m, spread = T._band(T._xy_block(meta["vars"], dims, f["batch"][r]) / scale, [x / scale for x in refs])
                rows["fit"].setdefault(j, {"vars": meta["vars"], "reference_spread": round(spread, 4), "ours_to_reference": []})["ours_to_reference"].append(round(m, 4))
            for i in range(n_steps):
                order = open(os.path.join(run_dir, "step%d_ordering" % i)).read().split()
                S = np.loadtxt(os.path.join(run_dir, "step%d" % i))
                dims = [3 if v.startswith("X") else 2 for v in order]
                refs = [T._xy_block(order, dims, fx["seed%d_step%d_samples" % (s, i)]) for s in ref_seeds]
                r = np.random.RandomState(100 + i).permutation(S.shape[0])[:refs[0].shape[0]]
                m, spread = T._band(T._xy_block(order, dims, S[r]), refs)
                rows["step"].setdefault(i, {"reference_spread": round(spread, 4), "ours_to_reference": []})["ours_to_reference"].append(round(m, 4))
    out[case] = {"arguments": json.loads(str(fx["arguments"])), "fits": rows["fit"], "steps": rows["step"],
                 "seconds_per_update_here": np.round(np.median(np.array(timing), 0), 4).tolist(),
                 "seconds_per_update_reference_cpu": np.round(np.median(np.array([fx["seed%d_timing" % s] for s in ref_seeds]), 0), 2).tolist()}
    print(case, "done", flush=True)
json.dump(out, open(sys.argv[1] if len(sys.argv) > 1 else "pipeline_report.json", "w"), indent=1)
