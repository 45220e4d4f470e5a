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
# round 5, long horizon: every update's STRUCTURE against the reference (Manhattan-136, Plaza1, Plaza1-ADA-0.4: equality, so the
# report is a count) and the late posteriors of the complete Manhattan-136 run (updates 20 / 60 / 135)
if len(sys.argv) <= 2:
    import time
    out["long_horizon_structure"] = {}
    for case in T.STRUCTURE_CASES:
        path = os.path.join(T.GOLDEN, "pipeline_%s.npz" % case)
        if not os.path.exists(path):
            continue
        fx = np.load(path)
        t0 = time.time()
        fits, reuses, orderings, _ = T._run_structure(case, fx)
        ok = (len(fits) == int(fx["seed0_n_fits"]) and reuses == json.loads(str(fx["seed0_reuses"])) and
              all(orderings[i] == [str(v) for v in fx["seed0_step%d_ordering" % i]] for i in range(int(fx["seed0_n_steps"]))))
        out["long_horizon_structure"][case] = {"updates": int(fx["seed0_n_steps"]), "retrained_cliques": len(fits), "reused_roots": len(reuses),
                                               "widest_clique_D": max(f["D"] for f in fits), "equal_to_reference": bool(ok),
                                               "seconds_here_all_updates_20_iterations": round(time.time() - t0, 2),
                                               "seconds_reference_cpu_all_updates_20_iterations": round(float(np.sum(fx["seed0_timing"])), 1)}
        print(case, "structure:", out["long_horizon_structure"][case], flush=True)
    if os.path.exists(os.path.join(T.GOLDEN, "pipeline_manhattan136_full.npz")):
        rows, failures = T.compare_late()
        fx = np.load(os.path.join(T.GOLDEN, "pipeline_manhattan136_full.npz"))
        out["manhattan136_full_late_posteriors"] = {"arguments": json.loads(str(fx["arguments"])), "reference_seeds": len(fx["seeds"]), "updates": 136,
                                                    "compared_at_updates": list(T.LATE_STEPS), "rows": rows, "failures": failures,
                                                    "seconds_reference_cpu_whole_run": [round(float(np.sum(fx["seed%d_timing" % int(s)])), 1) for s in fx["seeds"]]}
        print("manhattan136_full late posteriors:", len(rows), "rows,", len(failures), "over their bound", flush=True)
    # round 6: Plaza1 through update 30 at the reference's own budget (six seeds), posteriors at updates 10 / 20 / 30
    if os.path.exists(os.path.join(T.GOLDEN, "pipeline_plaza1_late.npz")):
        from scipy.stats import mannwhitneyu
        rows, failures = T.compare_late("plaza1_late", seeds=(0, 1, 2), rmse_seeds=tuple(range(12)), data="Plaza1EFG", late_steps=T.PLAZA_LATE_STEPS)
        fx = np.load(os.path.join(T.GOLDEN, "pipeline_plaza1_late.npz"), allow_pickle=True)
        for r in rows:
            if r["kind"] == "trajectory-rmse":
                r["rank_sum_p_two_sided"] = float(mannwhitneyu(np.array(r["ours_per_seed"]), np.array(r["reference_per_seed"]), alternative="two-sided").pvalue)
        out["plaza1_late_posteriors"] = {"arguments": json.loads(str(fx["arguments"])), "reference_seeds": len(fx["seeds"]), "updates": int(fx["seed0_n_steps"]),
                                         "compared_at_updates": list(T.PLAZA_LATE_STEPS), "rows": rows, "failures": failures,
                                         "reference_provenance": [json.loads(str(fx["seed%d_provenance" % int(s)])) for s in fx["seeds"] if ("seed%d_provenance" % int(s)) in fx.files],
                                         "seconds_reference_cpu_whole_run": [round(float(np.sum(fx["seed%d_timing" % int(s)])), 1) for s in fx["seeds"]]}
        print("plaza1_late posteriors:", len(rows), "rows,", len(failures), "over their bound", flush=True)
json.dump(out, open(sys.argv[1] if len(sys.argv) > 1 else "pipeline_report.json", "w"), indent=1)
