"""Host-only: writes tests/golden/manhattan136_full_rmse_reference_seeds.json and rebuilds profiles/r05_manhattan136_late_rmse_ours_vs_reference.json's `reference_per_seed` and `summary` from the
per-pose error files (profiles/r05_manhattan136_pose_errors_*.npz; scripts/exp/pose_scatter.py --reduce makes the reference's).
    python scripts/exp/late_summary.py"""
import json, os
import numpy as np
from scipy.stats import mannwhitneyu
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
P = os.path.join(ROOT, "profiles")
path = os.path.join(P, "r05_manhattan136_late_rmse_ours_vs_reference.json")
d = json.load(open(path))
ref = np.load(os.path.join(P, "r05_manhattan136_pose_errors_reference.npz"))
ours = np.load(os.path.join(P, "r05_manhattan136_pose_errors_ours_n2000.npz"))
rm = lambda E: np.sqrt((E[:, :, :2].astype(np.float64) ** 2).sum(2).mean(1))
seeds = [int(s) for s in ref["seeds"]]
d["reference_per_seed"] = {str(s): {str(u): round(float(rm(ref["update%d" % u])[k]), 3) for u in (20, 60, 135)} for k, s in enumerate(seeds)}
summ = {}
for u in (20, 60, 135):
    a, b = rm(ours["update%d" % u]), rm(ref["update%d" % u])
    first, rest = b[:12], b[12:]
    summ["update_%d" % u] = {
        "ours_seeds": len(a), "ours_mean_pm_se": "%.2f +- %.2f" % (a.mean(), a.std(ddof=1) / np.sqrt(len(a))), "ours_median": round(float(np.median(a)), 2),
        "reference_seeds": len(b), "reference_mean_pm_se": "%.2f +- %.2f" % (b.mean(), b.std(ddof=1) / np.sqrt(len(b))), "reference_median": round(float(np.median(b)), 2),
        "rank_sum_one_sided_p_ours_greater": round(float(mannwhitneyu(a, b, alternative="greater").pvalue), 3),
        "runs_at_or_above_8.4_m": "%d of %d ours (%.0f %%), %d of %d reference (%.0f %%)" % ((a >= 8.4).sum(), len(a), 100 * (a >= 8.4).mean(), (b >= 8.4).sum(), len(b), 100 * (b >= 8.4).mean()),
        "reference_first_twelve_seeds_mean_pm_se": "%.2f +- %.2f" % (first.mean(), first.std(ddof=1) / np.sqrt(len(first))),
        "reference_later_seeds_mean_pm_se": ("%.2f +- %.2f (%d seeds)" % (rest.mean(), rest.std(ddof=1) / np.sqrt(len(rest)), len(rest))) if len(rest) > 1 else None}
d["summary"] = summ
json.dump(d, open(path, "w"), indent=1)
# the test fixture of tests/test_pipeline_gpu.py::test_late_trajectory_error_over_many_seeds_is_distributed_like_the_reference_s: DATA
# (numbers reduced from the reference's own runs), next to pipeline_manhattan136_full.npz which holds the first six seeds' samples
json.dump({"what": "trajectory RMSE (m) of the reference's complete Manhattan-136 runs at updates 20 / 60 / 135, one entry per seed: "
                   "tests/golden/make_pipeline_fixture.py --worker manhattan136_full <seed> out.npz, reduced by scripts/exp/pose_scatter.py --reduce "
                   "and scripts/exp/late_summary.py", "seeds": seeds,
           "rmse": {str(u): [round(float(v), 4) for v in rm(ref["update%d" % u])] for u in (20, 60, 135)}},
          open(os.path.join(ROOT, "tests", "golden", "manhattan136_full_rmse_reference_seeds.json"), "w"), indent=1)
for u, s in summ.items():
    print(u, s)
