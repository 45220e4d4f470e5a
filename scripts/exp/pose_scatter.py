"""Host-only: the open finding of round 5 pose by pose.  Inputs are per-seed posterior-mean errors of every pose of the complete
Manhattan-136 run at updates 20 / 60 / 135 ([seed, pose, (ex, ey, sd_x, sd_y)]): ours from `DUMP=... scripts/exp/late_rmse.py N`
(MI355X), the reference's from its own runs (workers of tests/golden/make_pipeline_fixture.py, reduced by --reduce below).
    python scripts/exp/pose_scatter.py profiles/r05_manhattan136_pose_errors_reference.npz profiles/r05_manhattan136_pose_errors_ours_n2000.npz [more of ours ...]
    python scripts/exp/pose_scatter.py --reduce out.npz <dir with manhattan136_full_seed*.npz> ...      (build container only)"""
import glob, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LATE = (20, 60, 135)


def reduce_reference(out, dirs):
    sys.path.insert(0, os.path.join(ROOT, "nf-isam_amd")); sys.path.insert(0, ROOT)
    from slam.RunBatch import graph_file_parser
    nodes, truth, _ = graph_file_parser(os.path.join(ROOT, "tests", "data", "ManhattanPlaza136", "factor_graph.fg"), "fg", prior_cov_scale=0.1)
    tr = {str(v.name): np.asarray(truth[v], dtype=float)[:2] for v in nodes}
    res, seeds = {i: [] for i in LATE}, []
    for d in dirs:
        for f in sorted(glob.glob(os.path.join(d, "manhattan136_full_seed*.npz")), key=lambda p: int(p.split("seed")[-1].split(".")[0])):
            z = np.load(f)
            seeds.append(int(f.split("seed")[-1].split(".")[0]))
            for i in LATE:
                A, off, row = z["step%d_samples" % i].astype(np.float64), 0, {}
                for n in [str(v) for v in z["step%d_ordering" % i]]:
                    if n.startswith("X"):
                        row[int(n[1:])] = np.concatenate([A[:, off:off + 2].mean(0) - tr[n], A[:, off:off + 2].std(0)])
                    off += 3 if n.startswith("X") else 2
                res[i].append(np.array([row[k] for k in sorted(row)]))
    np.savez_compressed(out, seeds=np.array(seeds), **{"update%d" % i: np.array(res[i], dtype=np.float32) for i in LATE})
    print("reference seeds", seeds)


def rmse(E):
    return np.sqrt((E[:, :, :2].astype(np.float64) ** 2).sum(2).mean(1))


def scatter(E):            # across-seed variance of the posterior mean, summed over poses and axes; and its robust (MAD) twin
    X = E[:, :, :2].astype(np.float64)
    return X.var(0, ddof=1).sum(), ((1.4826 * np.median(np.abs(X - np.median(X, 0)), 0)) ** 2).sum()


def main():
    if sys.argv[1] == "--reduce":
        return reduce_reference(sys.argv[2], sys.argv[3:])
    ref = np.load(sys.argv[1])
    rng = np.random.default_rng(0)
    for f in sys.argv[2:]:
        ours = np.load(f)
        print(os.path.basename(f))
        for u in LATE:
            R, O = ref["update%d" % u], ours["update%d" % u]
            (vr, mr), (vo, mo) = scatter(R), scatter(O)
            k = R.shape[0]
            sub = np.array([scatter(O[rng.choice(O.shape[0], k, replace=False)])[0] for _ in range(4000)]) if O.shape[0] > k else np.array([vo])
            ratio = np.sqrt(O[:, 1:, :2].astype(np.float64).var(0, ddof=1).sum(1) / R[:, 1:, :2].astype(np.float64).var(0, ddof=1).sum(1))
            print("  update %3d  rmse mean / median / max: ours %.2f / %.2f / %.1f (%d seeds)  reference %.2f / %.2f / %.1f (%d seeds)" % (
                u, rmse(O).mean(), np.median(rmse(O)), rmse(O).max(), O.shape[0], rmse(R).mean(), np.median(rmse(R)), rmse(R).max(), k))
            print("              posterior sd (mean over poses): ours %.2f reference %.2f | across-seed scatter of the posterior mean, total variance: "
                  "ours %.0f (robust %.0f) reference %.0f (robust %.0f); per-pose sd ratio median %.2f; P(%d of ours scatter <= reference) = %.3f" % (
                      O[:, :, 2:].mean(), R[:, :, 2:].mean(), vo, mo, vr, mr, np.median(ratio), k, float((sub <= vr).mean())))


if __name__ == "__main__":
    main()
