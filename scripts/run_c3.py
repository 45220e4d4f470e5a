"""BASELINE config 3 (SURVEY.md §8d "C3"): a batch of 8 independent cliques, D in {6,8,8,10,10,12,12,12}, n = 2000 each,
drawn from the ring family of config C2 (1-2 range constraints each), K = 9, H = 8, L = 1, Adam lr 0.01, 500 fixed
iterations, trained as ONE batched launch sequence (grid.y = clique).  Prints one JSON line.
Also runs the scaling-shape batch of §8d (64 cliques of the Plaza shape, n = 2000, D = 15, 300 iterations)."""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nf-isam_amd")); sys.path.insert(0, ROOT)
import nfisam_hip as nh
import bench as BM

K, H, B, L = 9, 8, 5.0, 1
dev = torch.device("cuda:0")


ring_clique = BM.ring_clique


def train_batch(shapes, n, iters, seed0):
    xs, kps = [], []
    for c, (n_lmk, n_pose, n_obs) in enumerate(shapes):
        s, circ = ring_clique(n, n_lmk, n_pose, n_obs, np.random.RandomState(seed0 + c))
        x, _, _ = BM.normalize(s, circ)
        D = x.shape[1]
        xs.append(torch.from_numpy(x).to(dev))
        kps.append(nh.pack(torch.from_numpy(BM.init_blob_np(D, K, H, L, seed0 + c)).to(dev), D, K, H, L))
    tb = nh.TrainBatch(xs, kps, K, H, B, L, lr=0.01, max_iters=iters, early_stop=False)
    tb.prepare(use_graph=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    done = tb.run(use_graph=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert done == [iters] * len(shapes), done
    Ds = [int(x.shape[1]) for x in xs]
    fl = sum(BM.flops_per_sample_iter(D, K, H, L) * n for D in Ds)
    il = [t.cpu().numpy() for t in tb.iter_loss]
    return dict(cliques=len(shapes), D=Ds, n=n, iterations=iters, seconds=dt, us_per_iteration=1e6 * dt / iters,
                samples_per_s=len(shapes) * n * iters / dt, tflops=fl * iters / dt / 1e12,
                first_loss=[float(v[0]) for v in il], final_loss=[float(v[iters - 1]) for v in il])


if __name__ == "__main__":
    c3 = BM.C3_SHAPES   # D = 6 8 8 10 10 12 12 12
    only = sys.argv[1] if len(sys.argv) > 1 else ""          # "c3" | "scaling" | "c2" | "" (the first two)
    out = {}
    if only in ("", "c3"):
        out["C3"] = train_batch(c3, 2000, 500, 100)
    plaza_shape = [(3, 2, 3)] * 64          # D = 3 + 6 + 6 = 15
    if only in ("", "scaling"):
        out["scaling_shape_64x_n2000_D15"] = train_batch(plaza_shape, 2000, 300, 200)
    if only == "c2":                        # BASELINE config[1]: one clique, D = 6, n = 4096, four layers (profiling target)
        L = 4
        prob, _ = BM.regime_problem("C2_single_clique_n4096_D6_L4", seed0=7)
        w = BM.Workload(prob, L, dev)
        r, _ = w.record(300, 20, torch.cuda.synchronize)
        out["C2"] = r
    print(json.dumps(out))
