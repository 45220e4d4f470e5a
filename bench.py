#!/usr/bin/env python3
"""bench.py — flow-training throughput of the NF-iSAM hot path on MI355X.

Metric (BASELINE.json): flow-training samples/sec/GPU := n x iterations / wall time of the
training loop, timed where the reference times it (src/slam/NFiSAM.py:435,451-492).

Workload at N=1 = BASELINE config[1] ("C2"): one clique, n=4096 particles,
D=6 columns [range-obs | landmark xy | pose x y theta] drawn from the clique's own generative
model (ring-shaped posterior), normalised as NFiSAM.normalize_training_samples does, L=4 stacked
NSF_AR layers, K=9 bins, H=8, B=5, Adam lr=0.02, fixed number of iterations (no early stop).
A "step" is ONE full-batch training iteration (forward + analytic backward + gradient reduction
+ Adam).  With N>1 every rank trains its own independent clique of the same shape (weak scaling,
no data-path collective: independent cliques never exchange data).

Prints ONE JSON line on rank 0 (see the driver contract in the task description), including
  roofline     : dominant kernel (nsf_train2_kernel) algorithmic FLOP / its average launch
                 duration, measured live with HIP events, against the fp32 peak of gfx950
                 (157.3 TFLOP/s = f32 MFMA peak = f32 packed-VALU peak).
  cpu_baseline : the oracle (PyTorch-eager CPU restatement of the reference path, validated against
                 the reference) timed on the host cores on a bounded sample of the same workload.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "nf-isam_amd"))
sys.path.insert(0, ROOT)

FP32_PEAK_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md: f32 MFMA == f32 vector peak
K, H, B, L = 9, 8, 5.0, 4
N_PART, D = 4096, 6
LR = 0.02


def c2_clique(n, seed):
    """Synthetic single-clique batch of config C2: columns [obs | lx ly | px py ptheta]."""
    rng = np.random.RandomState(seed)
    v = rng.randn(n, 3) * np.array([0.02, 0.004, 0.002])            # pose prior in the tangent space
    w = v[:, 2]
    small = np.abs(w) < 1e-9
    ws = np.where(small, 1.0, w)
    a = np.where(small, 1.0, np.sin(ws) / ws)
    b = np.where(small, 0.0, (1 - np.cos(ws)) / ws)
    px = a * v[:, 0] - b * v[:, 1]
    py = b * v[:, 0] + a * v[:, 1]
    r = 42.4 + 2.0 * rng.randn(n)
    phi = rng.uniform(-np.pi, np.pi, n)
    lx, ly = px + r * np.cos(phi), py + r * np.sin(phi)
    obs = np.hypot(lx - px, ly - py) + 2.0 * rng.randn(n)
    s = np.stack([obs, lx, ly, px, py, w], 1)
    circular = [False, False, False, False, False, True]
    return s, circular


def ring_clique(n, n_lmk, n_pose, n_obs, rng):
    """Synthetic clique of the range-only SLAM family (configs C3 / Plaza shape): columns
    [obs (n_obs) | landmarks xy (n_lmk) | poses x y theta (n_pose)]; every pose is a prior pose pushed through odometry
    noise, landmark j sits on a ring around pose 0, observation k is a noisy range pose(k % n_pose) -> landmark
    (k % n_lmk).  D = n_obs + 2 n_lmk + 3 n_pose."""
    poses = []
    base = np.zeros((n, 3))
    for p in range(n_pose):
        v = rng.randn(n, 3) * np.array([0.2, 0.04, 0.02]) + np.array([20.0 * p, 0.0, 0.0])
        poses.append(base + v)
    lm = []
    for j in range(n_lmk):
        r = 42.4 + 10.0 * j + 2.0 * rng.randn(n)
        phi = rng.uniform(-np.pi, np.pi, n)
        lm.append(np.stack([poses[0][:, 0] + r * np.cos(phi), poses[0][:, 1] + r * np.sin(phi)], 1))
    obs = [np.hypot(lm[k % n_lmk][:, 0] - poses[k % n_pose][:, 0], lm[k % n_lmk][:, 1] - poses[k % n_pose][:, 1]) +
           2.0 * rng.randn(n) for k in range(n_obs)]
    s = np.concatenate([np.stack(obs, 1)] + lm + poses, 1)
    circ = [False] * (n_obs + 2 * n_lmk) + [False, False, True] * n_pose
    return s, circ


# (n_lmk, n_pose, n_obs) of BASELINE config[2] "C3": 8 cliques, D = 6 8 8 10 10 12 12 12, n = 2000 each
C3_SHAPES = [(1, 1, 1), (2, 1, 1), (2, 1, 1), (1, 2, 2), (1, 2, 2), (2, 2, 2), (2, 2, 2), (2, 2, 2)]
# Plaza1 / Manhattan clique shapes (D = 15, 16, 17)
PLAZA_SHAPE = (3, 2, 3)
SHAPE_OF_D = {15: (3, 2, 3), 16: (2, 3, 3), 17: (3, 3, 2)}


def normalize(samples, circular):
    """NFiSAM.normalize_training_samples (src/slam/NFiSAM.py:515-548), host side, not timed."""
    from scipy.stats import circmean
    s = np.array(samples, dtype=np.float64, copy=True)
    mean = np.zeros(s.shape[1]); std = np.zeros(s.shape[1])
    for c in range(s.shape[1]):
        if circular[c]:
            mean[c] = circmean(s[:, c], high=np.pi, low=-np.pi)
            s[:, c] = (s[:, c] - mean[c] + np.pi) % (2 * np.pi) - np.pi
        else:
            mean[c] = s[:, c].mean()
            s[:, c] -= mean[c]
        std[c] = s[:, c].std()
    std = np.clip(std, 1e-5, None)
    return (s / std).astype(np.float32), mean.astype(np.float32), std.astype(np.float32)


def flops_per_sample_iter(D, K, H, L):
    """forward = L*{2[H D(D-1)/2 + (D-1)(H^2 + H Po)] + (D-1)(2H+Po) + D(8K+40)};
    a training iteration = 3x forward (forward + backward)."""
    Po = 3 * K - 1
    fwd = L * (2 * (H * D * (D - 1) // 2 + (D - 1) * (H * H + H * Po)) + (D - 1) * (2 * H + Po) + D * (8 * K + 40))
    return 3 * fwd


def init_blob_np(D, K, H, L, seed):
    """Reference initialisation (flows.py:62-63 + torch nn.Linear default), numpy RNG."""
    rng = np.random.RandomState(seed)
    Po = 3 * K - 1
    parts = []
    for _ in range(L):
        parts.append(rng.uniform(-0.5, 0.5, Po))
        for i in range(1, D):
            for fan_in, cnt in ((i, H * i), (i, H), (H, H * H), (H, H), (H, Po * H), (H, Po)):
                bound = 1.0 / math.sqrt(fan_in)
                parts.append(rng.uniform(-bound, bound, cnt))
    return np.concatenate(parts).astype(np.float32)


def cpu_baseline(x, blob, budget_s=12.0):
    """Oracle timed on the host cores: bounded sample of the SAME workload (same batch, same model).
    The path is ~20k tiny eager ops per iteration, so more threads than ~8 only add overhead
    (the reference measured 5.2e4 samples/s on 8 threads for this shape): the
    thread count is capped at 8 and stated in `cores`."""
    import torch
    from oracle import nsf_torch as O
    cores = min(os.cpu_count() or 1, 8)
    torch.set_num_threads(cores)
    xt = torch.from_numpy(x)
    b0 = torch.from_numpy(blob)
    t0 = time.perf_counter()
    O.train(xt, b0, K, H, B, L, lr=LR, max_iters=1, early_stop=False)        # warm-up + estimate
    per = time.perf_counter() - t0
    iters = int(max(2, min(100, budget_s / max(per, 1e-6))))
    t0 = time.perf_counter()
    O.train(xt, b0, K, H, B, L, lr=LR, max_iters=iters, early_stop=False)
    dt = time.perf_counter() - t0
    out = {"value": x.shape[0] * iters / dt, "unit": "samples/s", "cores": cores, "kind": "port",
           "sample": "%d full-batch Adam iterations of the same clique (n=%d, D=%d, L=%d, K=%d) with the "
                     "PyTorch-eager CPU restatement of the reference path (oracle/nsf_torch.py), %d threads"
                     % (iters, x.shape[0], x.shape[1], L, K, cores),
           "ms_per_step": 1e3 * dt / iters}
    try:   # also the plain-C port (OpenMP over particles), for information
        os.environ["OMP_NUM_THREADS"] = str(min(os.cpu_count() or 1, 16))
        from oracle import c_oracle as CO
        CO.train(x, blob, K, H, B, L, lr=LR, max_iters=1, early_stop=False, dtype=np.float32)
        it2 = 10
        t0 = time.perf_counter()
        CO.train(x, blob, K, H, B, L, lr=LR, max_iters=it2, early_stop=False, dtype=np.float32)
        dt2 = time.perf_counter() - t0
        out["c_port_value"] = x.shape[0] * it2 / dt2
        out["c_port_threads"] = int(os.environ["OMP_NUM_THREADS"])
    except Exception as e:   # noqa: BLE001
        out["c_port_value"] = None
        out["c_port_error"] = str(e)[:100]
    return out


def incremental_update_wallclock():
    """Second half of the metric: wall-clock per incremental update, timed as the reference does
    (around update_physical_and_working_graphs + incremental_inference, FactorGraphSolver.py:803-808) on
    BASELINE config[0] (small_range_gaussian_problem/journal_paper/case1, the reference's run_nfisam.py
    arguments), end to end through the drop-in solver.  Runs AFTER the timed training region."""
    import random
    import tempfile
    import torch
    from slam.NFiSAM import NFiSAM, NFiSAMArgs
    from slam.RunBatch import graph_file_parser, group_nodes_factors_incrementally
    g = np.load(os.path.join(ROOT, "tests", "golden", "small_range_case1.npz"))
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "factor_graph.fg")
        open(path, "w").write(str(g["factor_graph_fg"]))
        nodes, truth, factors = graph_file_parser(path, "fg")
    steps = group_nodes_factors_incrementally(nodes, factors, incremental_step=1)
    out = []
    for rep in range(2):                      # rep 0 warms up (module load, graph instantiation)
        random.seed(rep); np.random.seed(rep); torch.manual_seed(rep)
        solver = NFiSAM(NFiSAMArgs(num_knots=9, flow_iterations=2000, local_sample_num=2000, learning_rate=.025,
                                   hidden_dim=8, cuda_training=True, elimination_method="pose_first",
                                   training_set_frac=1.0, loss_delta_tol=.01, posterior_sample_num=1000))
        ms = []
        for vs, fs in steps:
            for v in vs:
                solver.add_node(v)
            for f in fs:
                solver.add_factor(f)
            t0 = time.perf_counter()
            solver.update_physical_and_working_graphs()
            solver.incremental_inference()
            ms.append(1e3 * (time.perf_counter() - t0))
        out = ms
    ref = [float(t) for t in g["run1_step_timing"]]
    return {"workload": "config[0]: small_range_gaussian_problem journal_paper/case1, 6 incremental updates, "
                        "K=9 n=2000 <=2000 it lr .025 window 50 tol .01, 1000 posterior samples",
            "ms_per_update": [round(v, 3) for v in out], "mean_ms": float(np.mean(out)),
            "reference_stored_gpu_run_s": ref,
            "note": "reference column = example/.../case1/run1/step_timing (authors' GPU, same arguments)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--cliques", type=int, default=1, help="independent cliques per GPU (default: config C2 = 1)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-update-bench", action="store_true",
                    help="skip the end-to-end incremental-update timing (used for rocprofv3 runs so that the kernel "
                         "statistics contain the C2 workload only)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import nfisam_hip as nh

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    # ---- synthetic workload, resident in HBM before the timed region -----------------------
    nc = args.cliques
    xs_np, blobs_np = [], []
    for c in range(nc):
        s, circ = c2_clique(N_PART, seed=1000 * rank + c)
        xn, _, _ = normalize(s, circ)
        xs_np.append(xn)
        blobs_np.append(init_blob_np(D, K, H, L, seed=7 + 1000 * rank + c))
    xs = [torch.from_numpy(x).to(dev) for x in xs_np]
    kp0 = [nh.pack(torch.from_numpy(b).to(dev), D, K, H, L) for b in blobs_np]
    use_graph = not args.no_graph

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # warm-up: W untimed iterations (also instantiates kernels / graphs)
    tbw = nh.TrainBatch(xs, [p.clone() for p in kp0], K, H, B, L, lr=LR, max_iters=max(args.warmup, 1),
                        early_stop=False)
    tbw.run(use_graph=use_graph)
    tbw.close()
    # timed: exactly K iterations
    tb = nh.TrainBatch(xs, [p.clone() for p in kp0], K, H, B, L, lr=LR, max_iters=args.steps, early_stop=False)
    tb.prepare(use_graph=use_graph)          # one-time graph capture, outside the timed region
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    barrier()
    t0 = time.perf_counter()
    ev0.record()
    iters = tb.run(use_graph=use_graph)
    ev1.record()
    barrier()
    dt = time.perf_counter() - t0
    assert all(i == args.steps for i in iters), iters
    il = tb.iter_loss[0].cpu().numpy()
    assert np.all(np.isfinite(il)) and il[-1] < il[0], (il[0], il[-1])
    gpu_ms = ev0.elapsed_time(ev1)

    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- dominant kernel: average launch duration, HIP events on the launch stream ---------
    reps = 200
    tbk = nh.TrainBatch(xs[:1], [kp0[0].clone()], K, H, B, L, lr=LR, max_iters=10 ** 6, early_stop=False)

    def train_kernel_once():   # the gradient kernel exactly as the timed region launches it (per-tile slabs)
        tbk.gradient_only()
    for _ in range(20):
        train_kernel_once()
    torch.cuda.synchronize()
    # `reps` launches of the same kernel captured in a graph (no host launch gaps between them), timed
    # with HIP events on the stream they run on: per-launch duration + one ~1.5 us kernel boundary.
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _ in range(reps):
            train_kernel_once()
    graph.replay()
    torch.cuda.synchronize()
    k0, k1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    k0.record()
    graph.replay()
    k1.record()
    torch.cuda.synchronize()
    kern_us = 1e3 * k0.elapsed_time(k1) / reps

    fl = flops_per_sample_iter(D, K, H, L)
    launch_flops = fl * N_PART
    achieved = launch_flops / (kern_us * 1e-6) / 1e12

    if rank == 0:
        total = world * nc * N_PART * args.steps
        out = {
            "metric": "flow-training samples/sec/GPU + wall-clock per incremental update",
            "value": total / dt,
            "unit": "samples/s (n x training iterations / s, whole job)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "C2 single-clique 6-D ring posterior [obs|landmark xy|pose xyθ], n=4096, "
                                   "NSF_AR x4 layers, K=9, H=8, B=5, Adam lr=0.02, fixed iterations",
                       "cliques_per_gpu": nc, "particles": N_PART, "D": D, "layers": L, "K": K, "H": H,
                       "hipgraph": use_graph, "parallelism": "independent cliques per GPU (no collective)"},
            "per_gpu_value": nc * N_PART * args.steps / dt,
            "gpu_ms_per_step_events": gpu_ms / args.steps,
            "final_loss": float(il[-1]), "first_loss": float(il[0]),
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP32_PEAK_TFLOPS, "traffic": None,
                         "kernel": "nsf_train2_kernel<9,8,true>", "kernel_us": kern_us,
                         "flop_per_launch": launch_flops,
                         "note": "fp32 VALU/transcendental-issue-bound kernel, latency-bound on this single-clique "
                                 "workload (189 MFLOP per launch = 1.2 us at peak; 128 tiles x 6 waves on 128 of "
                                 "256 CUs, 7 dependent unit passes per wave); priced against the fp32 peak "
                                 "(157.3 TFLOP/s = f32 MFMA = packed f32 VALU). Algorithmic HBM bytes: 98 KB (x) + "
                                 "110 KB parameters per launch; `traffic` (measured, profiles/) is larger because "
                                 "every 32-particle tile writes its own 27 KB copy of the gradient with plain stores "
                                 "(3.5 MB per launch, read once by the Adam kernel) instead of float atomics - no "
                                 "re-reads, 130 GB/s, far below the HBM bound."},
        }
        if args.no_update_bench or world > 1:       # end-to-end update timing and CPU baseline: N = 1 only
            out["wall_clock_per_incremental_update"] = None
        else:
            try:
                out["wall_clock_per_incremental_update"] = incremental_update_wallclock()
            except Exception as e:   # noqa: BLE001  (must never break the contract line)
                out["wall_clock_per_incremental_update"] = {"error": str(e)[:200]}
        vj = os.path.join(ROOT, "profiles", "r01_valu_issue_utilisation.json")
        if os.path.exists(vj):   # the binding roofline of this path is VALU issue: counters from separate --pmc passes
            try:
                v = json.load(open(vj))
                out["roofline"]["valu_issue_utilisation"] = {"single_clique_C2": v["single_clique_C2"]["utilisation"],
                                                             "batch_64_cliques": v["batch_64_cliques_n2000_D15"]["utilisation"],
                                                             "source": "profiles/r01_valu_issue_utilisation.json"}
            except Exception:   # noqa: BLE001
                pass
        tj = os.path.join(ROOT, "profiles", "r01_train_kernel_traffic.json")
        if os.path.exists(tj):   # HBM bytes per launch from a separate rocprofv3 --pmc pass (profiles/README.md)
            try:
                out["roofline"]["traffic"] = json.load(open(tj))["hbm_bytes_per_launch"]
            except Exception:   # noqa: BLE001
                pass
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(xs_np[0], blobs_np[0])
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    tb.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
