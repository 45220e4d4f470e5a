#!/usr/bin/env python3
"""bench.py — flow-training throughput of the NF-iSAM hot path on MI355X.

Metric (BASELINE.json): flow-training samples/sec/GPU := n x iterations / wall time of the
training loop, timed where the reference times it (src/slam/NFiSAM.py:435,451-492).

Headline workload at N=1 = BASELINE config[2] ("C3"), the largest single-GPU configuration:
8 independent cliques of the range-only SLAM family, D = 6 8 8 10 10 12 12 12 columns
[range obs | landmarks xy | poses x y theta], n = 2000 particles each, drawn from the cliques' own generative
model, normalised as NFiSAM.normalize_training_samples does, one NSF_AR layer (flow_number = 1, the
reference's default), K = 9 bins, H = 8, B = 5, Adam lr = 0.01, fixed number of iterations (no early stop),
trained as ONE batched launch sequence (grid.y = clique).  A "step" is ONE full-batch training iteration of
all 8 cliques (forward + analytic backward + gradient reduction + Adam).  With N > 1 every rank trains its
own 8 cliques (weak scaling, no data-path collective: independent cliques never exchange data).
With N > 1 a second, UNTIMED-for-`value` regime runs behind the headline and is reported as `exchange`: one incremental
update of a branching factor graph through `slam.ParallelNFiSAM` -- a binary "meeting tree" of 2^ceil(log2 N) robots, whose
Bayes tree has one leaf arm per rank and ceil-log2 levels of joins that cross ranks -- i.e. the path north_star describes
("RCCL over xGMI carrying separator samples between parent/child cliques"): cross-rank tree edges, bytes, this rank's
point-to-point and all-gather wall clock, and the update's wall clock on every rank.

Launch: `python bench.py --gpus N --steps K --warmup W`.  Under torchrun (WORLD_SIZE set) this process is one
rank; without it and N > 1 the script starts N child ranks itself BEFORE touching the GPU and relays rank 0's
line.

Prints ONE JSON line on rank 0 (driver contract), including
  roofline     : dominant kernel (the gradient kernel of the C3 batch) algorithmic FLOP / its average launch
                 duration, measured live with HIP events, against the fp32 peak of gfx950
                 (157.3 TFLOP/s = f32 MFMA peak = f32 packed-VALU peak); the binding resource is VALU issue.
  regimes      : the same live measurement for the other regimes of the path: C2 (BASELINE config[1]: one
                 clique, 4 stacked layers), one Plaza1-shaped clique (n = 2000, D = 15: the latency regime of the
                 real datasets), a batch of 64 such cliques (the throughput regime / scaling shape), the last two with
                 hidden_dim 16 (the reference's parameter grids sweep it, src/slam/NFiSAM.py:589-609), and C2 with
                 hidden_dim 16 and 4 (the two-dims-per-wave kernel is instantiated for both: H = 4 since round 4, H = 16 --
                 one layer's panels resident -- since round 5; before that the generic kernel: 192 us per iteration).
  cpu_baseline : the oracle (PyTorch-eager CPU restatement of the reference path, validated against
                 the reference) timed on the host cores on a bounded sample of the same workload, next to the
                 TRUE reference's figures measured in the build container (profiles/history/r02_cpu_reference_vs_port.json).
"""
import argparse
import json
import math
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "nf-isam_amd"))
sys.path.insert(0, ROOT)

FP32_PEAK_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md: f32 MFMA == f32 vector peak
K, H, B = 9, 8, 5.0
LR = 0.01


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


# ---- workloads ---------------------------------------------------------------------------------------------
def c3_problem(seed0):
    """-> list of (x normalised [2000, D], reference-order blob)."""
    out = []
    for c, sh in enumerate(C3_SHAPES):
        s, circ = ring_clique(2000, *sh, np.random.RandomState(seed0 + c))
        x, _, _ = normalize(s, circ)
        out.append((x, init_blob_np(x.shape[1], K, H, 1, seed0 + c)))
    return out


REGIME_HIDDEN = {"plaza_clique_n2000_D15_H16": 16, "batch64_n2000_D15_H16": 16, "C2_single_clique_n4096_D6_L4_H16": 16,
                 "C2_single_clique_n4096_D6_L4_H04": 4}     # hidden_dim of a regime (default H)


def regime_problem(name, seed0):
    """-> (list of (x, blob), L)"""
    if name in REGIME_HIDDEN:                # the same cliques with hidden_dim 16 / 4 (the reference's grids sweep it: NFiSAM.py:589-609)
        h = REGIME_HIDDEN[name]
        base, L = regime_problem(name[:-4], seed0)
        return [(x, init_blob_np(x.shape[1], K, h, L, seed0 + c)) for c, (x, _) in enumerate(base)], L
    if name == "C2_single_clique_n4096_D6_L4":
        s, circ = c2_clique(4096, seed0)
        x, _, _ = normalize(s, circ)
        return [(x, init_blob_np(6, K, H, 4, seed0))], 4
    if name == "plaza_clique_n2000_D15":
        s, circ = ring_clique(2000, *PLAZA_SHAPE, np.random.RandomState(seed0))
        x, _, _ = normalize(s, circ)
        return [(x, init_blob_np(15, K, H, 1, seed0))], 1
    if name == "single_clique_n1000_D15":    # (round 6: a clique of <= 1024 particles takes the two-lanes-per-particle family, csrc/nsf_half.h)
        s, circ = ring_clique(1000, *PLAZA_SHAPE, np.random.RandomState(seed0))
        x, _, _ = normalize(s, circ)
        return [(x, init_blob_np(15, K, H, 1, seed0))], 1
    if name == "batch64_n2000_D15":
        out = []
        for c in range(64):
            s, circ = ring_clique(2000, *PLAZA_SHAPE, np.random.RandomState(seed0 + c))
            x, _, _ = normalize(s, circ)
            out.append((x, init_blob_np(15, K, H, 1, seed0 + c)))
        return out, 1
    raise KeyError(name)


class Workload:
    """Device-resident clique batch + the two timed things: whole iterations, and the gradient kernel alone."""

    def __init__(self, problem, L, dev, hidden=None):
        import torch
        import nfisam_hip as nh
        self.torch, self.nh, self.L, self.dev = torch, nh, L, dev
        self.H = H if hidden is None else hidden
        self.xs = [torch.from_numpy(x).to(dev) for x, _ in problem]
        self.kp0 = [nh.pack(torch.from_numpy(b).to(dev), x.shape[1], K, self.H, L) for x, b in problem]
        self.n_samples = sum(int(x.shape[0]) for x, _ in problem)
        self.flop_per_launch = sum(flops_per_sample_iter(x.shape[1], K, self.H, L) * x.shape[0] for x, _ in problem)

    def batch(self, iters):
        """`iters` fixed iterations: the window early-stop rule is armed with a tolerance of 0 (never fires) and a window
        of `iters`, so that the hipGraph chunk length divides `iters` and every iteration is a graph replay."""
        return self.nh.TrainBatch(self.xs, [p.clone() for p in self.kp0], K, self.H, B, self.L, lr=LR, max_iters=iters,
                                  average_window=iters, loss_delta_tol=0.0, early_stop=True)

    def time_iterations(self, iters, warmup, barrier, reduce_max=None):
        """`iters` training iterations of the prepared plan, timed R times (R from `iters` alone, so every rank agrees):
        each replay restarts from the initial parameters (re-initialised in place, untimed), starts behind a
        barrier + synchronize, ends at the rank's own synchronize (the closing barrier follows, outside the clock: with
        RCCL it is a collective of tens of microseconds that would be charged to a 0.3 ms region of an exchange-free job)
        and is reduced with MAX over ranks; the MEDIAN replay is reported, so that one
        slow graph launch does not decide a 0.4 ms measurement.  -> (median seconds, median GPU ms, first loss, final loss)"""
        torch = self.torch
        tbw = self.batch(max(warmup, 1))
        tbw.run(use_graph=True)
        tbw.close()
        tb = self.batch(iters)
        tb.prepare(use_graph=True)               # one-time graph capture, outside the timed region
        reps = int(min(200, max(5, math.ceil(0.06 / (iters * 25e-6)))))      # >= ~60 ms of timed work
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        dts, gms = [], []
        ev_reps = 3                              # extra replays that carry the two HIP events (GPU time of the region): the events'
        for r in range(reps + ev_reps):          # own stream packets and host calls stay out of the wall-clock replays
            with_events = r >= reps or os.environ.get("BENCH_EVENTS_IN_REPLAY") == "1"      # (=1: as up to round 5, every replay carries them: A/B)
            if r > 0:
                tb.reset(self.kp0)
            barrier()
            t0 = time.perf_counter()
            if with_events:
                ev0.record()
            done = tb.run(use_graph=True)
            if with_events:
                ev1.record()
            torch.cuda.synchronize()             # LOCAL: this rank's K steps are done
            dt = time.perf_counter() - t0
            barrier()                            # the closing barrier is an RCCL collective: outside dt, MAX over ranks below
            assert all(i == iters for i in done), done
            if with_events:
                gms.append(ev0.elapsed_time(ev1))
            if r < reps:
                dts.append(reduce_max(dt) if reduce_max is not None else dt)
        il = [t.cpu().numpy() for t in tb.iter_loss]
        for v in il:
            assert np.all(np.isfinite(v)) and (iters == 1 or v[iters - 1] < v[0]), (v[0], v[iters - 1])   # --steps 1: one loss value
        # which form the plan ran in: chunk-persistent launches (a chunk's iterations in ONE launch per chain) report the XCDs
        # their (clique, dim) groups sat on; 0 = one launch per iteration
        self.persistent = tb.xcd_span() > 0
        self.chunk_iters = max(c for c in range(1, min(iters, 128) + 1) if iters % c == 0)    # nsf_kernels.hip: chunk_length()
        tb.close()
        self.replays = reps
        return (float(np.median(dts)), float(np.median(gms)), float(np.mean([v[0] for v in il])),
                float(np.mean([v[iters - 1] for v in il])))

    def time_persistent_kernel(self, iters, reps=7):
        """Duration of the chunk-persistent launch(es) of ONE chunk of the same plan shape, by itself: a second plan whose
        chunk is two graphs (training launches | closing Adam + bookkeeping) with two HIP timing events recorded on the
        stream around the first (nfisam_nsf_train_plan_kernel_ms); median over `reps` runs of `iters` iterations, the last
        chunk of each.  -> ms per launch"""
        tb = self.batch(iters)
        tb.prepare(use_graph=True, timing=True)
        ms = []
        for r in range(reps + 1):
            if r > 0:
                tb.reset(self.kp0)
            tb.run(use_graph=True)
            self.torch.cuda.synchronize()
            if r > 0:                                # (the first run pages the kernels in)
                ms.append(tb.kernel_ms())
        tb.close()
        return float(np.median(ms))

    def time_gradient_kernel(self, reps=200):
        """Average duration of the gradient kernel as ONE launch over all (clique, dim) groups: `reps` launches captured in
        a graph (no host launch gaps), HIP events on the stream they run on; includes one ~1.5 us kernel boundary.  (A
        training plan may issue the same kernel as `chains` concurrent launches over disjoint groups -- parallel graph
        branches, nfisam_nsf_train_chains -- which shortens the ITERATION, `gpu_us_per_iteration_events`, not the kernel.)
        -> (us per launch, chains of the plan)"""
        torch = self.torch
        tbk = self.nh.TrainBatch(self.xs, [p.clone() for p in self.kp0], K, self.H, B, self.L, lr=LR, max_iters=10 ** 6,
                                 early_stop=False)
        chains = tbk.chains()
        for _ in range(20):
            tbk.gradient_only()
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for _ in range(reps):
                tbk.gradient_only()
        graph.replay()
        torch.cuda.synchronize()
        k0, k1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        k0.record()
        graph.replay()
        k1.record()
        torch.cuda.synchronize()
        us = 1e3 * k0.elapsed_time(k1) / reps
        tbk.close()
        return us, chains

    def record(self, iters, warmup, barrier, reduce_max=None):
        dt, gpu_ms, l0, l1 = self.time_iterations(iters, warmup, barrier, reduce_max)
        kus, chains = self.time_gradient_kernel()
        ach_plain = self.flop_per_launch / (kus * 1e-6) / 1e12
        gpu_us_it = 1e3 * gpu_ms / iters
        # The kernel of the timed region.  One launch per iteration: the gradient kernel timed alone (above).  Chunk-persistent
        # plan: ONE launch runs `chunk_iters` iterations; it is timed by itself right behind the
        # timed region, in a plan of the same shape that records two HIP timing events on the stream around the persistent
        # launch(es) of a chunk (time_persistent_kernel) -- as the plain kernel always was.
        if self.persistent:
            launch_us, launch_flop = 1e3 * self.time_persistent_kernel(iters), self.flop_per_launch * self.chunk_iters
        else:
            launch_us, launch_flop = kus, self.flop_per_launch
        ach = launch_flop / (launch_us * 1e-6) / 1e12
        return dict(cliques=len(self.xs), D=[int(x.shape[1]) for x in self.xs] if len(self.xs) <= 8 else int(self.xs[0].shape[1]),
                    particles_per_clique=int(self.xs[0].shape[0]), layers=self.L, hidden_dim=self.H, iterations=iters, replays=self.replays,
                    us_per_iteration=1e6 * dt / iters, gpu_us_per_iteration_events=1e3 * gpu_ms / iters,
                    samples_per_s=self.n_samples * iters / dt, gradient_kernel_us=kus,
                    # one launch per iteration: the plan issues an iteration as `chains` concurrent launches over disjoint groups;
                    # the chunk-persistent form is ONE launch per chunk over all groups
                    launches_per_training_iteration=(1.0 / self.chunk_iters) if self.persistent else chains,
                    flop_per_launch=self.flop_per_launch, achieved_tflops=ach, frac_of_fp32_peak=ach / FP32_PEAK_TFLOPS,
                    chunk_persistent=bool(self.persistent), iterations_per_launch=self.chunk_iters if self.persistent else 1,
                    training_launch_us=launch_us, training_launch_flop=launch_flop,
                    one_launch_per_iteration={"kernel_us": kus, "achieved_tflops": ach_plain, "frac_of_fp32_peak": ach_plain / FP32_PEAK_TFLOPS},
                    first_loss=l0, final_loss=l1), dt


def cpu_baseline(problem, budget_s=12.0):
    """Oracle timed on the host cores: bounded sample of the SAME workload (the eight C3 cliques, one after the other --
    the reference trains cliques sequentially).  The path is ~4k tiny eager ops per iteration, so more threads than ~8
    only add overhead: the thread count is capped at 8 and stated in `cores`."""
    import torch
    from oracle import nsf_torch as O
    cores = min(os.cpu_count() or 1, 8)
    torch.set_num_threads(cores)
    t_total, n_total = 0.0, 0
    per_clique = budget_s / len(problem)
    iters_used = []
    for x, blob in problem:
        xt, b0 = torch.from_numpy(x), torch.from_numpy(blob)
        O.train(xt, b0, K, H, B, 1, lr=LR, max_iters=1, early_stop=False)        # warm-up (first call pages torch in)
        t0 = time.perf_counter()
        O.train(xt, b0, K, H, B, 1, lr=LR, max_iters=2, early_stop=False)        # estimate
        per = (time.perf_counter() - t0) / 2
        iters = int(max(2, min(60, per_clique / max(per, 1e-6))))
        t0 = time.perf_counter()
        O.train(xt, b0, K, H, B, 1, lr=LR, max_iters=iters, early_stop=False)
        t_total += time.perf_counter() - t0
        n_total += x.shape[0] * iters
        iters_used.append(iters)
    out = {"value": n_total / t_total, "unit": "samples/s", "cores": cores, "kind": "port",
           "sample": "%s full-batch Adam iterations of the 8 C3 cliques (n=2000, D=6..12, L=1, K=9), one clique after the "
                     "other, with the PyTorch-eager CPU restatement of the reference path (oracle/nsf_torch.py), %d threads"
                     % ("/".join(str(i) for i in iters_used), cores)}
    ref = os.path.join(ROOT, "profiles", "history", "r02_cpu_reference_vs_port.json")
    if os.path.exists(ref):   # the TRUE reference cannot travel to the GPU box: its figures were taken in the build container
        try:
            r = json.load(open(ref))
            out["reference_measured"] = {
                "where": "build container (%s, %d threads, torch %s), scripts/cpu_reference_vs_port.py" %
                         (r["cpu"], r["threads"], r["torch"]),
                "c3_reference_samples_per_s": r["c3_batch"]["reference_samples_per_s"],
                "c3_port_samples_per_s": r["c3_batch"]["port_samples_per_s"],
                "port_over_reference": r["c3_batch"]["port_over_reference"],
                "note": "the port is ~1.5x FASTER than the reference it stands for (it batches the D spline evaluations "
                        "of a layer like the reference but skips the reference's per-dim torch.split/.clone/.cuda() "
                        "scratch tensors), so speed-ups quoted against `value` understate the speed-up over the reference",
            }
        except Exception as e:   # noqa: BLE001
            out["reference_measured"] = {"error": str(e)[:100]}
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


def meeting_tree(depth):
    """A factor graph whose Bayes tree (natural ordering) is a balanced binary tree of JOINS: 2^depth robots ("arms": prior
    pose B -odometry-> A -odometry-> X, a landmark M ranged from B and A) meet pairwise (a relative-pose factor between the
    two end poses, the first robot then drives on to a new pose), the pairs' survivors meet again, and so on.  Cliques
    (checked in tests/test_clique_parallel.py): per arm {B, M | A} <- {A | X}; per join {X_2j, X_2j+1 | Y_j} (root: no
    separator) with two child subtrees whose separators are DISJOINT -- the clique simulator, like the reference's, assumes
    the priors inside a clique do not overlap (src/sampler/SimulationBasedSampler.py:19).  `assign_subtrees` puts one arm
    on each of 2^depth ranks; every join then has one local and one remote child: 2^depth - 1 cross-rank edges.
    -> (variables in elimination order, factors)"""
    from factors.Factors import (SE2R2RangeGaussianLikelihoodFactor, SE2RelativeGaussianLikelihoodFactor,
                                 UnarySE2ApproximateGaussianPriorFactor)
    from geometry.TwoDimension import SE2Pose
    from slam.Variables import R2Variable, SE2Variable, VariableType
    cov = np.diag([0.3, 0.3, 0.05]) ** 2
    levels = [[] for _ in range(depth + 1)]
    leaf = {"B": [], "M": [], "A": []}
    factors = []

    def build(level, idx, y0):
        if level == 0:
            Bv, Av, Xv = SE2Variable("B%d" % idx), SE2Variable("A%d" % idx), SE2Variable("X%d" % idx)
            Mv = R2Variable("M%d" % idx, VariableType.Landmark)
            leaf["B"].append(Bv); leaf["M"].append(Mv); leaf["A"].append(Av); levels[0].append(Xv)
            factors.append(UnarySE2ApproximateGaussianPriorFactor(Bv, SE2Pose(0, y0, 0), cov))
            factors.append(SE2RelativeGaussianLikelihoodFactor(Bv, Av, SE2Pose(10, 0, 0), covariance=cov))
            factors.append(SE2RelativeGaussianLikelihoodFactor(Av, Xv, SE2Pose(10, 0, 0), covariance=cov))
            factors.append(SE2R2RangeGaussianLikelihoodFactor(Bv, Mv, 15.0, 0.5))
            factors.append(SE2R2RangeGaussianLikelihoodFactor(Av, Mv, 11.2, 0.5))
            return Xv, (20.0, y0)
        e1, (x1, y1) = build(level - 1, 2 * idx, y0)
        e2, (x2, y2) = build(level - 1, 2 * idx + 1, y0 + 40.0 * 2 ** (level - 1))
        factors.append(SE2RelativeGaussianLikelihoodFactor(e1, e2, SE2Pose(x2 - x1, y2 - y1, 0), covariance=cov))
        if level == depth:
            return None, None
        Yv = SE2Variable("%s%d" % ("YZWVUT"[level - 1], idx))
        levels[level].append(Yv)
        factors.append(SE2RelativeGaussianLikelihoodFactor(e1, Yv, SE2Pose(10, 0, 0), covariance=cov))
        return Yv, (x1 + 10.0, y1)

    build(depth, 0, 0.0)
    order = leaf["B"] + leaf["M"] + leaf["A"]
    for lv in levels[:depth]:
        order += lv
    return order, factors


def exchange_regime(world, rank, reps=2):
    """One incremental update of the meeting tree through `slam.ParallelNFiSAM` (the reference's dependency:
    src/slam/FactorGraphSolver.py:409-477 upward, :524-531 downward), `reps` times on fresh solvers (the first pays module
    load and kernel paging); the LAST is reported.  Every rank returns its own record; rank 0 gathers them."""
    import random
    import torch
    import torch.distributed as dist
    from slam.NFiSAM import NFiSAMArgs
    from slam.ParallelNFiSAM import ParallelNFiSAM
    depth = max(1, int(math.ceil(math.log2(max(world, 2)))))
    rec = None
    for rep in range(reps):
        random.seed(rep); np.random.seed(rep + 17 * rank); torch.manual_seed(rep + 17 * rank)
        order, factors = meeting_tree(depth)
        solver = ParallelNFiSAM(NFiSAMArgs(num_knots=K, flow_iterations=600, local_sample_num=2000, learning_rate=.02, hidden_dim=H,
                                           cuda_training=True, elimination_method="natural", training_set_frac=1.0,
                                           loss_delta_tol=.01, posterior_sample_num=500), posterior="sharded")
        for v in order:
            solver.add_node(v)
        for f in factors:
            solver.add_factor(f)
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        solver.update_physical_and_working_graphs()
        t1 = time.perf_counter()
        res = solver.incremental_inference()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        assert all(np.all(np.isfinite(res[v])) for v in order)
        up, down = solver.exchange_stats[-1], solver.posterior_exchange_stats
        rec = dict(rank=rank, update_ms=1e3 * (t2 - t0), graph_update_ms=1e3 * (t1 - t0), upward_pass_ms=up["upward_pass_ms"],
                   downward_pass_ms=down["downward_pass_ms"], cliques_trained_here=up["cliques_trained_here"],
                   p2p_ms=up["p2p_ms"] + down["p2p_ms"], p2p_send_ms=up["p2p_send_ms"] + down["p2p_send_ms"],
                   p2p_wait_ms=up["p2p_wait_ms"] + down["p2p_wait_ms"], all_gather_ms=up["all_gather_ms"] + down["all_gather_ms"],
                   cross_rank_edges=up["cross_rank_edges"] + down["cross_rank_edges"], bytes=up["bytes"] + down["bytes"],
                   upward=dict(up), downward=dict(down), cliques=len(solver.physical_bayes_tree.clique_ordering()),
                   owners=solver.owner_log[-1])
    # a raw link figure next to the pass's (which waits for producers): ping-pong of one separator batch between ranks 0 and 1
    ping_us = None
    if world > 1:
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
        buf = torch.zeros(2000, 3, dtype=torch.float32, device=dev)
        if rank in (0, 1):
            for it in range(25):
                if it == 5:
                    if dev.type == "cuda":
                        torch.cuda.synchronize()
                    tp = time.perf_counter()
                if rank == 0:
                    dist.send(buf, dst=1); dist.recv(buf, src=1)
                else:
                    dist.recv(buf, src=0); dist.send(buf, dst=0)
            if dev.type == "cuda":
                torch.cuda.synchronize()
            ping_us = 1e6 * (time.perf_counter() - tp) / 40.0           # one-way time of a [2000, 3] fp32 batch (24 KB)
    rec["p2p_one_way_us_24KB"] = ping_us
    gathered = [None] * world
    dist.all_gather_object(gathered, rec)
    return gathered


def replicas_regime(world, rank, use_dist, updates=20, replicas=8):
    """Chain-shaped configs ([3] Manhattan, [4] Plaza1: `pose_first` ordering, a 5-6-clique chain per update) do not shard by
    clique ("replicas only").  What N GPUs do for them is run independent problems: here EVERY rank runs `replicas`
    independent Plaza1 runs (seeds rank x replicas ..) over the first `updates` incremental updates through
    slam.ReplicaNFiSAM (their cliques in the slots of one batched training plan, a conveyor of chunks), and reports its
    wall-clock per replica-update.  No collective on the data path; the reference loops over its eight Plaza cases one after the
    other on one device (example/slam/plaza_dataset/run_nfisam.py:11-21).  The second half of the metric -- wall-clock per
    incremental update -- at N GPUs for the configs the exchange regime says nothing about."""
    import random
    import torch
    from slam.NFiSAM import NFiSAMArgs
    from slam.ReplicaNFiSAM import ReplicaNFiSAM
    from slam.RunBatch import graph_file_parser, group_nodes_factors_incrementally
    rec = dict(rank=rank)
    try:
        nodes, truth, factors = graph_file_parser(os.path.join(ROOT, "tests", "data", "Plaza1EFG", "factor_graph.fg"), "fg")
        steps = group_nodes_factors_incrementally(nodes, factors, incremental_step=5)[:updates]
        args = NFiSAMArgs(num_knots=9, flow_iterations=2000, local_sample_num=2000, learning_rate=.01, hidden_dim=8, cuda_training=True,
                          elimination_method="pose_first", training_set_frac=1.0, loss_delta_tol=.01, average_window=50)
        for rep_no in range(2):                  # (the first pass pays module load, plan creation and kernel paging)
            random.seed(rep_no); np.random.seed(1000 * rank + rep_no); torch.manual_seed(1000 * rank + rep_no)
            rep = ReplicaNFiSAM(args, [rank * replicas + r for r in range(replicas)])
            own = [[] for _ in range(replicas)]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            rep.run_incrementally(steps, on_update=lambda r, k, samples, seconds: own[r].append(seconds))
            torch.cuda.synchronize()
            total = time.perf_counter() - t0
        rec.update(total_s=total, per_replica_update_ms=1e3 * total / (replicas * len(steps)), replicas=replicas, updates=len(steps),
                   own_ms_per_update_median=1e3 * float(np.median(np.concatenate([np.array(o) for o in own]))),
                   fit_iterations=int(sum(rep.fit_iterations)))
    except Exception as e:   # noqa: BLE001  (must never break the contract line, nor leave the other ranks in the gather alone)
        rec["error"] = repr(e)[:300]
    if not use_dist:
        return [rec]
    import torch.distributed as dist
    gathered = [None] * world
    dist.all_gather_object(gathered, rec)
    return gathered


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N fresh child processes (one rank per GPU) BEFORE this
    process touches the GPU, relay rank 0's JSON line, exit with the worst child code."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0, _ = procs[0].communicate()
    codes = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out0.decode())
    sys.stdout.flush()
    return max(abs(c) for c in codes)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--regime-steps", type=int, default=200, help="iterations timed per secondary regime")
    ap.add_argument("--no-regimes", action="store_true", help="headline workload only (rocprofv3 runs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-exchange", action="store_true", help="N > 1: skip the ParallelNFiSAM exchange regime")
    ap.add_argument("--no-replicas", action="store_true", help="skip the replicas regime (8 Plaza1 runs per GPU, first 20 updates)")
    ap.add_argument("--no-update-bench", action="store_true",
                    help="skip the end-to-end incremental-update timing (used for rocprofv3 runs so that the kernel "
                         "statistics contain the headline workload only)")
    args = ap.parse_args()

    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        sys.exit(spawn_ranks(args))            # nothing above has touched the GPU
    world = int(env_world) if env_world is not None else 1
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d disagrees with WORLD_SIZE=%d" % (args.gpus, world))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    # one rank per GPU; BENCH_DIST_BACKEND=gloo lets several ranks share one GPU (smoke test of the launch path on a
    # 1-GPU box -- RCCL refuses two ranks on one device)
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    local_rank = local_rank % torch.cuda.device_count() if backend != "nccl" else local_rank
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # BENCH_FORCE_DIST=1: take the process-group branches at world size 1 too (the only way to execute the RCCL calls --
    # init with device_id, barrier, device all-reduce, teardown -- on a 1-GPU box; needs RANK/WORLD_SIZE/MASTER_* like torchrun sets)
    use_dist = world > 1 or os.environ.get("BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL prints a version banner on STDOUT when its first communicator comes up; the contract is ONE JSON line there,
        # so file descriptor 1 points at stderr until the group has done its first collective
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            if backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
            else:
                dist.init_process_group(backend, rank=rank, world_size=world)
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)
    red_dev = dev if backend == "nccl" else torch.device("cpu")

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()

    # ---- headline: C3, resident in HBM before the timed region ---------------------------------
    problem = c3_problem(seed0=100 + 1000 * rank)
    wl = Workload(problem, 1, dev)

    def reduce_max(v):                # every replay: the slowest rank's time
        if not use_dist:
            return v
        t = torch.tensor([v], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    head, dt = wl.record(args.steps, args.warmup, barrier, reduce_max)

    regimes = {}
    if rank == 0 and world == 1 and not args.no_regimes:
        for name in ("C2_single_clique_n4096_D6_L4", "plaza_clique_n2000_D15", "single_clique_n1000_D15", "batch64_n2000_D15", "plaza_clique_n2000_D15_H16",
                     "batch64_n2000_D15_H16", "C2_single_clique_n4096_D6_L4_H16", "C2_single_clique_n4096_D6_L4_H04"):
            prob, L = regime_problem(name, seed0=7)
            regimes[name], _ = Workload(prob, L, dev, REGIME_HIDDEN.get(name)).record(args.regime_steps, 20, lambda: torch.cuda.synchronize())

    # ---- every N: what N GPUs do for the chain-shaped configs (Plaza1, Manhattan): independent runs per GPU ("replicas only") ----
    replicas_block = None
    if not args.no_replicas and not args.no_update_bench:
        try:
            per = replicas_regime(world, rank, use_dist)
            good = [r for r in per if "error" not in r]
            replicas_block = {
                "workload": "per GPU: 8 independent Plaza1 runs (778-pose range-only dataset, seeds 8 x rank ..) over the first 20 incremental "
                            "updates (K=9, n=2000, <=2000 it, lr .01, window 50, tol .01) through slam.ReplicaNFiSAM; no collective",
                "world": world, "runs": sum(r.get("replicas", 0) for r in good),
                "per_replica_update_ms_per_rank": [round(r["per_replica_update_ms"], 3) if "error" not in r else None for r in per],
                "per_replica_update_ms": max(r["per_replica_update_ms"] for r in good) if good else None,
                "replica_updates_per_s": sum(r["replicas"] * r["updates"] / r["total_s"] for r in good) if good else None,
                "total_s_per_rank": [round(r["total_s"], 3) if "error" not in r else None for r in per],
                "fit_iterations_per_rank": [r.get("fit_iterations") for r in per],
                "errors": [r["error"] for r in per if "error" in r] or None,
                "note": "per_replica_update_ms = the slowest rank's wall clock / (8 replicas x 20 updates); replica_updates_per_s = whole job. "
                        "Not part of `value`."}
        except Exception as e:   # noqa: BLE001  (must never break the contract line)
            replicas_block = {"error": repr(e)[:300]}

    # ---- N > 1 (or the forced process group of a 1-GPU box): the path that DOES exchange -- ParallelNFiSAM on a branching tree ----
    exchange = None
    hard_exit = False
    if use_dist and not args.no_exchange:
        # The regime has NEVER run with more than one RCCL rank (no multi-GPU node was available to this build: DESIGN.md 7); a
        # mismatched point-to-point pair would be a hang, and a hang here would take the headline line -- already measured -- with
        # it.  So it runs in a thread under a watchdog (BENCH_EXCHANGE_TIMEOUT seconds, default 240): if it does not come back,
        # `exchange` says so, the line is printed, and the rank leaves with os._exit (no barrier, no teardown: they would hang too).
        import threading
        box = {}

        def _run_exchange():
            try:
                torch.cuda.set_device(local_rank)
                box["per_rank"] = exchange_regime(world, rank)
            except Exception as e:   # noqa: BLE001  (must never break the contract line)
                box["error"] = repr(e)[:300]
        th = threading.Thread(target=_run_exchange, daemon=True)
        th.start()
        th.join(float(os.environ.get("BENCH_EXCHANGE_TIMEOUT", "240")))
        if th.is_alive():
            hard_exit = True
            box["error"] = "the exchange regime did not return within its watchdog's time on rank %d" % rank
        elif "error" in box:
            hard_exit = True              # (the other ranks may sit in a collective this rank never entered: do not join their barrier)
    if "per_rank" in (box if use_dist and not args.no_exchange else {}):
        try:
            per_rank = box["per_rank"]
            r0 = per_rank[0]
            exchange = {
                "workload": "one incremental update of a binary meeting tree of %d robots (%d cliques, D = 6..9+obs, n = 2000, K=9, H=8, "
                            "<= 600 iterations + window stop, 500 posterior samples) through slam.ParallelNFiSAM: upward pass sharded by "
                            "subtree, child -> parent separator batches [2000, Ds] fp32 point-to-point on cross-rank tree edges, models "
                            "replicated by two all_gathers, downward pass sharded with parent -> child batches [500, Ds]" % (2 ** max(1, int(math.ceil(math.log2(max(world, 2))))), r0["cliques"]),
                "backend": backend + (" (RCCL)" if backend == "nccl" else ""), "world": world,
                "cross_rank_edges": r0["cross_rank_edges"], "bytes": r0["bytes"],
                "p2p_ms": max(r["p2p_ms"] for r in per_rank), "all_gather_ms": max(r["all_gather_ms"] for r in per_rank),
                "p2p_send_ms_max": max(r["p2p_send_ms"] for r in per_rank), "p2p_wait_ms_max": max(r["p2p_wait_ms"] for r in per_rank),
                "p2p_one_way_us_24KB": per_rank[min(1, world - 1)]["p2p_one_way_us_24KB"] if world > 1 else None,
                "update_ms_per_rank": [round(r["update_ms"], 3) for r in per_rank], "update_ms": max(r["update_ms"] for r in per_rank),
                "cliques_trained_per_rank": [r["cliques_trained_here"] for r in per_rank],
                "per_rank": [{k: r[k] for k in ("rank", "update_ms", "graph_update_ms", "upward_pass_ms", "downward_pass_ms", "p2p_ms",
                                                "p2p_send_ms", "p2p_wait_ms", "all_gather_ms", "cliques_trained_here")} for r in per_rank],
                "upward": {k: r0["upward"][k] for k in ("cross_rank_edges", "bytes", "all_gather_bytes")},
                "downward": {k: r0["downward"][k] for k in ("cross_rank_edges", "bytes", "all_gather_bytes")},
                "note": "p2p_ms / all_gather_ms = the slowest rank's wall clock inside send + recv + drain / inside the two all_gathers; "
                        "p2p_wait_ms contains the PRODUCER'S remaining work (a parent waits for its remote child's fit), not link time -- "
                        "p2p_one_way_us_24KB is the link's own figure (ping-pong of one [2000, 3] fp32 batch between ranks 0 and 1).  "
                        "Not part of `value`: the headline stays the exchange-free weak-scaling line."}
        except Exception as e:   # noqa: BLE001  (must never break the contract line)
            exchange = {"error": repr(e)[:300]}
    elif use_dist and not args.no_exchange:
        exchange = {"error": box.get("error", "no result")}

    if rank == 0:
        total = world * wl.n_samples * args.steps
        ach = head["achieved_tflops"]
        out = {
            "metric": "flow-training samples/sec/GPU + wall-clock per incremental update",
            "value": total / dt,
            "unit": "samples/s (n x training iterations / s, whole job)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps,
            "timing": "median of %d replays of the %d-step plan (each bracketed by barrier + synchronize, MAX over ranks)" % (wl.replays, args.steps),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "C3 (BASELINE config[2]): 8 independent cliques of the range-only SLAM family per GPU, "
                                   "D = 6 8 8 10 10 12 12 12, n = 2000 each, NSF_AR x1 layer, K=9, H=8, B=5, Adam lr=0.01, "
                                   "fixed iterations, one batched launch sequence",
                       "cliques_per_gpu": len(problem), "particles": 2000, "D": head["D"], "layers": 1, "K": K, "H": H,
                       "hipgraph": True, "parallelism": "independent cliques per GPU (no collective)"},
            "per_gpu_value": wl.n_samples * args.steps / dt,
            "gpu_ms_per_step_events": head["gpu_us_per_iteration_events"] / 1e3,
            "final_loss": head["final_loss"], "first_loss": head["first_loss"],
            "roofline": {"bound": "valu_issue", "achieved": ach, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": ach / FP32_PEAK_TFLOPS, "traffic": None,
                         "kernel": "nsf_train1_kernel<9,8,true>" if head["chunk_persistent"] else "nsf_train1_kernel<9,8>",
                         "kernel_us": head["training_launch_us"], "flop_per_launch": head["training_launch_flop"],
                         "iterations_per_launch": head["iterations_per_launch"],
                         "launches_per_training_iteration": head["launches_per_training_iteration"],
                         "achieved_in_training": head["flop_per_launch"] / (head["gpu_us_per_iteration_events"] * 1e-6) / 1e12,
                         "one_launch_per_iteration": head["one_launch_per_iteration"],
                         "note": "fp32 INSTRUCTION-ISSUE-bound kernel (SURVEY.md §8d: ~600 flop/B, HBM does not bind), priced "
                                 "against the fp32 peak (157.3 TFLOP/s = f32 MFMA = packed f32 VALU).  `kernel` is the kernel the "
                                 "TIMED REGION ran.  Chunk-persistent form (nsf_train1_kernel<9,8,true>, DESIGN.md §3.1e/f): one "
                                 "launch (over all (clique, dim) groups) runs `iterations_per_launch` training iterations (gradient + the "
                                 "previous iteration's Adam update; the blocks of a (clique, dim) group exchange their gradient "
                                 "copies as tagged words, no kernel boundary); `flop_per_launch` = 333 MFLOP x iterations per launch, "
                                 "`kernel_us` = the duration of that launch between "
                                 "two HIP timing events recorded on its stream, measured right behind the timed region in a plan of "
                                 "the same shape whose chunk end is a graph of its own (rocprofv3's per-launch average for the same "
                                 "command: profiles/; `achieved_in_training` divides by the timed region's own GPU time per "
                                 "iteration, closing Adam and bookkeeping kernels included).  One launch per iteration "
                                 "(nsf_train1_kernel<9,8>): `kernel_us` times the gradient kernel as ONE launch over all groups, "
                                 "200 back to back in a graph (`one_launch_per_iteration` holds that figure in either case).  Per "
                                 "(dim, 64-particle tile) unit a wave issues ~560 VALU instructions (~450 of them the spline, 64 "
                                 "transcendentals), 164 v_mfma_f32_4x4x1 (the conditioner mat-vecs, particle on the lane), 48-64 "
                                 "v_mfma_f32_16x16x4 (the weight-gradient GEMMs) and ~220 LDS instructions; f32 MFMA and VALU issue of "
                                 "a SIMD do not overlap on gfx950 (profiles/history/r02_mfma_valu_issue_microbench.txt).  C3 is 2496 "
                                 "(dim, tile) units on 1024 SIMDs: per iteration and wave ~12 k cycles of arithmetic and ~15 k of "
                                 "staging (the group's 8 gradient copies -> Adam -> panel), epilogue (block sum, copy out) and "
                                 "barriers at three waves per SIMD (profiles/r04_phase_cycles_persistent.txt) -- a latency-plus-issue "
                                 "regime; the 64-clique batch (`regimes.batch64_n2000_D15`) is the throughput regime at ~95 % of what "
                                 "its instruction mix allows (DESIGN.md §3.1c).  Algorithmic HBM bytes per iteration: 608 KB (x, "
                                 "read once per CHUNK by the persistent form: the tile stays in LDS) + 112 KB (parameters); "
                                 "`traffic` = HBM-side bytes per launch from separate rocprofv3 --pmc passes of an earlier run of "
                                 "this workload (lower bound of the gfx950 FETCH_SIZE range, see `traffic_profiled`), not measured "
                                 "in this run."},
            "regimes": regimes,
            "replicas": replicas_block,
            "exchange": exchange,
            # (round 6: nobody should have to dig for it -- the rank still leaves with exit code 0 so that the driver keeps the line)
            "exchange_failed": (None if exchange is None else ("error" in exchange)),
        }
        import glob
        tjs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_train_kernel_traffic.json")))
        if tjs:   # HBM bytes per launch from separate rocprofv3 --pmc passes of an EARLIER run of this workload (latest round)
            try:
                t = json.load(open(tjs[-1]))
                out["roofline"]["traffic_profiled"] = dict(t, source="profiles/" + os.path.basename(tjs[-1]))
                if t.get("kernel") == out["roofline"]["kernel"]:
                    per_it = t.get("hbm_side_bytes_per_iteration_lower", t.get("hbm_side_bytes_per_launch_lower"))
                    out["roofline"]["traffic"] = per_it * head["iterations_per_launch"]
                    out["roofline"]["traffic_provenance"] = ("rocprofv3 --pmc passes of scripts/collect_profiles.sh (%s, collected %s): lower bound of the "
                                                             "HBM-side bytes per ITERATION of %s (separate FETCH_SIZE / WRITE_SIZE passes) x the %d "
                                                             "iterations of a launch of this run; a counter figure of that collection, not of this run"
                                                             % (os.path.basename(tjs[-1]), t.get("profiled_at", "in an earlier round"), t.get("kernel"),
                                                                head["iterations_per_launch"]))
                    out["roofline"]["traffic_profiled_at"] = t.get("profiled_at")
            except Exception:   # noqa: BLE001
                pass
        ujs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_issue_utilisation.json")))
        uj = ujs[-1] if ujs else ""
        if os.path.exists(uj):   # VALU issue / MFMA busy fractions from the same EARLIER profiled run (not measured here)
            try:
                u = json.load(open(uj))
                out["roofline"]["issue_profiled"] = {k: {f: u[k][f] for f in ("valu_issue_frac", "mfma_busy_frac", "issue_frac")}
                                                     for k in ("C3", "batch64")}
                out["roofline"]["issue_profiled"]["source"] = "profiles/" + os.path.basename(uj)
            except Exception:   # noqa: BLE001
                pass
        if args.no_update_bench or world > 1:       # end-to-end update timing and CPU baseline: N = 1 only
            out["wall_clock_per_incremental_update"] = None
        else:
            try:
                out["wall_clock_per_incremental_update"] = incremental_update_wallclock()
            except Exception as e:   # noqa: BLE001  (must never break the contract line)
                out["wall_clock_per_incremental_update"] = {"error": str(e)[:200]}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(problem)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
        sys.stdout.flush()
    if hard_exit:                      # (the exchange regime hangs in a collective: so would the barrier and the teardown)
        sys.stderr.write("bench.py: rank %d: THE EXCHANGE REGIME FAILED (%s); the headline line above is unaffected, "
                         "`exchange_failed` is true in it; leaving without barrier or teardown\n" % (rank, box.get("error", "?")))
        sys.stderr.flush()
        os._exit(0)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
