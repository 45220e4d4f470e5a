"""Replica batching: R independent runs of one problem advanced in lock-step, their pending cliques
trained by ONE batched launch sequence, must reproduce the R sequential runs with the same seeds.  Both sides are forced
to the same training-kernel family (NFISAM_TRAIN=split), so the comparison is bit for bit: same simulated batches,
same parameters, same early-stop iteration, same posterior samples (the loss record itself is summed with float atomics and
agrees to the last bits only)."""
import os
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _steps(tmp_path):
    from slam.RunBatch import graph_file_parser, group_nodes_factors_incrementally
    g = np.load(os.path.join(GOLDEN, "small_range_case1.npz"))
    p = tmp_path / "factor_graph.fg"
    p.write_text(str(g["factor_graph_fg"]))
    nodes, truth, factors = graph_file_parser(str(p), "fg")
    return group_nodes_factors_incrementally(nodes, factors, incremental_step=1)


def _args():
    from slam.NFiSAM import NFiSAMArgs
    return NFiSAMArgs(num_knots=9, flow_iterations=600, local_sample_num=2000, learning_rate=.025, hidden_dim=8,
                      cuda_training=True, elimination_method="pose_first", training_set_frac=1.0, loss_delta_tol=.01,
                      posterior_sample_num=300)


def test_replicas_reproduce_sequential_runs(tmp_path):
    from slam.NFiSAM import NFiSAM
    from slam.ReplicaNFiSAM import ReplicaNFiSAM
    steps = _steps(tmp_path)[:4]
    seeds = [11, 12, 13]
    old = os.environ.get("NFISAM_TRAIN")
    os.environ["NFISAM_TRAIN"] = "split"
    try:
        seq = []
        for s in seeds:
            random.seed(s); np.random.seed(s); torch.manual_seed(s)
            solver = NFiSAM(_args())
            per_step = []
            for vs, fs in steps:
                for v in vs:
                    solver.add_node(v)
                for f in fs:
                    solver.add_factor(f)
                solver.update_physical_and_working_graphs()
                res = solver.incremental_inference()
                per_step.append((np.hstack([res[v] for v in solver.elimination_ordering]), dict(solver._temp_training_loss)))
            seq.append(per_step)
        rep = ReplicaNFiSAM(_args(), seeds)
        for k, (vs, fs) in enumerate(steps):
            for v in vs:
                rep.add_node(v)
            for f in fs:
                rep.add_factor(f)
            out = rep.update()
            assert rep.last_batches == [len(seeds)]            # one clique per replica per update, trained together
            for r, solver in enumerate(rep.solvers):
                got = np.hstack([out[r][v] for v in solver.elimination_ordering])
                ref, ref_loss = seq[r][k]
                assert got.shape == ref.shape and np.all(np.isfinite(got))
                assert solver._temp_training_loss.keys() == ref_loss.keys()
                for name, l in solver._temp_training_loss.items():
                    # same stop iteration; the loss RECORD is summed through 64 atomic slots (last-bit differences), the
                    # gradient path is atomics-free, so parameters and therefore the samples below are bit-identical
                    assert np.count_nonzero(l) == np.count_nonzero(ref_loss[name])
                    np.testing.assert_allclose(np.array(l), np.array(ref_loss[name]), rtol=2e-6)
                np.testing.assert_array_equal(got, ref)                                      # same posterior samples
        # the replicas really are different runs
        a = np.hstack([out[0][v] for v in rep.solvers[0].elimination_ordering])
        b = np.hstack([out[1][v] for v in rep.solvers[1].elimination_ordering])
        assert not np.array_equal(a, b)
    finally:
        if old is None:
            os.environ.pop("NFISAM_TRAIN", None)
        else:
            os.environ["NFISAM_TRAIN"] = old


def test_replicas_with_default_kernels_match_in_distribution(tmp_path):
    """Default kernel selection: the batched launch takes the throughput kernel, a single run the latency kernel; the
    results then agree to kernel rounding, i.e. in distribution (MMD of the final posterior below the 1000-sample floor
    of SURVEY.md §4) and in the recorded first-iteration loss (identical batches, identical initial parameters)."""
    from slam.NFiSAM import NFiSAM
    from slam.ReplicaNFiSAM import ReplicaNFiSAM
    from utils.Statistics import MMDb
    steps = _steps(tmp_path)[:3]
    seeds = [21, 22]
    rep = ReplicaNFiSAM(_args(), seeds)
    first = None
    for vs, fs in steps:
        for v in vs:
            rep.add_node(v)
        for f in fs:
            rep.add_factor(f)
        out = rep.update()
        if first is None:
            first = [dict(s._temp_training_loss) for s in rep.solvers]
    for r, s in enumerate(seeds):
        random.seed(s); np.random.seed(s); torch.manual_seed(s)
        solver = NFiSAM(_args())
        for k, (vs, fs) in enumerate(steps):
            for v in vs:
                solver.add_node(v)
            for f in fs:
                solver.add_factor(f)
            solver.update_physical_and_working_graphs()
            res = solver.incremental_inference()
            if k == 0:      # first update: identical batch and initial parameters (later batches depend on the trained child)
                for name, l in solver._temp_training_loss.items():
                    assert abs(l[0] - first[r][name][0]) < 5e-4
        order = solver.elimination_ordering
        a = np.hstack([res[v][:, :2] for v in order])
        b = np.hstack([out[r][v][:, :2] for v in order])
        assert MMDb(a, b) < 0.12


def test_free_running_replicas_reproduce_sequential_runs(tmp_path):
    """`ReplicaNFiSAM.run_incrementally`: every replica goes through the incremental steps at its own pace (its cliques train in
    its slot of the shared plan while the others are anywhere in their own updates).  Each replica still reproduces the
    sequential run with its seed bit for bit, step by step."""
    from slam.NFiSAM import NFiSAM
    from slam.ReplicaNFiSAM import ReplicaNFiSAM
    steps = _steps(tmp_path)[:5]
    seeds = [31, 32, 33, 34]
    old = os.environ.get("NFISAM_TRAIN")
    os.environ["NFISAM_TRAIN"] = "split"
    try:
        seq = []
        for s in seeds:
            random.seed(s); np.random.seed(s); torch.manual_seed(s)
            solver = NFiSAM(_args())
            per_step = []
            for vs, fs in steps:
                for v in vs:
                    solver.add_node(v)
                for f in fs:
                    solver.add_factor(f)
                solver.update_physical_and_working_graphs()
                res = solver.incremental_inference()
                per_step.append(np.hstack([res[v] for v in solver.elimination_ordering]))
            seq.append(per_step)
        rep = ReplicaNFiSAM(_args(), seeds)
        seen = {}

        def on_update(r, k, samples, seconds):        # `samples`: in the elimination ordering of step k (replica r may be a step ahead by now)
            seen[(r, k)] = np.hstack(list(samples.values()))
        out = rep.run_incrementally(steps, on_update=on_update)
        assert sorted(seen) == [(r, k) for r in range(len(seeds)) for k in range(len(steps))]
        for r in range(len(seeds)):
            assert len(out[r]) == len(steps)
            for k in range(len(steps)):
                np.testing.assert_array_equal(seen[(r, k)], seq[r][k])
        assert not np.array_equal(seen[(0, len(steps) - 1)], seen[(1, len(steps) - 1)])     # the replicas really are different runs
    finally:
        if old is None:
            os.environ.pop("NFISAM_TRAIN", None)
        else:
            os.environ["NFISAM_TRAIN"] = old
