"""Multi-rank solver path on the GPU: `ParallelNFiSAM` with TWO child processes running the REAL
clique fit on a branching Bayes tree (`elimination_method="natural"`), compared with the single-process `NFiSAM` run of
the same problem by MMD of the posterior samples.  The GPU box has one device, so the two ranks share it and talk over
gloo (host tensors); on a multi-GPU node the same code runs one rank per GPU over RCCL.

Reference order preserved: children's separator samples -> fit -> separator factor for the parent
(src/slam/FactorGraphSolver.py:436-470); posterior root -> leaves (FactorGraphSolver.py:497-550, 524-531)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import json, os, random, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(%(root)r, "nf-isam_amd")); sys.path.insert(0, %(root)r)
import torch.distributed as dist
from slam.NFiSAM import NFiSAM, NFiSAMArgs
from slam.Variables import R2Variable, SE2Variable, VariableType
from geometry.TwoDimension import SE2Pose
from factors.Factors import (SE2R2RangeGaussianLikelihoodFactor, SE2RelativeGaussianLikelihoodFactor,
                             UnarySE2ApproximateGaussianPriorFactor)

mode, out = sys.argv[1], sys.argv[2]
rank = int(os.environ.get("RANK", "0"))
if mode != "single":
    if os.environ.get("PAR_BACKEND") == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=int(os.environ["WORLD_SIZE"]), device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=int(os.environ["WORLD_SIZE"]))
random.seed(3); np.random.seed(3 + rank); torch.manual_seed(3 + rank)

# Two robots (arms) that meet: arm a = B_a (prior) -odom-> A_a -odom-> X_a, a landmark M_a ranged from B_a and A_a, and
# an odometry-type constraint between the meeting poses X_1, X_2.  Natural ordering B1 B2 M1 M2 A1 A2 X1 X2 gives
#   root {X2, X1}  <-  {A1 | X1} <- {M1, B1 | A1}      and      {A2 | X2} <- {M2, B2 | A2}
# i.e. two sibling subtrees with DISJOINT separators ({X1} and {X2}: the clique simulator, like the reference's,
# assumes the priors inside a clique do not overlap, src/sampler/SimulationBasedSampler.py:19).
cov = np.diag([0.3, 0.3, 0.05]) ** 2
B_ = [SE2Variable("B%%d" %% a) for a in (1, 2)]
A_ = [SE2Variable("A%%d" %% a) for a in (1, 2)]
X_ = [SE2Variable("X%%d" %% a) for a in (1, 2)]
M_ = [R2Variable("M%%d" %% a, VariableType.Landmark) for a in (1, 2)]
factors0 = []
for a in range(2):
    y0 = 40.0 * a
    factors0.append(UnarySE2ApproximateGaussianPriorFactor(B_[a], SE2Pose(0, y0, 0), cov))
    factors0.append(SE2RelativeGaussianLikelihoodFactor(B_[a], A_[a], SE2Pose(10, 0, 0), covariance=cov))
    factors0.append(SE2RelativeGaussianLikelihoodFactor(A_[a], X_[a], SE2Pose(10, 0, 0), covariance=cov))
    factors0.append(SE2R2RangeGaussianLikelihoodFactor(B_[a], M_[a], 15.0, 0.5))
    factors0.append(SE2R2RangeGaussianLikelihoodFactor(A_[a], M_[a], 11.2, 0.5))
factors0.append(SE2RelativeGaussianLikelihoodFactor(X_[0], X_[1], SE2Pose(0, 40, 0), covariance=cov))
args = NFiSAMArgs(num_knots=9, flow_iterations=600, local_sample_num=2000, learning_rate=.02, hidden_dim=8,
                  cuda_training=True, elimination_method="natural", training_set_frac=1.0, loss_delta_tol=.01,
                  posterior_sample_num=800)
if mode == "single":
    solver = NFiSAM(args)
else:
    from slam.ParallelNFiSAM import ParallelNFiSAM
    solver = ParallelNFiSAM(args, posterior=mode)
order = B_ + M_ + A_ + X_
for v in order:
    solver.add_node(v)
for f in factors0:
    solver.add_factor(f)
solver.update_physical_and_working_graphs()
res = solver.incremental_inference()
tree = solver.physical_bayes_tree
info = dict(n_cliques=len(tree.clique_ordering()), max_children=max(len(c.children) for c in tree.clique_ordering()),
            owners=getattr(solver, "owner_log", [None])[-1] if hasattr(solver, "owner_log") else None,
            trained_here=sorted(solver._temp_training_loss.keys()))
np.savez(out, info=json.dumps(info), **{str(v.name): res[v] for v in order})
if mode != "single":
    dist.barrier()
    dist.destroy_process_group()
'''


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _mmd(a, b, sigma):
    def k(x, y):
        d = ((x[:, None, :] - y[None, :, :]) ** 2).sum(-1)
        return np.exp(-d / (2 * sigma ** 2))
    return float(np.sqrt(max(k(a, a).mean() + k(b, b).mean() - 2 * k(a, b).mean(), 0.0)))


def _run(tmp_path, mode, world, backend="gloo"):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % dict(root=ROOT))
    port = _free_port()
    procs, outs = [], []
    for r in range(world):
        out = str(tmp_path / ("%s_rank%d.npz" % (mode, r)))
        outs.append(out)
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   PAR_BACKEND=backend)
        procs.append(subprocess.Popen([sys.executable, str(script), mode, out], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT))
    logs = [p.communicate(timeout=600)[0].decode() for p in procs]
    for p, log in zip(procs, logs):
        assert p.returncode == 0, log[-3000:]
    return [dict(np.load(o)) for o in outs]


@pytest.mark.timeout(900)
@pytest.mark.parametrize("posterior", ["replicated", "sharded"])
def test_two_rank_solver_matches_single_process(tmp_path, posterior):
    single = _run(tmp_path, "single", 1)[0]
    ranks = _run(tmp_path, posterior, 2)
    info = [json.loads(str(r["info"])) for r in ranks]
    # the tree branches and both ranks trained something: the upward pass really was sharded
    assert info[0]["n_cliques"] == 5 and info[0]["max_children"] == 2, info[0]
    owners = info[0]["owners"]
    assert owners == info[1]["owners"] and set(owners.values()) == {0, 1}, owners
    names = [k for k in single if k != "info"]
    # replication: both ranks end with the same posterior samples (shared seed / gathered blocks)
    for v in names:
        assert ranks[0][v].shape == single[v].shape == (800, 2 if v[0] == "M" else 3)
        np.testing.assert_allclose(ranks[0][v], ranks[1][v], atol=1e-5)
        assert np.all(np.isfinite(ranks[0][v]))
    # same posterior as the single-process solver: MMD (RBF, sigma = sqrt(dim) x scale of the problem) on xy, per
    # variable, against the sample-vs-sample floor of two independent single-process draws (~0.05)
    for v in names:
        a, b = ranks[0][v][:, :2], single[v][:, :2]
        scale = max(1.0, float(b.std(0).max()))
        m = _mmd(a / scale, b / scale, np.sqrt(2.0))
        assert m < 0.16, (v, m)          # two separately trained posteriors: 0.05-0.13 seen across kernel revisions; a wrong separator message gives > 0.5
        assert np.linalg.norm(a.mean(0) - b.mean(0)) < 0.5 + 0.25 * scale, (v, a.mean(0), b.mean(0))


@pytest.mark.timeout(600)
def test_bench_two_ranks_prints_the_contract_line():
    """`python bench.py --gpus 2` (no launcher: bench.py starts the ranks itself before touching the GPU).  The box has
    one GPU, so the ranks share it over gloo (`BENCH_DIST_BACKEND=gloo`); on an 8-GPU node the same path runs over RCCL.
    The LAST stdout line is the driver's JSON line: whole-job value = 2 x per-rank value, n_gpus = 2, weak scaling."""
    env = dict(os.environ, BENCH_DIST_BACKEND="gloo")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5"],
                       env=env, capture_output=True, text=True, timeout=550)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.strip().split("\n") if l.strip()]
    d = json.loads(lines[-1])
    assert d["n_gpus"] == 2 and d["steps"] == 20 and d["warmup"] == 5 and d["scaling"] == "weak"
    assert d["metric"].startswith("flow-training samples/sec") and d["dtype"] == "f32" and d["vs_baseline"] is None
    assert abs(d["value"] - 2 * d["per_gpu_value"]) < 1e-6 * d["value"]
    assert d["value"] > 1e7 and 0 < d["ms_per_step"] < 5.0
    assert d["roofline"]["frac"] > 0 and d["cpu_baseline"] is None
    # the regime that DOES exchange (slam.ParallelNFiSAM on the two-robot meeting tree: 5 cliques, one join): one child ->
    # parent separator batch [2000, 3] upward and one parent -> child batch [500, 3] downward cross the two ranks
    x = d["exchange"]
    assert "error" not in x, x
    assert x["world"] == 2 and x["upward"]["cross_rank_edges"] == 1 and x["downward"]["cross_rank_edges"] == 1
    assert x["upward"]["bytes"] == 2000 * 3 * 4 and x["downward"]["bytes"] == 500 * 3 * 4 and x["bytes"] == 2500 * 3 * 4
    assert len(x["update_ms_per_rank"]) == 2 and all(0 < t < 60000 for t in x["update_ms_per_rank"])
    assert sum(x["cliques_trained_per_rank"]) == 5 and min(x["cliques_trained_per_rank"]) >= 2
    assert x["p2p_ms"] > 0 and x["all_gather_ms"] > 0 and x["p2p_one_way_us_24KB"] > 0
    assert d["exchange_failed"] is False
    # the regime for the chain-shaped configs (Plaza1 / Manhattan: "replicas only", round 6): every rank runs 8 independent Plaza1
    # runs over the first 20 updates; wall-clock per replica-update per rank and for the job
    rp = d["replicas"]
    assert "error" not in rp and rp["errors"] is None, rp
    assert rp["world"] == 2 and rp["runs"] == 16 and len(rp["per_replica_update_ms_per_rank"]) == 2
    assert all(0 < t < 2000 for t in rp["per_replica_update_ms_per_rank"]) and abs(rp["per_replica_update_ms"] - max(rp["per_replica_update_ms_per_rank"])) < 1e-3
    assert rp["replica_updates_per_s"] > 1 and all(i > 8 * 20 * 100 for i in rp["fit_iterations_per_rank"])


@pytest.mark.timeout(600)
@pytest.mark.parametrize("posterior", ["replicated", "sharded"])
def test_one_rank_over_rccl_matches_single_process(tmp_path, posterior):
    """The "nccl" (= RCCL) branches of `ParallelNFiSAM` -- device tensors in every collective, no host staging -- executed
    for real: the box has one GPU, so the group has ONE rank (RCCL refuses two ranks on a device); every broadcast /
    gather of the solver still goes through RCCL.  Same comparison as the two-rank gloo test."""
    single = _run(tmp_path, "single", 1)[0]
    one = _run(tmp_path, posterior, 1, backend="nccl")[0]
    info = json.loads(str(one["info"]))
    assert info["n_cliques"] == 5 and set(info["owners"].values()) == {0}, info
    for v in (k for k in single if k != "info"):
        a, b = one[v][:, :2], single[v][:, :2]
        assert one[v].shape == single[v].shape and np.all(np.isfinite(one[v]))
        scale = max(1.0, float(b.std(0).max()))
        assert _mmd(a / scale, b / scale, np.sqrt(2.0)) < 0.16, v
        assert np.linalg.norm(a.mean(0) - b.mean(0)) < 0.5 + 0.25 * scale, (v, a.mean(0), b.mean(0))


@pytest.mark.timeout(600)
def test_bench_under_the_launcher_over_rccl_prints_only_the_contract_line():
    """The driver's launch for N > 1 (`python -m torch.distributed.run ... bench.py --gpus N`), with the one rank a 1-GPU box
    allows: `BENCH_FORCE_DIST=1` makes bench.py take its process-group branches at world size 1, so `init_process_group("nccl",
    device_id=...)`, the barriers, the device all-reduce of the timings and the teardown run over RCCL.  RCCL's version banner
    must not reach stdout: the ONLY stdout line is the JSON record."""
    env = dict(os.environ, BENCH_FORCE_DIST="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "BENCH_DIST_BACKEND"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
                        "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1",
                        "--steps", "20", "--warmup", "5", "--no-regimes", "--no-update-bench", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=550)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.split("\n") if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["value"] > 1e7
    # the exchange regime over an RCCL group of ONE rank: every all_gather / broadcast of ParallelNFiSAM runs, no edge crosses
    x = d["exchange"]
    assert "error" not in x, x
    assert x["backend"].startswith("nccl") and x["world"] == 1 and x["cross_rank_edges"] == 0 and x["bytes"] == 0
    assert x["cliques_trained_per_rank"] == [5] and x["all_gather_ms"] > 0
    # The closing barrier of a timed replay is an RCCL collective (tens of microseconds): it must sit OUTSIDE the clock -- the
    # same 20-step plan timed without a process group, on this same box, within 10 % (round 3 had the barrier inside:
    # a 0.36 ms region of an exchange-free job would have read 10-25 % of fake scaling loss at N > 1).
    for k in ("BENCH_FORCE_DIST",):
        env.pop(k, None)
    q = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5", "--no-regimes",
                        "--no-update-bench", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=300)
    assert q.returncode == 0, q.stderr[-2000:]
    alone = json.loads([l for l in q.stdout.split("\n") if l.strip()][-1])
    assert abs(d["per_gpu_value"] / alone["per_gpu_value"] - 1.0) < 0.10, (d["per_gpu_value"], alone["per_gpu_value"])
    # (when RCCL prints its banner -- it does with this image's defaults -- it is in p.stderr, not in front of the record)


@pytest.mark.timeout(600)
def test_bench_line_survives_an_exchange_regime_that_does_not_come_back():
    """bench.py runs the `exchange` regime (ParallelNFiSAM over the process group: never executed with more than one RCCL rank)
    under a watchdog: when it does not return in time -- here: a watchdog of 10 ms against a regime that takes seconds -- the
    headline line is still printed, `exchange` says what happened, and the rank leaves with exit code 0 without entering another
    collective (a hang of the regime must not cost the driver its SCALE line)."""
    env = dict(os.environ, BENCH_FORCE_DIST="1", BENCH_EXCHANGE_TIMEOUT="0.01")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "BENCH_DIST_BACKEND"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
                        "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1",
                        "--steps", "20", "--warmup", "5", "--no-regimes", "--no-update-bench", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=550)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.split("\n") if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 1e7 and "did not return" in d["exchange"]["error"]
    assert d["exchange_failed"] is True and "EXCHANGE REGIME FAILED" in p.stderr


# ---- round 6: FOUR ranks (the one GPU shared over gloo), the real fit, the depth-2 meeting tree of bench.py -------------------------
WORKER4 = r'''
import json, os, random, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(%(root)r, "nf-isam_amd")); sys.path.insert(0, %(root)r)
import torch.distributed as dist
import bench
from slam.NFiSAM import NFiSAM, NFiSAMArgs

mode, out = sys.argv[1], sys.argv[2]
rank = int(os.environ.get("RANK", "0"))
if mode != "single":                      # (every rank is spawned before anything touches the GPU: the parent never initialises it)
    dist.init_process_group("gloo", rank=rank, world_size=int(os.environ["WORLD_SIZE"]))
random.seed(5); np.random.seed(5 + rank); torch.manual_seed(5 + rank)
order, factors = bench.meeting_tree(2)
args = NFiSAMArgs(num_knots=9, flow_iterations=600, local_sample_num=2000, learning_rate=.02, hidden_dim=8,
                  cuda_training=True, elimination_method="natural", training_set_frac=1.0, loss_delta_tol=.01,
                  posterior_sample_num=800)
if mode == "single":
    solver = NFiSAM(args)
else:
    from slam.ParallelNFiSAM import ParallelNFiSAM
    solver = ParallelNFiSAM(args, posterior=mode)
for v in order:
    solver.add_node(v)
for f in factors:
    solver.add_factor(f)
solver.update_physical_and_working_graphs()
res = solver.incremental_inference()
tree = solver.physical_bayes_tree
info = dict(n_cliques=len(tree.clique_ordering()), max_children=max(len(c.children) for c in tree.clique_ordering()),
            owners=solver.owner_log[-1] if hasattr(solver, "owner_log") else None,
            up=solver.exchange_stats[-1] if hasattr(solver, "exchange_stats") else None,
            down=getattr(solver, "posterior_exchange_stats", None),
            trained_here=sorted(solver._temp_training_loss.keys()))
np.savez(out, info=json.dumps(info), **{str(v.name): res[v] for v in order})
if mode != "single":
    dist.barrier()
    dist.destroy_process_group()
'''


@pytest.mark.timeout(1200)
def test_four_rank_solver_matches_single_process(tmp_path):
    """`ParallelNFiSAM` at world size FOUR with the real clique fit: four robots meet pairwise, the pairs'
    survivors meet again -- 11 cliques, three joins on two levels, every rank trains one arm, three child -> parent separator
    batches cross ranks upward and at least three parent -> child batches downward (`posterior="sharded"`), a rank talks to
    more than one peer and the replication all_gathers carry ragged lengths.  The box has one GPU: the four ranks share it
    over gloo (what stays unexecuted is the RCCL transport itself).  Compared with the single-process `NFiSAM` run by MMD."""
    def run(mode, world):
        script = tmp_path / "worker4.py"
        script.write_text(WORKER4 % dict(root=ROOT))
        port = _free_port()
        procs, outs = [], []
        for r in range(world):
            out = str(tmp_path / ("%s_rank%d.npz" % (mode, r)))
            outs.append(out)
            env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            procs.append(subprocess.Popen([sys.executable, str(script), mode, out], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
        logs = [p.communicate(timeout=1000)[0].decode() for p in procs]
        for p, log in zip(procs, logs):
            assert p.returncode == 0, log[-3000:]
        return [dict(np.load(o)) for o in outs]
    single = run("single", 1)[0]
    ranks = run("sharded", 4)
    info = [json.loads(str(r["info"])) for r in ranks]
    assert info[0]["n_cliques"] == 11 and info[0]["max_children"] == 2, info[0]
    owners = info[0]["owners"]
    assert all(i["owners"] == owners for i in info) and set(owners.values()) == {0, 1, 2, 3}, owners
    assert info[0]["up"]["cross_rank_edges"] == 3 and info[0]["down"]["cross_rank_edges"] >= 3, (info[0]["up"], info[0]["down"])
    assert sum(i["up"]["cliques_trained_here"] for i in info) == 11 and min(i["up"]["cliques_trained_here"] for i in info) >= 2
    names = [k for k in single if k != "info"]
    for v in names:
        for r in range(1, 4):                                       # every rank ends with the same samples of every variable
            np.testing.assert_allclose(ranks[0][v], ranks[r][v], atol=1e-5)
        assert ranks[0][v].shape == single[v].shape and np.all(np.isfinite(ranks[0][v]))
    for v in names:
        a, b = ranks[0][v][:, :2], single[v][:, :2]
        scale = max(1.0, float(b.std(0).max()))
        m = _mmd(a / scale, b / scale, np.sqrt(2.0))
        assert m < 0.16, (v, m)
        assert np.linalg.norm(a.mean(0) - b.mean(0)) < 0.5 + 0.25 * scale, (v, a.mean(0), b.mean(0))
