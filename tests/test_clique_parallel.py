"""Multi-process (gloo, world_size 2, CPU) tests of the clique-sharding layer: deterministic
subtree assignment, leaves-first execution, point-to-point routing of separator batches across
rank boundaries, and the weak-scaling shard layout used by bench.py.  The per-clique `fit` is a
stand-in (the real one needs a GPU); what is under test is scheduling + message routing."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from slam.CliqueParallel import CliqueTree, assign_subtrees, run_tree, shard_independent


def bushy_tree():
    #            root
    #        /         \
    #       a           b
    #     /   \       /   \
    #    a1   a2     b1    b2
    #                |
    #                b11
    parent = {"root": None, "a": "root", "b": "root", "a1": "a", "a2": "a", "b1": "b", "b2": "b", "b11": "b1"}
    cost = {c: 1.0 for c in parent}
    cost["b11"] = 3.0
    return CliqueTree(parent, cost)


def chain_tree(n=6):
    return CliqueTree({i: (i + 1 if i + 1 < n else None) for i in range(n)})


def test_assignment_is_deterministic_and_keeps_subtrees_together():
    t = bushy_tree()
    a2 = assign_subtrees(t, 2)
    assert a2 == assign_subtrees(bushy_tree(), 2)
    # the two sibling subtrees under the root go to different ranks, each entirely on one rank
    assert len({a2[c] for c in t.subtree("a")}) == 1 and len({a2[c] for c in t.subtree("b")}) == 1
    assert a2["a"] != a2["b"]
    assert a2["root"] == a2["b"]                      # root sits with its costliest child
    a4 = assign_subtrees(t, 4)
    assert len(set(a4.values())) == 4
    assert a4["a1"] != a4["a2"] and a4["b1"] != a4["b2"]
    a1 = assign_subtrees(t, 1)
    assert set(a1.values()) == {0}


def test_chain_does_not_shard():
    """pose_first ordering gives a chain: 'replicas only', every clique on rank 0."""
    assert set(assign_subtrees(chain_tree(), 8).values()) == {0}


def test_forest_of_independent_cliques_is_balanced():
    t = CliqueTree({i: None for i in range(64)})
    a = assign_subtrees(t, 8)
    counts = np.bincount(list(a.values()), minlength=8)
    assert counts.min() == counts.max() == 8


def test_leaves_first_order():
    t = bushy_tree()
    order = t.leaves_first()
    pos = {c: i for i, c in enumerate(order)}
    for c, p in t.parent.items():
        if p is not None:
            assert pos[c] < pos[p]


def test_shard_independent_covers_everything_once():
    for n, w in ((64, 8), (10, 4), (3, 8), (128, 2)):
        got = sum((shard_independent(n, w, r) for r in range(w)), [])
        assert got == list(range(n))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def two_remote_children_tree():
    # p (rank 0) has two children trained on rank 1; consumed in the order (c2, c1), produced in the order (c1, c2)
    parent = {"p": None, "c1": "p", "c2": "p", "g1": "c1"}
    assign = {"p": 0, "c1": 1, "c2": 1, "g1": 1}
    return CliqueTree(parent), assign


def both_directions_tree():
    #        r(0)
    #      /      \
    #    a(1)      b(0)
    #    |          |
    #    a1(0)     b1(1)        edges: a1 0->1, b1 1->0, a 1->0  -- both directions between the two ranks, interleaved
    #    |
    #    a2(1)                  a2 1->0
    parent = {"r": None, "a": "r", "b": "r", "a1": "a", "b1": "b", "a2": "a1"}
    assign = {"r": 0, "a": 1, "b": 0, "a1": 0, "b1": 1, "a2": 1}
    return CliqueTree(parent), assign


TREES = {"bushy": lambda: (bushy_tree(), None), "two_remote_children": two_remote_children_tree,
         "both_directions": both_directions_tree}


def _fit_factory(tree, reverse_consumption=False):
    names = sorted(tree.parent, key=str)
    code = {c: float(i + 1) for i, c in enumerate(names)}

    def shape_of(c):
        return (4 + names.index(c), 3)              # every edge its own shape: a swapped batch cannot go unnoticed

    def fit(c, msgs):
        # stand-in for "sample the training batch from the children's flows, train, emit separator
        # samples": a deterministic function of the clique and everything below it
        base = torch.full(shape_of(c), code[c])
        for j, m in enumerate(msgs):
            base = base + 0.5 * (j + 1) * m.sum() / m.numel()
        return base
    return fit, shape_of, code


def _worker(rank, world, port, out_dir, which="bushy"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tree, assign = TREES[which]()
    fit, shape_of, _ = _fit_factory(tree)
    log, stats = [], {}
    produced = run_tree(tree, fit, rank, world, device="cpu", message_shape=shape_of, assignment=assign, exchange_log=log, exchange_stats=stats)
    torch.save({"produced": {k: v for k, v in produced.items()}, "log": log, "stats": stats}, os.path.join(out_dir, "rank%d.pt" % rank))
    # weak-scaling shards + the max-over-ranks timing reduction bench.py performs
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert t.item() == float(world)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
@pytest.mark.parametrize("which", ["bushy", "two_remote_children", "both_directions"])
def test_two_process_tree_run_matches_single_process(tmp_path, which):
    """No message carries a tag (RCCL has none): the k-th operation of rank 0 towards rank 1 must be the counterpart of
    the k-th operation of rank 1 towards rank 0.  Trees: sibling subtrees on two ranks; one parent with two remote
    children on the SAME rank; edges crossing in BOTH directions between the two ranks, interleaved."""
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path), which), nprocs=world, join=True)
    merged, logs, stats = {}, [], []
    for r in range(world):
        part = torch.load(os.path.join(str(tmp_path), "rank%d.pt" % r))
        assert not (set(part["produced"]) & set(merged))          # every clique ran on exactly one rank
        merged.update(part["produced"])
        logs.append(part["log"])
        stats.append(part["stats"])
    tree, assign = TREES[which]()
    assert set(merged) == set(tree.parent)
    assignment = assign_subtrees(tree, world) if assign is None else assign
    fit, shape_of, _ = _fit_factory(tree)

    def ref(c):           # single-process reference of the same recursion
        return fit(c, [ref(k) for k in tree.children[c]])
    for c in tree.parent:
        assert torch.equal(merged[c], ref(c)), c
    # the two ranks' operation sequences towards each other are mirror images, edge by edge
    assert len(logs[0]) == len(logs[1]) > 0
    for (k0, p0, e0), (k1, p1, e1) in zip(logs[0], logs[1]):
        assert e0 == e1 and {k0, k1} == {"send", "recv"} and p0 == 1 and p1 == 0, (logs[0], logs[1])
    crossing = [c for c in tree.parent if tree.parent[c] is not None and assignment[c] != assignment[tree.parent[c]]]
    assert sorted(e for _, _, e in logs[0]) == sorted(crossing)
    # the exchange's own accounting (bench.py's `exchange` block is made of it): every crossing batch counted once on each side
    total = sum(4 * shape_of(c)[0] * shape_of(c)[1] for c in crossing)
    assert stats[0]["bytes_sent"] + stats[1]["bytes_sent"] == total == stats[0]["bytes_received"] + stats[1]["bytes_received"]
    assert stats[0]["bytes_sent"] == stats[1]["bytes_received"] and stats[0]["sends"] + stats[0]["recvs"] == len(crossing)
    assert all(st["send_s"] >= 0 and st["wait_s"] >= 0 for st in stats)
    if which == "both_directions":
        assert {k for k, _, _ in logs[0]} == {"send", "recv"}
    if which == "two_remote_children":
        assert [k for k, _, _ in logs[0]] == ["recv", "recv"]


def test_edge_exchange_rejects_inconsistent_use():
    from slam.CliqueParallel import EdgeExchange
    with pytest.raises(ValueError):
        EdgeExchange([("e", 0, 0, (2, 2))], rank=0)
    with pytest.raises(ValueError):
        EdgeExchange([("e", 0, 1, (2, 2)), ("e", 1, 0, (2, 2))], rank=0)
    ex = EdgeExchange([("e", 0, 1, (2, 2))], rank=0)
    with pytest.raises(ValueError):
        ex.send("e", torch.zeros(3, 2))            # not the announced shape
    with pytest.raises(ValueError):
        ex.recv("e")                               # rank 0 is the sender of this edge


@pytest.mark.parametrize("depth", [1, 2, 3])
def test_meeting_tree_of_the_bench_exchange_regime_branches_and_shards_as_designed(depth):
    """bench.py's N > 1 `exchange` regime: the binary meeting tree of 2^depth robots.  Host-only check of what the regime
    rests on: the Bayes tree (natural ordering) has one {B, M | A} <- {A | X} arm per robot and 2^depth - 1 join cliques
    {X_2j, X_2j+1 | Y_j} whose two child subtrees have DISJOINT separators; `assign_subtrees` over 2^depth ranks gives every
    rank one arm and every join one local and one remote child: 2^depth - 1 cross-rank child -> parent edges."""
    import bench
    from slam.CliqueParallel import CliqueTree, assign_subtrees
    from slam.NFiSAM import NFiSAM, NFiSAMArgs
    order, factors = bench.meeting_tree(depth)
    arms = 2 ** depth
    assert len(order) == 4 * arms + (arms - 2)
    s = NFiSAM(NFiSAMArgs(num_knots=9, hidden_dim=8, elimination_method="natural", cuda_training=True))
    for v in order:
        s.add_node(v)
    for f in factors:
        s.add_factor(f)
    s.update_physical_and_working_graphs()
    cl = s._working_bayes_tree.clique_ordering()
    assert len(cl) == 2 * arms + (arms - 1)
    joins = [c for c in cl if len(c.children) == 2]
    assert len(joins) == arms - 1 and all(len(c.children) <= 2 for c in cl)
    for c in joins:
        a, b = [set(v.name for v in ch.separator) for ch in c.children]
        assert a and b and not (a & b) and (a | b) <= set(v.name for v in c.frontal)
    ids = {id(c): k for k, c in enumerate(cl)}
    parent = {ids[id(c)]: (ids[id(c.parent)] if c.parent is not None else None) for c in cl}
    cost = {ids[id(c)]: float(c.dim) ** 2 for c in cl}
    for world in (1, 2, arms):
        a = assign_subtrees(CliqueTree(parent, cost), world)
        cross = sum(1 for c in cl if c.parent is not None and a[ids[id(c)]] != a[ids[id(c.parent)]])
        assert set(a.values()) == set(range(world))
        assert cross == (0 if world == 1 else (1 if world == 2 else arms - 1)), (world, cross)
