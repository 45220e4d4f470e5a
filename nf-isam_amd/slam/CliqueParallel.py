"""slam.CliqueParallel — sharding of Bayes-tree cliques over the GPUs of one node.

The reference is single-process and trains cliques strictly one after the other
(src/slam/FactorGraphSolver.py:409-477).  What can run concurrently is dictated by the data
dependencies of NF-iSAM:

  * a clique's training batch needs samples from each CHILD clique's trained flow
    (FlowsPriorFactor.sample, src/slam/NFiSAM.py:271-288) -> leaves first, parents after children;
  * sibling subtrees never exchange anything -> they are the sharding unit.

So: one process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on the GPU box,
"gloo" in the CPU tests), whole subtrees are assigned to ranks by a greedy longest-processing-time
rule, every rank trains its own cliques with NO collective on the data path, and only a tree edge
whose two ends live on different ranks costs one point-to-point message: the child's separator
sample batch [n, Ds] fp32 (<= 2000 x 11 x 4 B = 88 KB, latency-bound on a 153 GB/s xGMI link —
never an all-reduce).  Chain-shaped trees (the `pose_first` ordering of every shipped large
example) have no sibling subtrees: they run on one rank ("replicas only").

Nothing here touches the kernels; it is scheduling + message routing around
`slam.NFiSAM.NFiSAM.fit_clique_density_model`.
"""
from typing import Callable, Dict, Hashable, List, Optional, Sequence

import numpy as np
import torch
import torch.distributed as dist


class CliqueTree:
    """Minimal rooted tree over clique ids: parent[c] is None for the root(s)."""

    def __init__(self, parent: Dict[Hashable, Optional[Hashable]], cost: Dict[Hashable, float] = None):
        self.parent = dict(parent)
        self.children = {c: [] for c in parent}
        for c, p in parent.items():
            if p is not None:
                self.children[p].append(c)
        for c in self.children:
            self.children[c].sort(key=str)          # deterministic on every rank
        self.cost = {c: 1.0 for c in parent} if cost is None else dict(cost)
        self.roots = sorted([c for c, p in parent.items() if p is None], key=str)

    def subtree(self, c):
        out, stack = [], [c]
        while stack:
            v = stack.pop()
            out.append(v)
            stack.extend(self.children[v])
        return out

    def subtree_cost(self, c):
        return sum(self.cost[v] for v in self.subtree(c))

    def leaves_first(self):
        """Cliques in an order where every clique comes after all of its children
        (the reference pops a BFS ordering from the end, FactorGraphSolver.py:419-425)."""
        order, stack = [], list(self.roots)
        while stack:
            v = stack.pop()
            order.append(v)
            stack.extend(self.children[v])
        return order[::-1]


def assign_subtrees(tree: CliqueTree, world_size: int) -> Dict[Hashable, int]:
    """clique -> rank.  Walk down from the roots: while a clique has several child subtrees and
    ranks to spare, the subtrees are dealt to rank groups by descending cost (LPT); a subtree that
    ends up with a single rank stays entirely on it.  A clique with children on several ranks runs
    on the rank of its costliest child (that child's message then stays local).  Deterministic."""
    assignment: Dict[Hashable, int] = {}

    def place(c, ranks: Sequence[int]):
        kids = tree.children[c]
        if len(ranks) == 1 or len(kids) < 2:
            if len(kids) == 1 and len(ranks) > 1:
                place(kids[0], ranks)
                assignment[c] = assignment[kids[0]]
            else:
                for v in tree.subtree(c):
                    assignment[v] = ranks[0]
            return
        kids_sorted = sorted(kids, key=lambda k: (-tree.subtree_cost(k), str(k)))
        if len(kids_sorted) >= len(ranks):
            # at least as many subtrees as ranks: longest-processing-time onto single ranks
            load = {r: 0.0 for r in ranks}
            for k in kids_sorted:
                r = min(ranks, key=lambda q: (load[q], q))
                place(k, [r])
                load[r] += tree.subtree_cost(k)
        else:
            # ranks to spare: deal them round-robin, costliest subtree first, and recurse
            groups: List[List[int]] = [[] for _ in kids_sorted]
            for j, r in enumerate(ranks):
                groups[j % len(kids_sorted)].append(r)
            for k, g in zip(kids_sorted, groups):
                place(k, g)
        assignment[c] = assignment[kids_sorted[0]]

    ranks = list(range(world_size))
    if len(tree.roots) == 1:
        place(tree.roots[0], ranks)
    else:       # a forest (e.g. a synthetic batch of independent cliques): LPT over the trees
        load = [0.0] * world_size
        for r_ in sorted(tree.roots, key=lambda k: (-tree.subtree_cost(k), str(k))):
            q = min(range(world_size), key=lambda j: (load[j], j))
            for v in tree.subtree(r_):
                assignment[v] = q
            load[q] += tree.subtree_cost(r_)
    return assignment


def shard_independent(n_items: int, world_size: int, rank: int) -> List[int]:
    """Indices of a batch of independent cliques owned by `rank` (contiguous blocks, remainder to the
    first ranks) — the weak-scaling layout of bench.py."""
    base, rem = divmod(n_items, world_size)
    start = rank * base + min(rank, rem)
    return list(range(start, start + base + (1 if rank < rem else 0)))


def send_separator_samples(samples: torch.Tensor, dst: int, tag: int = 0):
    """Child -> parent message: [n, Ds] float32 (device tensor with nccl/RCCL, CPU tensor with gloo)."""
    hdr = torch.tensor(list(samples.shape), dtype=torch.int64, device=samples.device)
    dist.send(hdr, dst=dst, tag=tag)
    dist.send(samples.contiguous(), dst=dst, tag=tag)


def recv_separator_samples(src: int, device, tag: int = 0) -> torch.Tensor:
    hdr = torch.empty(2, dtype=torch.int64, device=device)
    dist.recv(hdr, src=src, tag=tag)
    out = torch.empty(int(hdr[0]), int(hdr[1]), dtype=torch.float32, device=device)
    dist.recv(out, src=src, tag=tag)
    return out


def run_tree(tree: CliqueTree, fit: Callable[[Hashable, List[torch.Tensor]], torch.Tensor], rank: int,
             world_size: int, device="cpu") -> Dict[Hashable, torch.Tensor]:
    """Execute `fit(clique, child_messages) -> message_to_parent` for every clique owned by this rank,
    leaves first, routing messages over rank boundaries point-to-point.  Per clique the reference's
    order is kept: children's separator samples -> fit -> separator factor for the parent
    (FactorGraphSolver.py:436-470).  Returns {clique: message} for the cliques this rank ran.

    Deadlock-free by construction: all ranks walk the same global leaves-first order, a send is posted
    right after its clique finished and the matching recv is posted by the parent's rank when it
    reaches the parent, which is later in the same order on every rank."""
    assignment = assign_subtrees(tree, world_size)
    order = tree.leaves_first()
    index = {c: j for j, c in enumerate(order)}
    produced: Dict[Hashable, torch.Tensor] = {}
    for c in order:
        owner = assignment[c]
        if owner == rank:
            msgs = []
            for k in tree.children[c]:
                if assignment[k] == rank:
                    msgs.append(produced[k])
                else:
                    msgs.append(recv_separator_samples(assignment[k], device, tag=index[k]))
            produced[c] = fit(c, msgs)
        p = tree.parent[c]
        if owner == rank and p is not None and assignment[p] != rank:
            send_separator_samples(produced[c], assignment[p], tag=index[c])
    return produced
