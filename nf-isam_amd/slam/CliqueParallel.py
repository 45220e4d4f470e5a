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
never an all-reduce).  Messages are matched by ORDER, not by tag (RCCL has none): `EdgeExchange`.  Chain-shaped trees (the `pose_first` ordering of every shipped large
example) have no sibling subtrees: they run on one rank ("replicas only").

Nothing here touches the kernels; it is scheduling + message routing around
`slam.NFiSAM.NFiSAM.fit_clique_density_model`.
"""
import time
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


class EdgeExchange:
    """Point-to-point exchange of fixed-shape fp32 batches along the cross-rank edges of ONE pass over the tree, correct
    WITHOUT message tags.

    RCCL / NCCL has no tags: between two ranks, point-to-point operations are matched strictly in the order they were
    issued, and the sends and receives a rank addresses to one peer share a stream, so they also EXECUTE in that order.
    Matching by tag (gloo) hides both facts.  Here every rank is given the same list of edges in the same GLOBAL order
    (the order in which the pass visits the producing cliques), and for every peer it issues its operations -- sends and
    receives alike -- in exactly that order:

      * `send(key, batch)`   first posts the (not yet posted) receives that precede this edge in the pair's sequence,
                             then the send;
      * `recv(key)`          posts every operation of the pair up to this edge (they can only be receives: a send that
                             precedes it belongs to a clique this rank has already finished) and waits for this one.

    So the k-th operation of rank A towards B is always the counterpart of the k-th operation of B towards A, whatever
    the order in which the batches are consumed, with several edges between one pair and with edges in both directions.
    Both ends know the shape of a batch ([n, Ds]: sample count x separator dims), so there is no header message.
    Tensors stay on the device with the "nccl" backend (RCCL moves device memory); gloo gets host tensors."""

    def __init__(self, edges, rank: int, device="cpu", on_device: bool = False):
        """edges: list of (key, src_rank, dst_rank, shape) with src_rank != dst_rank, identical on every rank."""
        self.rank, self.device, self.on_device = rank, device, on_device
        self.edges = {}
        self.seq: Dict[int, List] = {}           # peer -> keys of the edges between this rank and the peer, global order
        self.pos: Dict[Hashable, int] = {}
        for key, src, dst, shape in edges:
            if src == dst:
                raise ValueError("edge %r does not cross ranks" % (key,))
            if key in self.edges:
                raise ValueError("duplicate edge key %r" % (key,))
            self.edges[key] = (src, dst, tuple(int(v) for v in shape))
            if rank in (src, dst):
                peer = dst if src == rank else src
                self.pos[key] = len(self.seq.setdefault(peer, []))
                self.seq[peer].append(key)
        self.cursor = {peer: 0 for peer in self.seq}
        self._recv: Dict[Hashable, tuple] = {}    # key -> (request, buffer)
        self._sends: List[tuple] = []             # (request, buffer): kept alive until drain()
        self.log: List[tuple] = []                # ("send" | "recv", peer, key) in issue order (tests)
        # what the pass cost this rank (bench.py's `exchange` block): seconds inside send() -- staging + isend --, seconds
        # BLOCKED in recv() / drain() (that is the producer's remaining work plus the transfer: a dependency wait, not
        # link time), bytes out / in
        self.stats = {"send_s": 0.0, "wait_s": 0.0, "bytes_sent": 0, "bytes_received": 0, "sends": 0, "recvs": 0}

    def _buffer_device(self):
        return self.device if self.on_device else "cpu"

    def _post_until(self, peer: int, upto: int):
        """Issue this rank's operations towards `peer` with sequence position < upto; they must all be receives."""
        while self.cursor[peer] < upto:
            key = self.seq[peer][self.cursor[peer]]
            src, dst, shape = self.edges[key]
            if dst != self.rank:
                raise RuntimeError("edge %r: its send has to be issued before a later operation of the same rank pair "
                                   "(the pass must visit producers in the global edge order)" % (key,))
            buf = torch.empty(shape, dtype=torch.float32, device=self._buffer_device())
            self._recv[key] = (dist.irecv(buf, src=src), buf)
            self.log.append(("recv", src, key))
            self.cursor[peer] += 1

    def send(self, key, batch: torch.Tensor):
        src, dst, shape = self.edges[key]
        if src != self.rank:
            raise ValueError("edge %r is not sent by rank %d" % (key, self.rank))
        t0 = time.perf_counter()
        t = batch.to(torch.float32).contiguous()
        if tuple(t.shape) != shape:
            raise ValueError("edge %r carries %s, announced %s" % (key, tuple(t.shape), shape))
        t = t.to(self._buffer_device())
        self._post_until(dst, self.pos[key])
        if self.cursor[dst] != self.pos[key]:
            raise RuntimeError("edge %r sent twice or out of order" % (key,))
        self._sends.append((dist.isend(t, dst=dst), t))
        self.log.append(("send", dst, key))
        self.cursor[dst] += 1
        self.stats["send_s"] += time.perf_counter() - t0
        self.stats["bytes_sent"] += 4 * t.numel()
        self.stats["sends"] += 1

    def recv(self, key) -> torch.Tensor:
        src, dst, _ = self.edges[key]
        if dst != self.rank:
            raise ValueError("edge %r is not received by rank %d" % (key, self.rank))
        t0 = time.perf_counter()
        self._post_until(src, self.pos[key] + 1)
        req, buf = self._recv[key]
        if req is not None:
            req.wait()
            self._recv[key] = (None, buf)
            self.stats["bytes_received"] += 4 * buf.numel()
            self.stats["recvs"] += 1
        self.stats["wait_s"] += time.perf_counter() - t0
        return buf

    def drain(self):
        """End of the pass: post what was never asked for, wait for every outstanding operation (the buffers of the
        sends are kept until here)."""
        t0 = time.perf_counter()
        for peer in self.seq:
            self._post_until(peer, len(self.seq[peer]))
        for key, (req, buf) in list(self._recv.items()):
            if req is not None:
                req.wait()
                self._recv[key] = (None, buf)
                self.stats["bytes_received"] += 4 * buf.numel()
                self.stats["recvs"] += 1
        for req, _ in self._sends:
            req.wait()
        self._sends = []
        self.stats["wait_s"] += time.perf_counter() - t0


def tree_edges(tree: CliqueTree, assignment: Dict[Hashable, int], order: Sequence, shape_of: Callable) -> List[tuple]:
    """Cross-rank child -> parent edges of a leaves-first pass, in the order the children are visited."""
    return [(c, assignment[c], assignment[tree.parent[c]], shape_of(c)) for c in order
            if tree.parent[c] is not None and assignment[tree.parent[c]] != assignment[c]]


def run_tree(tree: CliqueTree, fit: Callable[[Hashable, List[torch.Tensor]], torch.Tensor], rank: int,
             world_size: int, device="cpu", message_shape=None, assignment: Dict[Hashable, int] = None,
             exchange_log: List = None, exchange_stats: Dict = None) -> Dict[Hashable, torch.Tensor]:
    """Execute `fit(clique, child_messages) -> message_to_parent` for every clique owned by this rank,
    leaves first, routing messages over rank boundaries point-to-point (`EdgeExchange`: no tags, no headers).
    Per clique the reference's order is kept: children's separator samples -> fit -> separator factor for the
    parent (FactorGraphSolver.py:436-470).  `message_shape(clique)` = shape of the clique's message (known to both
    ends; default (5, 3), the stand-in of the tests).  Returns {clique: message} for the cliques this rank ran."""
    assignment = assign_subtrees(tree, world_size) if assignment is None else assignment
    order = tree.leaves_first()
    shape_of = message_shape if message_shape is not None else (lambda c: (5, 3))
    ex = EdgeExchange(tree_edges(tree, assignment, order, shape_of), rank, device=device,
                      on_device=(dist.get_backend() == "nccl"))
    produced: Dict[Hashable, torch.Tensor] = {}
    for c in order:
        if assignment[c] != rank:
            continue
        msgs = [produced[k] if assignment[k] == rank else ex.recv(k).to(device) for k in tree.children[c]]
        produced[c] = fit(c, msgs)
        p = tree.parent[c]
        if p is not None and assignment[p] != rank:
            ex.send(c, produced[c])
    ex.drain()
    if exchange_log is not None:
        exchange_log.extend(ex.log)
    if exchange_stats is not None:               # what the pass cost this rank (EdgeExchange.stats: bench.py's `exchange` block)
        exchange_stats.update(ex.stats)
    return produced
