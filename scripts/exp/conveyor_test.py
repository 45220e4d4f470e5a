"""Experiment / debug (GPU): the hand-stepped plan as a conveyor (begin / feed / peek / refill / end) on synthetic problems:
R slots, each slot trains a sequence of problems; several begin..end cycles; results against one plan run per problem."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "nf-isam_amd")); sys.path.insert(0, ROOT)
import nfisam_hip as nh
import bench as BM
dev = torch.device("cuda:0")
K, H, B, L = 9, 8, 5.0, 1
R, n, D, per_slot, cycles = 4, 2000, 7, 3, 4
rng = np.random.RandomState(0)
def problem(i):
    r = np.random.RandomState(i)
    x = torch.from_numpy((r.randn(n, D) * (0.5 + (i % 5) * 0.3)).astype(np.float32)).to(dev)
    kp = nh.pack(torch.from_numpy(BM.init_blob_np(D, K, H, L, i)).to(dev), D, K, H, L)
    return x, kp
kw = dict(lr=0.02, max_iters=600, average_window=50, loss_delta_tol=0.02, early_stop=True)
ref = {}
for i in range(R * per_slot * cycles):
    x, kp = problem(i)
    tb1 = nh.TrainBatch([x], [kp.clone()], K, H, B, L, **kw)
    it = tb1.run(use_graph=True)
    ref[i] = (it[0], tb1.kparams[0].clone(), tb1.iter_loss[0].clone())
    tb1.close()
print("reference iterations:", [ref[i][0] for i in range(8)], "...")
tb = nh.TrainBatch([torch.zeros(n, D, device=dev) for _ in range(R)], [torch.zeros_like(problem(0)[1]) for _ in range(R)], K, H, B, L, **kw)
tb.states[:, 1] = 1
nxt = 0
for cyc in range(cycles):
    t0 = time.time()
    tb.begin(); tb.feed(2)
    owner, seq0, keep, left = [None] * R, [0] * R, [None] * R, [per_slot] * R
    def load(r):
        global nxt
        i = nxt; nxt += 1
        x, kp = problem(i)
        keep[r] = (x, kp)
        tb.refill(r, x, kp)
        owner[r], seq0[r] = i, tb.enqueued()
        left[r] -= 1
    for r in range(R):
        load(r)
    bad = 0
    t_last = time.time()
    while any(o is not None for o in owner):
        seq, st = tb.peek()
        if seq < 0:
            continue
        for r in range(R):
            if owner[r] is not None and seq > seq0[r] and (st[r][1] != 0 or st[r][0] >= 600):
                i = owner[r]; owner[r] = None
                ok = st[r][0] == ref[i][0] and torch.equal(tb.kparams[r], ref[i][1])
                bad += 0 if ok else 1
                if not ok:
                    print("  MISMATCH problem %d slot %d: iterations %d vs %d, first losses %s vs %s" % (
                        i, r, st[r][0], ref[i][0], tb.iter_loss[r][:4].tolist(), ref[i][2][:4].tolist()))
                if left[r] > 0:
                    load(r)
                t_last = time.time()
        if time.time() - t_last > 10:
            print("  STUCK: seq %d enqueued %d states %r owner %r seq0 %r" % (seq, tb.enqueued(), st, owner, seq0)); sys.exit(1)
        time.sleep(2e-5)
    tb.end(); torch.cuda.synchronize()
    print("cycle %d: %.1f ms, %d chunks, mismatches %d" % (cyc, 1e3 * (time.time() - t0), tb.enqueued(), bad))
