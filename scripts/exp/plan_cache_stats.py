"""Experiment (GPU): how much of the replicas' "train" phase is plan creation (graph capture + instantiate)?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "nf-isam_amd")); sys.path.insert(0, ROOT)
import nfisam_hip as nh
stats = {"creates": 0, "create_s": 0.0, "runs": 0, "run_s": 0.0, "reset_s": 0.0}
_prep, _run, _reset = nh.TrainBatch.prepare, nh.TrainBatch.run, nh.TrainBatch.reset
def prepare(self, use_graph=True):
    new = getattr(self, "_plan", None) is None
    t0 = time.time(); _prep(self, use_graph)
    if new: stats["creates"] += 1; stats["create_s"] += time.time() - t0
per = []
def run(self, use_graph=True):
    t0 = time.time(); r = _run(self, use_graph); dt = time.time() - t0
    stats["runs"] += 1; stats["run_s"] += dt
    per.append((self.nc, self.max_D, max(r), sum(r) / len(r), dt))
    return r
def reset(self, kparams=None):
    t0 = time.time(); _reset(self, kparams); stats["reset_s"] += time.time() - t0
nh.TrainBatch.prepare, nh.TrainBatch.run, nh.TrainBatch.reset = prepare, run, reset
os.environ.setdefault("REPLICAS", "8"); os.environ["EVERY"] = "1000"
sys.argv = ["run_plaza1.py", "100000"]
f = os.path.join(ROOT, "scripts", "run_plaza1.py")
__file__ = f
try:
    exec(compile(open(f).read(), f, "exec"))
except SystemExit:
    pass
print(stats)
import numpy as np
a = np.array(per)
print('batches %d: mean cliques %.1f, mean max_D %.1f, mean of max iterations %.0f, mean of mean iterations %.0f, us per (max) iteration: median %.2f mean %.2f' % (len(a), a[:,0].mean(), a[:,1].mean(), a[:,2].mean(), a[:,3].mean(), np.median(1e6*a[:,4]/a[:,2]), 1e6*a[:,4].sum()/a[:,2].sum()))
for D in sorted(set(a[:,1])):
    m = a[:,1] == D
    print('  max_D %d: %d batches, us per iteration %.2f, max iters %.0f' % (D, m.sum(), 1e6*a[m,4].sum()/a[m,2].sum(), a[m,2].mean()))
