#!/usr/bin/env python3
"""Fixture for the end-to-end posterior parity test on BASELINE config[0]
(example/slam/small_range_gaussian_problem, journal_paper/case1).

Copies DATA (no source) out of the reference tree, runs only in the build container:
  * the input factor graph (`factor_graph.fg`, 22 text lines, MIT licence) — stored verbatim as a string;
  * the reference's own NF-iSAM posterior samples of that problem (`run1/step{0..5}`, 1000 rows each,
    produced by the reference authors with the arguments in `run1/parameters`), float32;
  * the dynamic-nested-sampling "ground truth" posteriors (`dyn1/step{0..3}.sample`; steps 4-5 are
    large blobs missing from the checkout), sub-sampled to 1000 rows with a fixed seed;
  * the column orderings of both.
Output: tests/golden/small_range_case1.npz
"""
import os

import numpy as np

CASE = "/root/reference/example/slam/small_range_gaussian_problem/journal_paper/case1"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "small_range_case1.npz")

out = {"factor_graph_fg": np.array(open(os.path.join(CASE, "factor_graph.fg")).read()),
       "run1_parameters": np.array(open(os.path.join(CASE, "run1", "parameters")).read())}
rng = np.random.RandomState(0)
for i in range(6):
    out["run1_step%d" % i] = np.loadtxt(os.path.join(CASE, "run1", "step%d" % i)).astype(np.float32)
    out["run1_step%d_ordering" % i] = np.array(open(os.path.join(CASE, "run1", "step%d_ordering" % i)).read())
for i in range(4):
    s = np.loadtxt(os.path.join(CASE, "dyn1", "step%d.sample" % i))
    idx = rng.choice(s.shape[0], size=min(1000, s.shape[0]), replace=False)
    out["dyn1_step%d" % i] = s[idx].astype(np.float32)
    out["dyn1_step%d_ordering" % i] = np.array(open(os.path.join(CASE, "dyn1", "step%d_ordering" % i)).read())
for k in ("fitting_timer", "posterior_sampling_timer", "step_timing"):
    out["run1_" + k] = np.array([float(t) for t in open(os.path.join(CASE, "run1", k)).read().split()])
np.savez_compressed(OUT, **out)
print("wrote", OUT, {k: getattr(v, "shape", None) for k, v in out.items() if k.startswith(("run1_step", "dyn1_step")) and v.ndim == 2})
