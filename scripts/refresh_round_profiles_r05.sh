#!/bin/bash
# Round 5: what profiles/r05_* is made from, in one gpurun call (no stamps builds: the dim-major unit's text did not change this round)
out=$1
mkdir -p $out
python bench.py 2>/dev/null | tail -1 > $out/bench_line.json
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $out/bench_line_driver_style_20_steps.json
bash scripts/collect_profiles.sh $out > $out/collect.log 2>&1
python scripts/run_plaza1.py 100000 $out/plaza1_end_to_end.json > $out/plaza1.log 2>&1
DATASET=Plaza1ADA0.4EFG python scripts/run_plaza1.py 100000 $out/plaza1_ada04_end_to_end.json > $out/plaza1_ada04.log 2>&1
DATASET=Manhattan200 STEP=1 ITERS=500 TOL=1e-9 python scripts/run_plaza1.py 100000 $out/manhattan200_end_to_end.json > $out/manhattan200.log 2>&1
REPLICAS=8 python scripts/run_plaza1.py 100000 $out/plaza1_replicas8.json > $out/plaza1_replicas8.log 2>&1
python scripts/pipeline_report.py $out/pipeline_parity_vs_reference.json > $out/pipeline.log 2>&1
BENCH_DIST_BACKEND=gloo python bench.py --gpus 2 --steps 20 --warmup 5 2>/dev/null | tail -1 > $out/bench_line_two_gloo_ranks_on_one_gpu.json
tail -n 2 $out/plaza1.log $out/plaza1_ada04.log $out/manhattan200.log $out/plaza1_replicas8.log $out/pipeline.log | cut -c1-300
