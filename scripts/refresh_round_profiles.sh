#!/bin/bash
# Everything profiles/rNN_* is made from, in one gpurun call:  scripts/refresh_round_profiles.sh gpurun_out/<dir>
out=$1
mkdir -p $out
python bench.py 2>/dev/null | tail -1 > $out/bench_line.json
bash scripts/collect_profiles.sh $out > $out/collect.log 2>&1
python scripts/run_plaza1.py 100000 $out/plaza1_end_to_end.json > $out/plaza1.log 2>&1
DATASET=Plaza1ADA0.4EFG python scripts/run_plaza1.py 100000 $out/plaza1_ada04_end_to_end.json > $out/plaza1_ada04.log 2>&1
DATASET=Manhattan200 STEP=1 ITERS=500 TOL=1e-9 python scripts/run_plaza1.py 100000 $out/manhattan200_end_to_end.json > $out/manhattan200.log 2>&1
REPLICAS=8 python scripts/run_plaza1.py 100000 $out/plaza1_replicas8.json > $out/plaza1_replicas8.log 2>&1                                # every replica at its own pace
FREE=0 REPLICAS=8 python scripts/run_plaza1.py 100000 $out/plaza1_replicas8_update_barrier.json > $out/plaza1_replicas8_barrier.log 2>&1   # slots, barrier per update
FREE=0 NFISAM_REPLICA_SLOTS=0 REPLICAS=8 python scripts/run_plaza1.py 100000 $out/plaza1_replicas8_lock_step.json > $out/plaza1_replicas8_lock.log 2>&1
python scripts/pipeline_report.py $out/pipeline_parity_vs_reference.json > $out/pipeline.log 2>&1
for s in "1 2000 15" "8 2000 12" "64 2000 15"; do python scripts/stamps3.py $s; done > $out/phase_cycles_stamps3.txt 2>&1
# the chunk-persistent form, per iteration: C3 (widest clique first) and one Plaza clique; then C3 with one launch per iteration
for s in "0 2000 12 persist" "1 2000 15 persist"; do python scripts/stamps3.py $s; done > $out/phase_cycles_persistent.txt 2>&1
NFISAM_PERSIST=0 python scripts/stamps3.py 0 2000 12 persist >> $out/phase_cycles_persistent.txt 2>&1
tail -2 $out/plaza1.log $out/plaza1_ada04.log $out/manhattan200.log $out/plaza1_replicas8.log | cut -c1-400
