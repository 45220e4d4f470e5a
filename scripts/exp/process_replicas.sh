#!/bin/bash
# One-off (GPU): R independent single-solver Plaza1 runs as R PROCESSES sharing the GPU (kernels of different processes
# overlap on the device?), against scripts/run_plaza1.py REPLICAS=R (one process, slots of one training plan).
R=${1:-8}
nproc
t0=$(date +%s.%N)
pids=()
for r in $(seq 0 $((R-1))); do
  SEED=$r EVERY=1000 python scripts/run_plaza1.py > /tmp/proc_rep_$r.log 2>&1 &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
t1=$(date +%s.%N)
echo "processes: $R runs from $t0 to $t1 (includes python start-up + import of every process)"
for r in $(seq 0 $((R-1))); do tail -1 /tmp/proc_rep_$r.log | cut -c1-200; done
