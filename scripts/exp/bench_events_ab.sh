#!/bin/bash
# the two HIP events of bench.py's replays (GPU time of the region): inside every wall-clock replay (up to round 5) against replays of their own
mkdir -p gpurun_out; rm -f gpurun_out/bench_events_ab.txt
for rep in 1 2 3; do
for cfg in "BENCH_EVENTS_IN_REPLAY=1" "BENCH_EVENTS_IN_REPLAY=0"; do
  for K in 20 500; do
  echo -n "$cfg | " >> gpurun_out/bench_events_ab.txt
  env $cfg python bench.py --steps $K --warmup 5 --no-regimes --no-cpu-baseline --no-update-bench --no-replicas 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('K=%3d  %.3f us per step  value %.4e  gpu events %.3f us per step  persistent launch %.1f us' % (d['steps'], 1e3*d['ms_per_step'], d['value'], 1e3*d['gpu_ms_per_step_events'], r['kernel_us']))" >> gpurun_out/bench_events_ab.txt
  done
done; done
cat gpurun_out/bench_events_ab.txt
