#!/bin/bash
# Diagnostic (GPU): the headline plan at K = 1 .. 125 steps (one chunk-persistent launch of K iterations + closing Adam + bookkeeping):
# total time per replay = intercept + slope x K.  The driver's bench is K = 20.    bash scripts/exp/short_plan_sweep.sh > gpurun_out/short_plan_sweep.txt
cd "$(dirname "$0")/../.."
for K in 1 2 5 10 20 40 80 125; do
  python bench.py --steps $K --warmup 5 --no-regimes --no-cpu-baseline --no-update-bench 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('K=%3d  %.2f us per step  %.1f us per replay   persistent launch %.1f us (%s iterations per launch)' % (d['steps'], 1e3*d['ms_per_step'], 1e3*d['ms_per_step']*d['steps'], r['kernel_us'], r.get('iterations_per_launch')))"
done
