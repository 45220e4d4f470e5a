#!/bin/bash
# round 6: the fits of an update enqueued (async_fits), with and without the lazy posterior, against the synchronous solver: Plaza1 end to end
mkdir -p gpurun_out; rm -f gpurun_out/async_fits.txt
for rep in 1 2; do
for cfg in "ASYNC=0 LAZY=0" "ASYNC=1 LAZY=0" "ASYNC=0 LAZY=1" "ASYNC=1 LAZY=1"; do
  echo -n "$cfg | " >> gpurun_out/async_fits.txt
  env $cfg python scripts/run_plaza1.py 1000 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('Plaza1 total %.3f s  walls %.3f  fit %.3f  sampling %.3f  posterior %.3f  graph %.3f  iterations %d' % (d['total_s'], d['wall_per_update_mean']*d['updates'], d['fitting_total_s'], d['sampling_total_s'], d['posterior_total_s'], d['graph_total_s'], d['training_sample_iters']/2000))" >> gpurun_out/async_fits.txt
done; done
cat gpurun_out/async_fits.txt
