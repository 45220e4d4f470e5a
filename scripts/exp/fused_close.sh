#!/bin/bash
# round 6: the end of a chunk-persistent chunk as ONE kernel (closing Adam blocks + bookkeeping block) against two kernels
mkdir -p gpurun_out; rm -f gpurun_out/fused_close.txt
python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "persist or twin or identical or chunk or span or validat or plan" 2>&1 | tail -3 >> gpurun_out/fused_close.txt
for rep in 1 2; do
for cfg in "NFISAM_FUSED_CLOSE=0" "NFISAM_FUSED_CLOSE=1"; do
  for K in 20 500; do
  echo -n "$cfg | " >> gpurun_out/fused_close.txt
  env $cfg python bench.py --steps $K --warmup 5 --no-regimes --no-cpu-baseline --no-update-bench --no-replicas 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('K=%3d  %.3f us per step  value %.4e  persistent launch %.1f us frac %.4f' % (d['steps'], 1e3*d['ms_per_step'], d['value'], r['kernel_us'], r['frac']))" >> gpurun_out/fused_close.txt
  done
  echo -n "$cfg | " >> gpurun_out/fused_close.txt
  env $cfg python scripts/run_plaza1.py 1000 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('Plaza1 total %.3f s fit %.3f s iterations %d samples/s %.4e' % (d['total_s'], d['fitting_total_s'], d['training_sample_iters']/2000, d['flow_training_samples_per_s']))" >> gpurun_out/fused_close.txt
done; done
cat gpurun_out/fused_close.txt
