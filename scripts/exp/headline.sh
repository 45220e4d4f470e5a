#!/bin/bash
mkdir -p gpurun_out; rm -f gpurun_out/headline.txt
for rep in 1 2; do
for K in 500 20; do
  python bench.py --steps $K --warmup 5 --no-regimes --no-cpu-baseline --no-update-bench 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('K=%3d  %.3f us per step  value %.4e  persistent launch %.1f us (%s iterations per launch) frac %.4f' % (d['steps'], 1e3*d['ms_per_step'], d['value'], r['kernel_us'], r.get('iterations_per_launch'), r['frac']))" >> gpurun_out/headline.txt
done; done
for cfg in "NFISAM_HALF=0 NFISAM_LONE_LEAN=0" "NFISAM_HALF=0"; do
  echo -n "$cfg | " >> gpurun_out/headline.txt
  env $cfg python scripts/time_grad.py 1 2000 15 2>&1 | grep -v amdgpu.ids >> gpurun_out/headline.txt
done
python scripts/time_grad.py 8 2000 12 2>&1 | grep -v amdgpu.ids >> gpurun_out/headline.txt
python scripts/time_grad.py 64 2000 15 2>&1 | grep -v amdgpu.ids >> gpurun_out/headline.txt
cat gpurun_out/headline.txt
