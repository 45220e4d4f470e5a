#!/bin/bash
mkdir -p gpurun_out; rm -f gpurun_out/span_check.txt
for cfg in "NFISAM_SPAN=0" "NFISAM_SPAN=1"; do
  for shape in "2000 15" "1000 15"; do
    echo -n "$cfg | " >> gpurun_out/span_check.txt
    env $cfg timeout 120 python scripts/time_grad.py 1 $shape 2>&1 | grep -v amdgpu.ids | tail -1 >> gpurun_out/span_check.txt
  done
  echo -n "$cfg | plaza1 30 updates: " >> gpurun_out/span_check.txt
  env $cfg timeout 300 python scripts/run_plaza1.py 30 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-400 >> gpurun_out/span_check.txt
done
cat gpurun_out/span_check.txt
