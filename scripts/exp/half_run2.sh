#!/bin/bash
mkdir -p gpurun_out; rm -f gpurun_out/half_run2.txt
for shape in "2000 15" "1500 15" "2000 8"; do
for cfg in "NFISAM_HALF=0" "NFISAM_HALF=1"; do
  echo -n "$cfg | " >> gpurun_out/half_run2.txt
  env $cfg python scripts/time_grad.py 1 $shape 2>&1 | grep -v amdgpu.ids >> gpurun_out/half_run2.txt
done; done
cat gpurun_out/half_run2.txt
