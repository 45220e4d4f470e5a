#!/bin/bash
mkdir -p gpurun_out; rm -f gpurun_out/half_split.txt
for cfg in "NFISAM_HALF=0" "NFISAM_HALF=0 NFISAM_PERSIST_SPLIT=1" "NFISAM_HALF=1" "NFISAM_HALF=1 NFISAM_PERSIST_SPLIT=1"; do
  echo "== $cfg" >> gpurun_out/half_split.txt
  env $cfg python scripts/time_grad.py 1 2000 15 2>&1 | grep -v amdgpu.ids >> gpurun_out/half_split.txt
  env $cfg python scripts/time_grad.py 1 2000 15 2>&1 | grep -v amdgpu.ids >> gpurun_out/half_split.txt
  env $cfg python scripts/stamps3.py 1 2000 15 persist 2>&1 | grep -v amdgpu.ids | head -17 >> gpurun_out/half_split.txt
done
cat gpurun_out/half_split.txt
