#!/bin/bash
mkdir -p gpurun_out; rm -f gpurun_out/half16_stamps.txt
for cfg in "NFISAM_HALF=2 NFISAM_HALF_W=4"; do
  echo "== $cfg (sixteen copies in one pass)" >> gpurun_out/half16_stamps.txt
  env $cfg python scripts/stamps3.py 1 2000 15 persist 2>&1 | grep -v amdgpu.ids | head -18 >> gpurun_out/half16_stamps.txt
done
cat gpurun_out/half16_stamps.txt
