#!/bin/bash
# round 6: first run of the two-lanes-per-particle dim-major kernel (nsf_half.h): parity, then the Plaza clique's time
mkdir -p gpurun_out
python -m pytest tests/test_hip_parity.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/half_parity.txt
for cfg in "NFISAM_HALF=0" "NFISAM_HALF=1 NFISAM_HALF_W=4" "NFISAM_HALF=1 NFISAM_HALF_W=8"; do
  for rep in 1 2; do
    echo "== $cfg" >> gpurun_out/half_time.txt
    env $cfg python scripts/time_grad.py 1 2000 15 >> gpurun_out/half_time.txt 2>&1
  done
done
env NFISAM_HALF=0 python scripts/time_grad.py 8 2000 12 >> gpurun_out/half_time.txt 2>&1
cat gpurun_out/half_parity.txt gpurun_out/half_time.txt
