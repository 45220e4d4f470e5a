#!/bin/bash
# round 6: sixteen copies in one pass (two lanes per particle) against the 64-particle two-wave build, single cliques of 1025..2048 particles
mkdir -p gpurun_out; rm -f gpurun_out/half16.txt
for rep in 1 2; do
for shape in "1500 15" "2000 8" "2000 15" "2000 12" "2048 16"; do
  for cfg in "NFISAM_HALF=0" "NFISAM_HALF=2"; do
    echo -n "$cfg | " >> gpurun_out/half16.txt
    env $cfg python scripts/time_grad.py 1 $shape 2>&1 | grep -v amdgpu.ids >> gpurun_out/half16.txt
  done
done
done
python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "persist or twin or identical or chunk" 2>&1 | tail -3 >> gpurun_out/half16.txt
cat gpurun_out/half16.txt
