#!/bin/bash
# round 6: four helper waves per block (staging only) in the lone-clique two-wave builds
mkdir -p gpurun_out; rm -f gpurun_out/helpers.txt
for rep in 1 2; do
for shape in "2000 15" "2000 8" "1000 15" "600 12" "1500 15"; do
  for cfg in "NFISAM_HELPERS=0" "NFISAM_HELPERS=1"; do
    echo -n "$cfg | " >> gpurun_out/helpers.txt
    env $cfg python scripts/time_grad.py 1 $shape 2>&1 | grep -v amdgpu.ids >> gpurun_out/helpers.txt
  done
done
done
python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "persist or twin or identical or chunk or span" 2>&1 | tail -3 >> gpurun_out/helpers.txt
cat gpurun_out/helpers.txt
