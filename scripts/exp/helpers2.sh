#!/bin/bash
mkdir -p gpurun_out; rm -f gpurun_out/helpers2.txt
for rep in 1 2; do
for shape in "2000 15" "2000 8" "1500 15" "2000 12"; do
  for cfg in "NFISAM_HALF=0" "NFISAM_HALF=2"; do
    echo -n "$cfg | " >> gpurun_out/helpers2.txt
    env $cfg python scripts/time_grad.py 1 $shape 2>&1 | grep -v amdgpu.ids >> gpurun_out/helpers2.txt
  done
done
done
python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "persist or twin or identical or chunk or span" 2>&1 | tail -3 >> gpurun_out/helpers2.txt
NFISAM_HALF=2 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "persist or twin or identical or chunk" 2>&1 | tail -3 >> gpurun_out/helpers2.txt
cat gpurun_out/helpers2.txt
