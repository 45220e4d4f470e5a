#!/bin/bash
mkdir -p gpurun_out; rm -f gpurun_out/helpers3.txt
for cfg in "NFISAM_HALF=2 NFISAM_HELPERS=0" "NFISAM_HALF=2 NFISAM_HELPERS=1"; do
echo "== $cfg" >> gpurun_out/helpers3.txt
env $cfg python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "test_chunk_persistent_kernel_is_bit_identical_to_one_launch_per_iteration" 2>&1 | grep -v "^$" | tail -45 >> gpurun_out/helpers3.txt
done
cat gpurun_out/helpers3.txt
