#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/gpu_tests.txt
NFISAM_HALF=2 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "persist or twin or identical or chunk" 2>&1 | tail -3 > gpurun_out/gpu_tests_half2.txt
bash scripts/exp/headline.sh > /dev/null 2>&1
python scripts/run_plaza1.py 1000 gpurun_out/plaza1_a.json 2>&1 | grep -v amdgpu.ids | tail -2 > gpurun_out/plaza_now.txt
python scripts/run_plaza1.py 1000 gpurun_out/plaza1_b.json 2>&1 | grep -v amdgpu.ids | tail -1 >> gpurun_out/plaza_now.txt
cat gpurun_out/gpu_tests.txt gpurun_out/gpu_tests_half2.txt gpurun_out/headline.txt gpurun_out/plaza_now.txt
