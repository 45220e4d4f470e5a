#!/bin/bash
mkdir -p gpurun_out; rm -f gpurun_out/lazy.txt
python -m pytest tests/test_pipeline_gpu.py -x -q -m gpu -k "lazy_posterior" 2>&1 | tail -15 >> gpurun_out/lazy.txt
for rep in 1 2; do
for cfg in "LAZY=0" "LAZY=1"; do
  echo -n "$cfg | " >> gpurun_out/lazy.txt
  env $cfg python scripts/run_plaza1.py 1000 2>&1 | grep -v amdgpu.ids | tail -1 >> gpurun_out/lazy.txt
done; done
cat gpurun_out/lazy.txt
