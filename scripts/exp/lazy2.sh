#!/bin/bash
mkdir -p gpurun_out; rm -f gpurun_out/lazy2.txt
for cfg in "LAZY=0" "LAZY=1" "LAZY=1"; do
  echo -n "$cfg | " >> gpurun_out/lazy2.txt
  env $cfg python scripts/run_plaza1.py 1000 2>&1 | grep -v amdgpu.ids | tail -1 >> gpurun_out/lazy2.txt
done
cat gpurun_out/lazy2.txt
