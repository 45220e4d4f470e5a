#!/bin/bash
mkdir -p gpurun_out
python -m pytest "$@" -x -q -m gpu 2>&1 | tail -30 > gpurun_out/gpu_one_test.txt
cat gpurun_out/gpu_one_test.txt
