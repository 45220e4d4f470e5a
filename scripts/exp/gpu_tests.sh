#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu 2>&1 | tail -40 > gpurun_out/gpu_tests.txt
cat gpurun_out/gpu_tests.txt
