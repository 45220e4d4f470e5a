#!/bin/bash
mkdir -p gpurun_out
python scripts/run_plaza1.py 1000 gpurun_out/plaza1_a.json 2>&1 | grep -v amdgpu.ids | tail -4
python scripts/run_plaza1.py 1000 gpurun_out/plaza1_b.json 2>&1 | grep -v amdgpu.ids | tail -1
TOP=60 python scripts/profile_host.py 156 2>&1 | grep -v amdgpu.ids > gpurun_out/plaza1_host_profile.txt
python scripts/profile_posterior.py 2>&1 | grep -v amdgpu.ids | head -60 > gpurun_out/plaza1_post_profile.txt
head -75 gpurun_out/plaza1_host_profile.txt
