#!/bin/bash
# round 6: single cliques, us per iteration -- the 64-particle family (three-wave build / two-wave "roomy" build / + helper waves) against two
# lanes per particle (up to eight copies without and with helper waves; NFISAM_HALF=2: up to sixteen copies)
mkdir -p gpurun_out; rm -f gpurun_out/half_sweep.txt
for shape in "500 11" "600 12" "1000 15" "1024 12" "1500 15" "2000 8" "2000 15" "2048 16"; do
  for cfg in "NFISAM_HALF=0 NFISAM_LONE_LEAN=0" "NFISAM_HALF=0 NFISAM_HELPERS=0" "NFISAM_HALF=0" "NFISAM_HALF=1 NFISAM_HELPERS=0" "NFISAM_HALF=1" "NFISAM_HALF=2"; do
    echo -n "$cfg | " >> gpurun_out/half_sweep.txt
    env $cfg python scripts/time_grad.py 1 $shape 2>&1 | grep -v amdgpu.ids >> gpurun_out/half_sweep.txt
  done
done
cat gpurun_out/half_sweep.txt
