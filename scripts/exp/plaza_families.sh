#!/bin/bash
# round 6: Plaza1 end to end, 64-particle family (NFISAM_HALF=1: two lanes only up to eight copies) against the default (sixteen copies, helper waves), six seeds each
mkdir -p gpurun_out; rm -f gpurun_out/plaza_families.txt
for seed in 0 1 2 3 4 5; do
  for cfg in "NFISAM_HALF=1" "NFISAM_HALF=2"; do
    echo -n "seed $seed $cfg | " >> gpurun_out/plaza_families.txt
    env SEED=$seed $cfg python scripts/run_plaza1.py 1000 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('total %.3f s fit %.3f s iterations %d samples/s %.4e' % (d['total_s'], d['fitting_total_s'], d['training_sample_iters']/2000, d['flow_training_samples_per_s']))" >> gpurun_out/plaza_families.txt
  done
done
cat gpurun_out/plaza_families.txt
