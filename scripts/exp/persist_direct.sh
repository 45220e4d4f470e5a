#!/bin/bash
mkdir -p gpurun_out; rm -f gpurun_out/persist_direct.txt
python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "round6_launch_forms or test_chunk_persistent_kernel_is_bit" 2>&1 | tail -1 >> gpurun_out/persist_direct.txt
NFISAM_PERSIST_DIRECT=1 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "round6_launch_forms or test_chunk_persistent_kernel_is_bit" 2>&1 | tail -1 >> gpurun_out/persist_direct.txt
for rep in 1 2 3; do
for cfg in "NFISAM_PERSIST_DIRECT=0" "NFISAM_PERSIST_DIRECT=1"; do
  echo -n "$cfg | " >> gpurun_out/persist_direct.txt
  env $cfg python bench.py --gpus 1 --steps 20 --warmup 5 --no-regimes --no-cpu-baseline --no-update-bench --no-replicas 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('K=%3d  %.3f us per step  value %.4e  persistent launch %.1f us' % (d['steps'], 1e3*d['ms_per_step'], d['value'], r['kernel_us']))" >> gpurun_out/persist_direct.txt
done; done
for cfg in "NFISAM_PERSIST_DIRECT=0" "NFISAM_PERSIST_DIRECT=1" "NFISAM_PERSIST_DIRECT=0" "NFISAM_PERSIST_DIRECT=1"; do
  echo -n "$cfg | " >> gpurun_out/persist_direct.txt
  env $cfg python scripts/run_plaza1.py 1000 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('Plaza1 total %.3f s fit %.3f s iterations %d' % (d['total_s'], d['fitting_total_s'], d['training_sample_iters']/2000))" >> gpurun_out/persist_direct.txt
done
cat gpurun_out/persist_direct.txt
