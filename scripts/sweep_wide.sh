#!/bin/bash
# A/B sweep of the throughput training kernels' launch knobs on the C3 batch and the 64-clique batch (gpurun)
run() {
  echo "== $*"
  env "$@" python scripts/run_c3.py 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line)
        for k, v in d.items(): print('   %-30s %8.2f us/iter  %6.2f TFLOP/s  loss %.3f' % (k, v['us_per_iteration'], v['tflops'], v['final_loss'][0]))
"
}
run NFISAM_DIM_MAJOR=0
run NFISAM_COND=scalar
run NFISAM_COND=mfma
run NFISAM_COND=mfma NFISAM_TILES_PER_BLOCK=2
run NFISAM_COND=mfma NFISAM_TILES_PER_BLOCK=8
run NFISAM_COND=mfma NFISAM_BIG_W=2
