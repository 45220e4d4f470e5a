#!/bin/bash
# split / default family against the dim-major kernel forced on, over single-clique shapes (gpurun)
for shape in "1 2000 2" "1 2000 3" "1 2000 6" "1 2000 10" "1 2000 15" "1 2000 17" "1 500 6" "1 500 15" "1 1000 9" "2 2000 9" "4 2000 6"; do
  a=$(python scripts/time_grad.py $shape 2>&1 | tail -1)
  b=$(NFISAM_TRAIN=wide NFISAM_DIM_MAJOR_MIN=0 python scripts/time_grad.py $shape 2>&1 | tail -1)
  echo "default : $a"
  echo "dimmajor: $b"
done
