#!/bin/bash
for fam in default split; do
  for s in 0 1 2 3 4 5 6 7; do
    if [ $fam = split ]; then export NFISAM_TRAIN=split; else unset NFISAM_TRAIN; fi
    r=$(SEED=$s python scripts/run_plaza1.py 2>&1 | grep "update 155" | sed 's/.*traj RMSE //')
    echo "$fam seed $s final RMSE $r"
  done
done
