#!/bin/bash
# Diagnostic: the complete Manhattan-136 run with every fit through the CPU oracle (scripts/exp/late_rmse.py, ORACLE_FIT=1), two processes side by side
mkdir -p gpurun_out/r05_oracle_fit
(ORACLE_FIT=1 ORACLE_THREADS=4 SEED0=0 python scripts/exp/late_rmse.py 2 2>&1 | grep -v amdgpu.ids | tail -4 > gpurun_out/r05_oracle_fit/a.log) &
(ORACLE_FIT=1 ORACLE_THREADS=4 SEED0=2 python scripts/exp/late_rmse.py 2 2>&1 | grep -v amdgpu.ids | tail -4 > gpurun_out/r05_oracle_fit/b.log) &
wait
cat gpurun_out/r05_oracle_fit/a.log gpurun_out/r05_oracle_fit/b.log
