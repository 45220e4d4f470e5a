#!/bin/bash
# Round 6: what profiles/r06_* is made from, in one gpurun call (the stamps library is built in the container and shipped:
# nf-isam_amd/csrc/_stamps3/libnfisam_hip_stamps3.so, `make OBJDIR=_stamps3/obj OUT=_stamps3/libnfisam_hip_stamps3.so EXTRA=-DNSF_STAMPS=3`)
out=$1
mkdir -p $out
python bench.py 2>/dev/null | tail -1 > $out/bench_line.json
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $out/bench_line_driver_style_20_steps.json
bash scripts/collect_profiles.sh $out > $out/collect.log 2>&1
python scripts/run_plaza1.py 100000 $out/plaza1_end_to_end.json > $out/plaza1.log 2>&1
DATASET=Plaza1ADA0.4EFG python scripts/run_plaza1.py 100000 $out/plaza1_ada04_end_to_end.json > $out/plaza1_ada04.log 2>&1
DATASET=Plaza2EFG python scripts/run_plaza1.py 100000 $out/plaza2_end_to_end.json > $out/plaza2.log 2>&1
DATASET=Manhattan200 STEP=1 ITERS=500 TOL=1e-9 python scripts/run_plaza1.py 100000 $out/manhattan200_end_to_end.json > $out/manhattan200.log 2>&1
REPLICAS=8 python scripts/run_plaza1.py 100000 $out/plaza1_replicas8.json > $out/plaza1_replicas8.log 2>&1
python scripts/pipeline_report.py $out/pipeline_parity_vs_reference.json > $out/pipeline.log 2>&1
BENCH_DIST_BACKEND=gloo python bench.py --gpus 2 --steps 20 --warmup 5 2>/dev/null | tail -1 > $out/bench_line_two_gloo_ranks_on_one_gpu.json
# the new kernel family and the helper waves: phase stamps of one Plaza clique and of a 1000-particle clique
{
for cfg in "NFISAM_HALF=0 NFISAM_LONE_LEAN=0" "NFISAM_HALF=0 NFISAM_HELPERS=0" "NFISAM_HALF=0" "NFISAM_HALF=2 NFISAM_HELPERS=0" "NFISAM_HALF=2"; do
  echo "== $cfg : scripts/stamps3.py 1 2000 15 persist"
  env $cfg python scripts/stamps3.py 1 2000 15 persist 2>&1 | grep -v amdgpu.ids | head -18
done
for cfg in "NFISAM_HALF=0 NFISAM_HELPERS=0" "NFISAM_HALF=1 NFISAM_HELPERS=0" "NFISAM_HALF=1"; do
  echo "== $cfg : scripts/stamps3.py 1 1000 15 persist"
  env $cfg python scripts/stamps3.py 1 1000 15 persist 2>&1 | grep -v amdgpu.ids | head -18
done
} > $out/phase_cycles_half_vs_64.txt 2>&1
bash scripts/exp/half_sweep.sh > /dev/null 2>&1; cp gpurun_out/half_sweep.txt $out/half_sweep.txt
{
echo "== one Plaza clique (n = 2000, D = 15), us per iteration: scripts/time_grad.py 1 2000 15"
for cfg in "NFISAM_HALF=0 NFISAM_LONE_LEAN=0" "NFISAM_HALF=0 NFISAM_LONE_LEAN=0 NFISAM_PERSIST_SPLIT=1" "NFISAM_HALF=0 NFISAM_HELPERS=0" "NFISAM_HALF=0" "NFISAM_HALF=2 NFISAM_HELPERS=0" "NFISAM_HALF=2" "NFISAM_HALF=2 NFISAM_HALF_W=8"; do
  echo -n "$cfg | "; env $cfg python scripts/time_grad.py 1 2000 15 2>&1 | grep -v amdgpu.ids
done
echo "== the launch's fixed cost: scripts/exp/lone_launch_sweep.py (two-wave build with helper waves, without, then NFISAM_LONE_LEAN=0)"
python scripts/exp/lone_launch_sweep.py 2>&1 | grep -v amdgpu.ids
NFISAM_HELPERS=0 python scripts/exp/lone_launch_sweep.py 2>&1 | grep -v amdgpu.ids | tail -2
NFISAM_LONE_LEAN=0 python scripts/exp/lone_launch_sweep.py 2>&1 | grep -v amdgpu.ids | tail -2
} > $out/lone_clique_variants.txt 2>&1
LAZY=1 python scripts/run_plaza1.py 100000 $out/plaza1_end_to_end_lazy_posterior.json > $out/plaza1_lazy.log 2>&1
bash scripts/exp/plaza_families.sh > /dev/null 2>&1; cp gpurun_out/plaza_families.txt $out/plaza1_families_six_seeds.txt
bash scripts/exp/short_plan_sweep.sh > $out/short_plan_sweep.txt 2>&1
bash scripts/exp/chunk_gaps.sh > /dev/null 2>&1; cp gpurun_out/chunk_gaps/summary.txt $out/chunk_gaps_plaza1_first_updates.txt
tail -n 2 $out/plaza1.log $out/plaza1_lazy.log $out/plaza1_ada04.log $out/plaza2.log $out/manhattan200.log $out/plaza1_replicas8.log $out/pipeline.log | cut -c1-300
