TG_H=16 python scripts/time_grad.py 1 4096 6 4 2>&1 | grep -v amdgpu.ids
TG_H=16 NFISAM_PAIR_STASH=0 python scripts/time_grad.py 1 4096 6 4 2>&1 | grep -v amdgpu.ids
TG_H=16 python scripts/time_grad.py 1 2000 8 2 2>&1 | grep -v amdgpu.ids
python -m pytest tests/test_hip_parity.py -q -k "hidden_dim_16_on_the_two or busy_device or multilayer or hidden_widths_between or every_kernel" 2>&1 | grep -v amdgpu.ids | tail -8
REPLICAS=8 python scripts/run_plaza1.py 100000 gpurun_out/r05_replicas8_b.json 2>&1 | tail -2 | cut -c1-200
