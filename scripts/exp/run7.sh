python -m pytest tests/test_hip_parity.py -q -x -k "hidden_dim_16_on_the_two" 2>&1 | grep -v amdgpu.ids | tail -60 > gpurun_out/r05_t7_h16.log
python -m pytest tests/test_hip_parity.py -q -k "busy_device" 2>&1 | grep -v amdgpu.ids | tail -30 > gpurun_out/r05_t7_probe.log
TG_H=16 python scripts/time_grad.py 1 4096 6 4 > gpurun_out/r05_t7_c2h16.log 2>&1
cat gpurun_out/r05_t7_h16.log | head -70; tail -n 8 gpurun_out/r05_t7_probe.log; grep -v amdgpu.ids gpurun_out/r05_t7_c2h16.log
