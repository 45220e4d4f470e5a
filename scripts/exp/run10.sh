NFISAM_PROBE_DEBUG=1 python -m pytest tests/test_hip_parity.py -q -k "busy_device" 2>&1 | grep -v amdgpu.ids | tail -30 > gpurun_out/r05_t10_probe.log
python -m pytest tests/test_hip_parity.py -q -k "hidden_dim_16_on_the_two" 2>&1 | grep -v amdgpu.ids | grep -B5 -A25 "^E " | head -150 > gpurun_out/r05_t10_h16.log
cat gpurun_out/r05_t10_probe.log | cut -c1-600 | tail -15; cat gpurun_out/r05_t10_h16.log | cut -c1-250
