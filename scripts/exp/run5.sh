python -m pytest tests/test_hip_parity.py -q -x -k "hidden_dim_16_on_the_two or hidden_dim_4_on_the_two or busy_device or hidden_widths_between or every_kernel_instantiation or multilayer" 2>&1 | tail -40 > gpurun_out/r05_t5_tests.log
TG_H=16 python scripts/time_grad.py 1 4096 6 4 > gpurun_out/r05_t5_c2h16.log 2>&1
TG_H=16 NFISAM_TRAIN=wide python scripts/time_grad.py 1 4096 6 4 >> gpurun_out/r05_t5_c2h16.log 2>&1
TG_H=16 python scripts/time_grad.py 1 2000 8 2 >> gpurun_out/r05_t5_c2h16.log 2>&1
python scripts/time_grad.py 1 4096 6 4 >> gpurun_out/r05_t5_c2h16.log 2>&1
tail -n 5 gpurun_out/r05_t5_tests.log; grep -v amdgpu.ids gpurun_out/r05_t5_c2h16.log
