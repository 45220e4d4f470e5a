python -m pytest tests/test_hip_parity.py -q -k "hidden_widths_between or busy_device or chunk_persistent_kernel_is_bit or oversubscribed or concurrent_plan or lean_builds" 2>&1 | tail -40 > gpurun_out/r05_t4_tests.log
for p in 1 0; do NFISAM_PERSIST=$p python scripts/time_grad.py 1 4096 6 1; NFISAM_PERSIST=$p python scripts/time_grad.py 1 3000 12 1; done > gpurun_out/r05_t4_n4096.log 2>&1
python scripts/ab.py 4 libnfisam_hip.so libnfisam_hip_prev.so > gpurun_out/r05_t4_ab.log 2>&1
tail -n 4 gpurun_out/r05_t4_tests.log; cat gpurun_out/r05_t4_n4096.log gpurun_out/r05_t4_ab.log | grep -v amdgpu.ids
