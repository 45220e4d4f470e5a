python -m pytest tests/test_hip_parity.py -q -x -k "hidden_widths_between" 2>&1 | tail -40 > gpurun_out/r05_t3_hidden.log
python -m pytest tests/test_hip_parity.py -q -k "busy_device or chunk_persistent_kernel_is_bit or oversubscribed or concurrent_plan" 2>&1 | tail -40 > gpurun_out/r05_t3_persist.log
python -m pytest tests/test_pipeline_gpu.py -q -k "structure_of_every_update" 2>&1 | tail -40 > gpurun_out/r05_t3_structure.log
for p in 1 0; do NFISAM_PERSIST=$p python scripts/time_grad.py 1 4096 6 1; NFISAM_PERSIST=$p python scripts/time_grad.py 1 4096 15 1; done > gpurun_out/r05_t3_n4096.log 2>&1
python scripts/ab.py 3 libnfisam_hip.so libnfisam_hip_prev.so libnfisam_hip.so,NFISAM_PERSIST_TRAFFIC=2 > gpurun_out/r05_t3_ab.log 2>&1
tail -3 gpurun_out/r05_t3_hidden.log gpurun_out/r05_t3_persist.log gpurun_out/r05_t3_structure.log; cat gpurun_out/r05_t3_n4096.log gpurun_out/r05_t3_ab.log
