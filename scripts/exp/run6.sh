python -m pytest tests/test_hip_parity.py -q -k "hidden_dim_16_on_the_two or busy_device" 2>&1 | tail -30 > gpurun_out/r05_t6_tests.log
python -m pytest tests/test_pipeline_gpu.py -q -k "structure_of_every_update" 2>&1 | tail -30 > gpurun_out/r05_t6_structure.log
python bench.py > gpurun_out/r05_bench_6.json 2> gpurun_out/r05_bench_6.err
tail -n 6 gpurun_out/r05_t6_tests.log gpurun_out/r05_t6_structure.log
