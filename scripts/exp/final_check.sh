#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu 2>&1 | tail -12 > gpurun_out/r06_gpu_tests_final.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2 >> gpurun_out/r06_gpu_tests_final.txt
python bench.py 2>gpurun_out/r06_bench_final.err | tail -1 > gpurun_out/r06_bench_line_final.json
python bench.py --steps 20 --warmup 5 --no-regimes --no-replicas 2>/dev/null | tail -1 > gpurun_out/r06_bench_line_final_20.json
cat gpurun_out/r06_gpu_tests_final.txt
python - <<'PY'
import json
for f in ("gpurun_out/r06_bench_line_final.json", "gpurun_out/r06_bench_line_final_20.json"):
    d = json.loads(open(f).read().strip().split("\n")[-1])
    print(f, d["ms_per_step"], "%.4e" % d["value"], d["roofline"]["frac"], d["roofline"]["kernel_us"])
PY
