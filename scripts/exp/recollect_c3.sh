#!/bin/bash
# the round's last binary: the two bench lines + scripts/collect_profiles.sh (kernel traces, --pmc passes, derived JSONs) again
out=gpurun_out/r06b
mkdir -p $out
python bench.py 2>/dev/null | tail -1 > $out/bench_line.json
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $out/bench_line_driver_style_20_steps.json
bash scripts/collect_profiles.sh $out > $out/collect.log 2>&1
head -4 $out/c3_trace_kernel_stats.csv | cut -c1-220
python - <<'PY'
import json
for f in ("bench_line.json", "bench_line_driver_style_20_steps.json", "c3_bench_line.json"):
    d = json.loads(open("gpurun_out/r06b/" + f).read().strip().split("\n")[-1])
    print(f, d["ms_per_step"], "%.4e" % d["value"], d["roofline"]["frac"], d["roofline"]["kernel_us"])
PY
