"""Diagnostic (GPU): the bench line's roofline + regimes in one short table (python scripts/regimes.py [bench args])."""
import json, subprocess, sys
out = subprocess.run([sys.executable, "bench.py"] + sys.argv[1:], capture_output=True, text=True).stdout.strip().split("\n")[-1]
d = json.loads(out)
print("C3: %.2f us/step, kernel %.2f us, frac %.4f" % (d["ms_per_step"] * 1e3, d["roofline"]["kernel_us"], d["roofline"]["frac"]))
for k, v in d.get("regimes", {}).items():
    print("%s: %s" % (k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items() if not isinstance(b, (str, dict, list))}))
