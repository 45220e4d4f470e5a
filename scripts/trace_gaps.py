"""Summarise a rocprofv3 kernel_trace CSV: per-kernel durations and the gaps between consecutive dispatches."""
import csv, glob, sys, collections
for f in glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv"):
    rows = [r for r in csv.DictReader(open(f))]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = [r for r in rows if "nsf_" in r["Kernel_Name"]]
    rows = rows[len(rows) // 5:]          # skip warm-up
    dur = collections.defaultdict(list); gap = collections.defaultdict(list)
    for a, b in zip(rows[:-1], rows[1:]):
        kind = lambda r: "train" if "train" in r["Kernel_Name"] else ("book" if "bookkeep" in r["Kernel_Name"] else "adam")
        ka, kb = kind(a), kind(b)
        dur[ka].append(int(a["End_Timestamp"]) - int(a["Start_Timestamp"]))
        gap[ka + "->" + kb].append(int(b["Start_Timestamp"]) - int(a["End_Timestamp"]))
    med = lambda v: sorted(v)[len(v) // 2] / 1e3
    print(f)
    for k, v in dur.items(): print("  duration %-6s median %.2f us  (n=%d, min %.2f)" % (k, med(v), len(v), min(v) / 1e3))
    for k, v in gap.items(): print("  gap %-12s median %.2f us  (min %.2f)" % (k, med(v), min(v) / 1e3))
    n_it = sum(1 for r in rows if "train" in r["Kernel_Name"])
    t = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e3 / max(n_it, 1)
    print("  per iteration: %.2f us" % t)
