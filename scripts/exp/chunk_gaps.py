"""Reduce a rocprofv3 --kernel-trace CSV of a training run to: per kernel name the count / mean duration, and the idle gap on the
device in front of every launch (start - previous end), summed per kernel name.   argv: trace dir"""
import csv, glob, sys, collections
rows = []
for f in glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
dur, gap, cnt = collections.Counter(), collections.Counter(), collections.Counter()
prev_end = None
for s, e, k in rows:
    name = k.split("(")[0][:70]
    dur[name] += e - s; cnt[name] += 1
    if prev_end is not None and s - prev_end < 200000:            # (gaps beyond 0.2 ms are host phases between fits, not chunk boundaries)
        gap[name] += max(0, s - prev_end)
    prev_end = max(prev_end or e, e)
tot = sum(dur.values())
print("%-72s %7s %10s %10s %12s" % ("kernel", "calls", "mean us", "total ms", "gap before us (mean)"))
for name, d in dur.most_common(14):
    print("%-72s %7d %10.2f %10.2f %12.2f" % (name, cnt[name], d / cnt[name] / 1e3, d / 1e6, gap[name] / cnt[name] / 1e3))
