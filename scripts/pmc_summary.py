"""Summarise rocprofv3 --pmc counter_collection CSVs: per kernel, mean counter value per dispatch."""
import csv, glob, sys, collections
for d in sys.argv[1:]:
    for f in glob.glob(d + "/*/*_counter_collection.csv"):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:40]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            if "nsf_" not in k: continue
            print(k)
            for c, v in sorted(cs.items()):
                v2 = [x for x in v if x > 0] or [0]
                print("   %-32s n=%4d mean=%14.1f  max=%14.1f" % (c, len(v), sum(v2) / len(v2), max(v)))
