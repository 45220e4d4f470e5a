#!/bin/bash
# counters of the throughput training kernels on the 64-clique batch: $1 = output dir, rest = env assignments
out=$1; shift
cd /tmp && export TMPDIR=/tmp
for set in "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_WAVES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  env "$@" rocprofv3 --pmc $set --output-format csv -d $out/p$((i++)) -- python3 $GRAFT_REPO_ROOT/scripts/run_c3.py scaling > /dev/null 2>&1
done
cd $GRAFT_REPO_ROOT
python3 - $out <<PY
import csv, glob, collections, sys
d=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1]+"/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"][:34]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in d.items():
    if "nsf_train" in k:
        print(k)
        for c,x in sorted(v.items()): print("   %-24s %14.0f  (n=%d)" % (c, sum(x)/len(x), len(x)))
PY
