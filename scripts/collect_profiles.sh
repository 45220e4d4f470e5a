#!/bin/bash
# Profile collection on the gpurun MI355X box (profiles/rNN_*): kernel-trace stats and PMC passes (each in its own run, never
# combined with tracing) of the bench.py headline workload (C3) and of the 64-clique throughput batch.
# usage: scripts/collect_profiles.sh <out dir under gpurun_out>
out=$GRAFT_REPO_ROOT/$1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --steps 500 --warmup 50 --no-cpu-baseline --no-update-bench"
# counters and per-kernel durations are per LAUNCH: the profiled runs issue every iteration as ONE launch (a plan would
# split big batches into two concurrent launches over disjoint (clique, dim) groups, nfisam_nsf_train_chains); the
# `*_chains_*` trace keeps the default so that the concurrent launches show up as they run in training
export NFISAM_CHAINS=1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/c3_trace -- $B --no-regimes > $out/c3_bench_line.json 2> $out/c3_trace.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/all_trace -- $B > $out/all_bench_line.json 2> $out/all_trace.err
unset NFISAM_CHAINS
rocprofv3 --kernel-trace --stats --output-format csv -d $out/c3_chains_trace -- $B --no-regimes > $out/c3_chains_bench_line.json 2> $out/c3_chains_trace.err
export NFISAM_CHAINS=1
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_SCA" \
           "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES"; do
  rocprofv3 --pmc $set --output-format csv -d $out/c3_pmc$i -- $B --no-regimes > /dev/null 2> $out/c3_pmc$i.err
  rocprofv3 --pmc $set --output-format csv -d $out/b64_pmc$i -- python3 $GRAFT_REPO_ROOT/scripts/run_c3.py scaling > /dev/null 2> $out/b64_pmc$i.err
  rocprofv3 --pmc $set --output-format csv -d $out/c2_pmc$i -- python3 $GRAFT_REPO_ROOT/scripts/run_c3.py c2 > /dev/null 2> $out/c2_pmc$i.err
  i=$((i+1))
done
cd $GRAFT_REPO_ROOT
python3 - $out <<'PY'
import csv, glob, collections, json, sys, os
out = sys.argv[1]
def summarize(prefix):
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in sorted(glob.glob(out + "/%s_pmc*/*/*counter_collection.csv" % prefix)):
        for r in csv.DictReader(open(f)):
            d[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    lines = []
    res = {}
    for k, v in d.items():
        if "nsf_" not in k:
            continue
        lines.append(k)
        res[k] = {}
        for c, x in sorted(v.items()):
            # the timed region = the last 500 (bench) / 300 (run_c3) launches; warm-up launches of other shapes are excluded by taking the modal value count
            lines.append("   %-26s n=%5d mean=%16.1f max=%16.1f" % (c, len(x), sum(x) / len(x), max(x)))
            res[k][c] = sum(x) / len(x)
    open(out + "/%s_pmc_summary.txt" % prefix, "w").write("\n".join(lines) + "\n")
    return res
c3 = summarize("c3")
b64 = summarize("b64")
c2 = summarize("c2")
json.dump(dict(c3=c3, b64=b64, c2=c2), open(out + "/pmc_means.json", "w"), indent=1)
for pre in ("c3_trace", "all_trace", "c3_chains_trace"):
    for f in glob.glob(out + "/%s/*/*kernel_stats.csv" % pre):
        os.system("cp %s %s/%s_kernel_stats.csv" % (f, out, pre))
os.system("rm -rf %s/*_pmc[0-9] %s/c3_trace %s/all_trace %s/c3_chains_trace" % (out, out, out, out))
PY
# the derived summaries bench.py reads (`roofline.traffic`, `roofline.issue_profiled`): <out>/derived_train_kernel_traffic.json and
# <out>/derived_issue_utilisation.json, stamped with the time of THIS collection; copy them to profiles/rNN_* to publish them
python3 scripts/derive_profile_json.py $out $out/derived
ls -la $out
