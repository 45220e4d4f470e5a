out=$GRAFT_REPO_ROOT/gpurun_out/c2pmc
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export NFISAM_CHAINS=1
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_SCA" \
           "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES"; do
  rocprofv3 --pmc $set --output-format csv -d $out/c2_pmc$i -- python3 $GRAFT_REPO_ROOT/scripts/run_c3.py c2 > /dev/null 2> $out/c2_pmc$i.err
  i=$((i+1))
done
cd $GRAFT_REPO_ROOT
python3 - $out <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
d = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(out + "/c2_pmc*/*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
lines = []
for k, v in d.items():
    if "nsf_" not in k:
        continue
    lines.append(k)
    for c, x in sorted(v.items()):
        lines.append("   %-26s n=%5d mean=%16.1f max=%16.1f" % (c, len(x), sum(x) / len(x), max(x)))
open(out + "/c2_pmc_summary.txt", "w").write("\n".join(lines) + "\n")
import os
os.system("rm -rf %s/c2_pmc[0-9]" % out)
PY
grep -A26 "nsf_train3" $out/c2_pmc_summary.txt | grep "SQ_WAIT_ANY\|SQ_WAVE_CYCLES\|SQ_BUSY"
