#!/bin/bash
# Round 5: rocprofv3 kernel-trace of the two kernels that are new this round (the WIDE persistent instantiation: n = 4096; the
# hidden_dim 16 two-dims-per-wave kernel: C2's shape) -- profiles/r05_new_kernels_kernel_stats.csv
out=$GRAFT_REPO_ROOT/gpurun_out/r05_newk
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/wide -- python3 $GRAFT_REPO_ROOT/scripts/time_grad.py 1 4096 6 1 > $out/wide.log 2> $out/wide.err
export TG_H=16
rocprofv3 --kernel-trace --stats --output-format csv -d $out/h16 -- python3 $GRAFT_REPO_ROOT/scripts/time_grad.py 1 4096 6 4 > $out/h16.log 2> $out/h16.err
cd $GRAFT_REPO_ROOT
for d in wide h16; do f=$(ls $out/$d/*/*kernel_stats.csv | head -1); echo "== $d: $(grep -v amdgpu $out/$d.log | tail -1)"; grep "nsf_" $f | cut -d, -f1-4 | cut -c1-160; cp $f $out/${d}_kernel_stats.csv; done
rm -rf $out/wide $out/h16
