#!/bin/bash
# round 6: where a real fit's time goes between its chunk-persistent launches: kernel trace (timestamps per dispatch) of the
# first 12 updates of Plaza1, reduced by scripts/exp/chunk_gaps.py
out=$GRAFT_REPO_ROOT/gpurun_out/chunk_gaps
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $out/trace -- python3 $GRAFT_REPO_ROOT/scripts/run_plaza1.py 12 > $out/run.log 2> $out/trace.err
cd $GRAFT_REPO_ROOT
python3 scripts/exp/chunk_gaps.py $out/trace > $out/summary.txt 2>&1
cat $out/summary.txt
rm -rf $out/trace
