#!/bin/bash
# kernel trace of a replayed small configuration: kernel-time sum, launches, busy span.  usage: tools/small_trace.sh hrnet 8 512 21 half
cd "$GRAFT_REPO_ROOT" && export TMPDIR=/tmp
M=$1; B=$2; S=$3; NC=$4; POL=$5
O=gpurun_out/trace_${M}_${POL}
rm -rf $O
export PSEG_PRECISION=$POL PSEG_GRAPH=1
rocprofv3 --kernel-trace --stats -d $O -o p -- python3 tools/bench_model.py $M $B $S $NC 12 > $O.log 2>&1 || { tail -5 $O.log; exit 1; }
grep -a 'ms/step' $O.log
python3 tools/prof_summary.py $(find $O -name "*.db" | head -1) 12 $O.csv > $O.md
head -40 $O.md
