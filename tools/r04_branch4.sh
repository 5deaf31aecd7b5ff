#!/bin/bash
cd "$GRAFT_REPO_ROOT" && export TMPDIR=/tmp
O=gpurun_out
export PSEG_PRECISION=${1:-half} PSEG_GRAPH=1
tag=cur
rm -rf $O/lt_$tag
rocprofv3 --kernel-trace -d $O/lt_$tag -o p -- python3 tools/bench_model.py hrnet 8 512 21 12 > $O/lt_$tag.log 2>&1 || { echo "trace failed"; tail -5 $O/lt_$tag.log; exit 1; }
grep -a 'ms/step\|lane exec' $O/lt_$tag.log
python3 tools/lane_timeline.py $(find $O/lt_$tag -name "*.db" | head -1) 1005 > $O/br4_$PSEG_PRECISION.txt
rm -rf $O/lt_$tag
cat $O/br4_$PSEG_PRECISION.txt
