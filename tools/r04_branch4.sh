#!/bin/bash
# per-lane kernel timeline of a replayed configuration.  usage: tools/r04_branch4.sh <policy> <kernels-per-step> <model> <B> <S> <nc>
cd "$GRAFT_REPO_ROOT" && export TMPDIR=/tmp
O=gpurun_out
export PSEG_PRECISION=${1:-half} PSEG_GRAPH=1
M=${3:-hrnet}; B=${4:-8}; S=${5:-512}; NC=${6:-21}
tag=cur
rm -rf $O/lt_$tag
rocprofv3 --kernel-trace -d $O/lt_$tag -o p -- python3 tools/bench_model.py $M $B $S $NC 12 > $O/lt_$tag.log 2>&1 || { echo "trace failed"; tail -5 $O/lt_$tag.log; exit 1; }
grep -a 'ms/step\|lane exec' $O/lt_$tag.log
python3 tools/lane_timeline.py $(find $O/lt_$tag -name "*.db" | head -1) ${2:-1005} > $O/lanes_${M}_$PSEG_PRECISION.txt
rm -rf $O/lt_$tag
cat $O/lanes_${M}_$PSEG_PRECISION.txt
