#!/bin/bash
cd "$GRAFT_REPO_ROOT" && export TMPDIR=/tmp
O=gpurun_out
export PSEG_PRECISION=half PSEG_GRAPH=1
tr() {  # tag, env...
  tag=$1; shift
  rm -rf $O/lt_$tag
  ( export "$@"; rocprofv3 --kernel-trace -d $O/lt_$tag -o p -- python3 tools/bench_model.py hrnet 8 512 21 12 > $O/lt_$tag.log 2>&1 ) || { echo "trace $tag failed"; tail -5 $O/lt_$tag.log; exit 1; }
  echo "== $tag: $* : $(grep -a 'ms/step' $O/lt_$tag.log)"
  python3 tools/lane_timeline.py $(find $O/lt_$tag -name "*.db" | head -1) 1005
  rm -rf $O/lt_$tag
}
{
tr A GPU_MAX_HW_QUEUES=4 PSEG_BRANCH_STREAMS=3
tr B GPU_MAX_HW_QUEUES=8 PSEG_BRANCH_STREAMS=3
tr C GPU_MAX_HW_QUEUES=8 PSEG_BRANCH_STREAMS=3 PSEG_LANES_OWN_STREAMS=1
tr D GPU_MAX_HW_QUEUES=4 PSEG_BRANCH_STREAMS=3 PSEG_LANES_OWN_STREAMS=1
tr E GPU_MAX_HW_QUEUES=4 PSEG_BRANCH_STREAMS=0
} > $O/br4.txt 2>&1
cat $O/br4.txt
