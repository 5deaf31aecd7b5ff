#!/bin/bash
# two-stream timeline of the bench step per policy: usage tools/r05_timeline.sh <tag> fp32 half ...
cd "$GRAFT_REPO_ROOT" && export TMPDIR=/tmp
TAG=$1; shift
for POL in "$@"; do
  O=gpurun_out/tl_${TAG}_$POL
  rm -rf $O
  rocprofv3 --kernel-trace --output-format csv -d $O -o p -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --precision $POL --also "" > $O.log 2>&1 || { tail -5 $O.log; exit 1; }
  python3 tools/timeline.py $(find $O -name "*kernel_trace.csv" | head -1) ${PSEG_TL_MARK:-sgd} > gpurun_out/${TAG}_timeline_$POL.txt 2>&1
  rm -rf $O
  echo "timeline $POL done"
done
