#!/bin/bash
# kernel table of the DeepLabV3+ bench step under one policy (one stream): usage tools/dl_trace.sh half|fp32 [overlap 0|1]
cd "$GRAFT_REPO_ROOT" && export TMPDIR=/tmp
POL=${1:-half}; OV=${2:-0}
O=gpurun_out/dltrace_${POL}_$OV
rm -rf $O
export PSEG_OVERLAP_WGRAD=$OV
rocprofv3 --kernel-trace --stats -d $O -o p -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --precision $POL --also "" > $O.log 2>&1 || { tail -5 $O.log; exit 1; }
python3 tools/prof_summary.py $(find $O -name "*.db" | head -1) 7 $O.csv > $O.md
head -34 $O.md
