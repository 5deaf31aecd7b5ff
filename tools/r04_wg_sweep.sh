#!/bin/bash
cd "$GRAFT_REPO_ROOT" && export TMPDIR=/tmp
O=gpurun_out
export PSEG_PRECISION=half
S="hr_32 hr_64 hr_128 hr_256 hr_32s2 l1_3x3 l1_1x1a"
run() { echo "== $*"; env "$@" timeout -k 10 200 python3 tools/bench_conv_half.py $S 2>&1 | grep -a "wgrad" | sed 's/GF floor.*| wgrad/| wgrad/'; }
{
run A=0
run PSEG_HWGRAD_STAGES=3
run PSEG_WGRAD_BPC=2
run PSEG_WGRAD_BPC=2 PSEG_HWGRAD_STAGES=3
run PSEG_WGRAD_BPC=4
run PSEG_WGRAD_BPC=4 PSEG_HWGRAD_BKP=32 PSEG_HWGRAD_STAGES=4
run PSEG_WGRAD_BPC=2 PSEG_HWGRAD_BKP=32 PSEG_HWGRAD_STAGES=4
} > $O/wg_sweep.txt 2>&1
cat $O/wg_sweep.txt
