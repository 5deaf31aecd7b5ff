#!/bin/bash
cd "$GRAFT_REPO_ROOT" && export TMPDIR=/tmp
O=gpurun_out
S="hr_32 hr_64 hr_128 hr_256 hr_32s2"
run() { echo "== $*"; env "$@" timeout -k 10 200 python3 tools/bench_conv.py fp32 $S 2>&1 | grep -a "wgrad"; }
{
run A=0
run PSEG_WGRAD_BPC=2
run PSEG_WGRAD_BPC=4
run PSEG_WGRAD_NARROW64=1
run PSEG_WGRAD_NARROW64=1 PSEG_WGRAD_BPC=2
} > $O/wg_sweep32.txt 2>&1
cat $O/wg_sweep32.txt
