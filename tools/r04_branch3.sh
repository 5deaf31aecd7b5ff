#!/bin/bash
cd "$GRAFT_REPO_ROOT" && export TMPDIR=/tmp
O=gpurun_out
run() { echo "$* : $(env "$@" PSEG_GRAPH=1 timeout -k 10 300 python3 tools/bench_model.py $CFG 2>&1 | grep -a 'ms/step' | tr '\n' ' ')"; }
{
CFG="hrnet 8 512 21 20"
run PSEG_PRECISION=half
run PSEG_PRECISION=half PSEG_BRANCH_STREAMS=3
run PSEG_PRECISION=half PSEG_BRANCH_STREAMS=1
run PSEG_PRECISION=half PSEG_BRANCH_STREAMS=0
run PSEG_PRECISION=fp32
run PSEG_PRECISION=fp32 PSEG_BRANCH_STREAMS=0
CFG="unet 8 256 2 30"
run PSEG_PRECISION=half
run PSEG_PRECISION=fp32
CFG="deeplabv3plus 16 512 21 10"
run PSEG_PRECISION=half PSEG_GRAPH=0
} > $O/br3_bench.txt 2>&1
cat $O/br3_bench.txt
