#!/bin/bash
cd "$GRAFT_REPO_ROOT" && export TMPDIR=/tmp
O=gpurun_out
run() { echo "$* : $(env "$@" PSEG_GRAPH=1 timeout -k 10 300 python3 tools/bench_model.py $CFG 2>&1 | grep -a 'ms/step\|lane exec' | tr '\n' ' ')"; }
{
CFG="unet 8 256 2 30"
run PSEG_PRECISION=half PSEG_BN_SMALL_GRAPH=0
run PSEG_PRECISION=half PSEG_BN_SMALL_GRAPH=1
run PSEG_PRECISION=fp32 PSEG_BN_SMALL_GRAPH=0
run PSEG_PRECISION=fp32 PSEG_BN_SMALL_GRAPH=1
CFG="hrnet 8 512 21 20"
run PSEG_PRECISION=half PSEG_BN_SMALL_GRAPH=0
run PSEG_PRECISION=half PSEG_BN_SMALL_GRAPH=1
run PSEG_PRECISION=fp32 PSEG_BN_SMALL_GRAPH=1
} > $O/br3_bench.txt 2>&1
cat $O/br3_bench.txt
