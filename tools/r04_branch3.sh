#!/bin/bash
cd "$GRAFT_REPO_ROOT" && export TMPDIR=/tmp
O=gpurun_out
run() { echo "$* : $(env "$@" PSEG_GRAPH=1 timeout -k 10 300 python3 tools/bench_model.py $CFG 2>&1 | grep -a 'ms/step\|lane exec')"; }
{
CFG="hrnet 8 512 21 20"
run PSEG_PRECISION=half
run PSEG_PRECISION=half PSEG_LANES_OWN_STREAMS=1
run PSEG_PRECISION=half PSEG_BRANCH_STREAMS=2
run PSEG_PRECISION=half PSEG_BRANCH_STREAMS=3
run PSEG_PRECISION=half PSEG_BRANCH_STREAMS=0
run PSEG_PRECISION=fp32
run PSEG_PRECISION=fp32 PSEG_BRANCH_STREAMS=2
run PSEG_PRECISION=fp32 PSEG_BRANCH_STREAMS=3
run PSEG_PRECISION=fp32 PSEG_BRANCH_STREAMS=0
run PSEG_PRECISION=mixed
run PSEG_PRECISION=limb
} > $O/br3_bench.txt 2>&1
cat $O/br3_bench.txt
