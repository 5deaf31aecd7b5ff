#!/bin/bash
cd "$GRAFT_REPO_ROOT" && export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 900 python3 -m pytest tests/test_lanes_gpu.py tests/test_models_gpu.py tests/test_half_models_gpu.py tests/test_dist_gpu.py tests/test_cli_gpu.py -x -q -m gpu -k "lane or replay or graph or hrnet or rccl_reducer or trainer" > $O/br3_tests.log 2>&1 || { echo tests failed; tail -30 $O/br3_tests.log; exit 1; }
tail -2 $O/br3_tests.log
run() { echo "$* : $(env "$@" timeout -k 10 300 python3 tools/bench_model.py $CFG 2>&1 | grep -a 'ms/step' | tr '\n' ' ' | cut -c1-230)"; }
{
CFG="hrnet 8 512 21 20"
run PSEG_PRECISION=half
run PSEG_PRECISION=fp32
CFG="unet 8 256 2 30"
run PSEG_PRECISION=half
run PSEG_PRECISION=fp32
CFG="deeplabv3plus 16 512 21 10"
run PSEG_PRECISION=half
run PSEG_PRECISION=fp32
} > $O/br3_bench.txt 2>&1
cat $O/br3_bench.txt
