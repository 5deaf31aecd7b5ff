#!/bin/bash
cd "$GRAFT_REPO_ROOT" && export TMPDIR=/tmp
O=gpurun_out
export PSEG_PRECISION=half
timeout -k 10 600 python3 -m pytest tests/test_ops_gpu.py tests/test_half_gpu.py -x -q -m gpu -k "bn or batchnorm or norm" > $O/bn_tests.log 2>&1 || { echo tests failed; tail -30 $O/bn_tests.log; exit 1; }
tail -1 $O/bn_tests.log
timeout -k 10 300 python3 tools/bench_bn_half.py 2>&1 | grep -v amdgpu.ids > $O/bn_bench.txt; cat $O/bn_bench.txt
for i in 1 2; do python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --precision half --also "" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('half step', d['ms_per_step'])"; done
