#!/bin/bash
cd "$GRAFT_REPO_ROOT" && export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_ops_gpu.py tests/test_half_gpu.py tests/test_half_models_gpu.py -x -q -m gpu -k "bn or batchnorm or norm or bitmask or block or strict or model" > $O/bn_tests.log 2>&1 || { echo tests failed; tail -30 $O/bn_tests.log; exit 1; }
tail -1 $O/bn_tests.log
export PSEG_PRECISION=half
for m in 1; do echo "rows=$m"; PSEG_BN_FWD_ROWS=$m timeout -k 10 300 python3 tools/bench_bn_half.py 2>&1 | grep -v amdgpu.ids | cut -c1-112; done > $O/bn_bench.txt; cat $O/bn_bench.txt
for m in 1 0 1 0 1; do PSEG_BN_FWD_ROWS=$m python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --precision half --also "" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rows=$m half step', d['ms_per_step'])"; done
