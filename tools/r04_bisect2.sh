#!/bin/bash
# usage: tools/r04_bisect2.sh <tag> [ENV=VAL ...]   -- the suite up to and including test_lanes_gpu.py under the given environment
cd "$GRAFT_REPO_ROOT" && export TMPDIR=/tmp
O=gpurun_out; T=$1; shift
for kv in "$@"; do export "$kv"; done
timeout -k 10 900 python3 -m pytest tests/test_abi_cpu.py tests/test_bench_path_gpu.py tests/test_cli_gpu.py tests/test_dist_gpu.py tests/test_half_gpu.py tests/test_half_models_gpu.py tests/test_lanes_gpu.py -x -q -m gpu > $O/bis_$T.log 2>&1; rc=$?
head -3 $O/bis_$T.log | cut -c1-200; tail -2 $O/bis_$T.log | cut -c1-200; echo "rc=$rc"
