#!/bin/bash
cd "$GRAFT_REPO_ROOT" && export TMPDIR=/tmp
O=gpurun_out
for f in test_bench_path_gpu test_cli_gpu test_dist_gpu test_half_gpu test_half_models_gpu; do
  timeout -k 10 600 python3 -m pytest tests/$f.py tests/test_lanes_gpu.py -x -q -m gpu > $O/bis3_$f.log 2>&1; rc=$?
  echo "$f + lanes: rc=$rc $(tail -1 $O/bis3_$f.log | cut -c1-120)"
done
