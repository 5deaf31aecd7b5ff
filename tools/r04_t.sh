#!/bin/bash
cd "$GRAFT_REPO_ROOT" && export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests/test_bench_path_gpu.py tests/test_ops_gpu.py tests/test_models_gpu.py -x -q -m gpu -k "ce_ or cross_entropy or compute_loss or loss or deeplab" 2>&1 | tail -3
timeout -k 10 200 python3 tools/bench_misc.py 2>&1 | grep -a "ce_" 
timeout -k 10 200 python3 tools/ce_time.py 2>&1 | tail -4
