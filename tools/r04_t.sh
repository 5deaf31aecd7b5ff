#!/bin/bash
cd "$GRAFT_REPO_ROOT" && export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests/test_bench_path_gpu.py -x -q -m gpu 2>&1 | tail -8
