#!/bin/bash
cd "$GRAFT_REPO_ROOT" && export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 900 python3 -m pytest tests/test_dist_gpu.py -x -q -m gpu -k "rccl_reducer" > $O/br5_tests.log 2>&1 || { echo tests failed; tail -40 $O/br5_tests.log; exit 1; }
tail -3 $O/br5_tests.log
