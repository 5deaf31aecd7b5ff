#!/bin/bash
cd "$GRAFT_REPO_ROOT" && export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests/test_dist_gpu.py tests/test_cli_gpu.py -x -q -m gpu 2>&1 | tail -6
