#!/bin/bash
cd "$GRAFT_REPO_ROOT" && export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest tests/test_half_gpu.py -x -q -m gpu -k "bitmask or batchnorm" 2>&1 | tail -15
