#!/bin/bash
cd "$GRAFT_REPO_ROOT" && export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest tests/test_lanes_gpu.py -x -q -m gpu 2>&1 | tail -5
