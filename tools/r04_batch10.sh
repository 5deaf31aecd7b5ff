#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_half_gpu.py -q -x -k "conv2d_fwd_half or dgrad_wgrad_half or c3_shapes_half" > gpurun_out/r04_b10_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -2 gpurun_out/r04_b10_tests.log | cut -c1-160
[ $rc -ne 0 ] && { grep -n "Error\|assert " gpurun_out/r04_b10_tests.log | head; exit 1; }
timeout -k 10 300 python tools/bench_conv_half.py > gpurun_out/r04_b10_bch.log 2>&1; tail -1 gpurun_out/r04_b10_bch.log
for rep in 1 2; do timeout -k 10 300 python bench.py --precision half --also "" --no-cpu-baseline --no-roofline --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('half ms/step', d['ms_per_step'])"; done
for st in 3; do echo "wgrad stages=$st: $(PSEG_HWGRAD_STAGES=$st timeout -k 10 300 python bench.py --precision half --also "" --no-cpu-baseline --no-roofline --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print(d['ms_per_step'])")"; done
echo "hrnet: $(PSEG_PRECISION=half python3 tools/bench_model.py hrnet 8 512 21 20 2>&1 | grep -a 'ms/step')"
