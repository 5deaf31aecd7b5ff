#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_half_gpu.py -q -x -k "dgrad_wgrad_half or c3_shapes_half" > gpurun_out/r04_b5_tests.log 2>&1; echo "wgrad tests rc=$?"; tail -2 gpurun_out/r04_b5_tests.log
timeout -k 10 600 python -m pytest tests/test_models_gpu.py -q -x -s -k "batch16 or deeplabv3plus_full_model" > gpurun_out/r04_b5_b16.log 2>&1; echo "b16 rc=$?"; grep -a "full model \[" gpurun_out/r04_b5_b16.log; tail -2 gpurun_out/r04_b5_b16.log
timeout -k 10 300 python tools/bench_conv_half.py > gpurun_out/r04_b5_bch.log 2>&1; tail -1 gpurun_out/r04_b5_bch.log
timeout -k 10 300 python bench.py --precision half --also "" --no-cpu-baseline --no-roofline --steps 20 --warmup 5 > gpurun_out/r04_b5_half.json 2> gpurun_out/r04_b5_half.err
python -c "import json;d=json.load(open('gpurun_out/r04_b5_half.json'));print('half ms/step', d['ms_per_step'])"
