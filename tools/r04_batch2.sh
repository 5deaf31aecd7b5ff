#!/bin/bash
# round 4, GPU batch 2: the bookkeeping-light gather_h loop -- correctness, per-shape times, step time
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
python -m pytest tests/test_half_gpu.py -q -x -k "conv2d_fwd_half or dgrad_wgrad_half or c3_shapes_half" > gpurun_out/r04_b2_tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/r04_b2_tests.log
python tools/bench_conv_half.py > gpurun_out/r04_b2_bch.log 2>&1; tail -1 gpurun_out/r04_b2_bch.log
python bench.py --precision half --also "" --no-cpu-baseline --no-roofline --steps 20 --warmup 5 > gpurun_out/r04_b2_half.json 2> gpurun_out/r04_b2_half.err
python -c "import json;d=json.load(open('gpurun_out/r04_b2_half.json'));print('half ms/step', d['ms_per_step'])"
python -m pytest tests/test_half_models_gpu.py -q -x -k "fallback or every_call" > gpurun_out/r04_b2_tests2.log 2>&1; echo "tests2 rc=$?"; tail -2 gpurun_out/r04_b2_tests2.log
