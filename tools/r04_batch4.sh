#!/bin/bash
# round 4, GPU batch 4: persistent gather kernel -- correctness, A/B per shape, step time
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_half_gpu.py -q -x -k "conv2d_fwd_half or dgrad_wgrad_half or c3_shapes_half" > gpurun_out/r04_b4_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 gpurun_out/r04_b4_tests.log
if [ $rc -ne 0 ]; then grep -n "Error\|assert\|FAILED" gpurun_out/r04_b4_tests.log | head -20; exit 1; fi
for pz in 0 1; do
  PSEG_HCONV_PERSIST=$pz timeout -k 10 300 python tools/bench_conv_half.py > gpurun_out/r04_b4_bch_p$pz.log 2>&1; echo "persist=$pz: $(tail -1 gpurun_out/r04_b4_bch_p$pz.log)"
done
for pz in 0 1; do
  PSEG_HCONV_PERSIST=$pz timeout -k 10 300 python bench.py --precision half --also "" --no-cpu-baseline --no-roofline --steps 20 --warmup 5 > gpurun_out/r04_b4_half_p$pz.json 2> gpurun_out/r04_b4_half_p$pz.err
  echo "persist=$pz: $(python -c "import json;d=json.load(open('gpurun_out/r04_b4_half_p$pz.json'));print('half ms/step', d['ms_per_step'], 'loss', d['config']['loss'])")"
done
