#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_half_gpu.py tests/test_ops_gpu.py -q -x -k "batchnorm or bn_ or norm" > gpurun_out/r04_b6_tests.log 2>&1; echo "bn tests rc=$?"; tail -2 gpurun_out/r04_b6_tests.log
timeout -k 10 600 python -m pytest tests/test_models_gpu.py tests/test_half_models_gpu.py -q -x -k "every_call or golden or block" > gpurun_out/r04_b6_tests2.log 2>&1; echo "model tests rc=$?"; tail -2 gpurun_out/r04_b6_tests2.log
for r in 0 1; do for pol in half fp32; do
  PSEG_BN_FWD_ROWS=$r timeout -k 10 300 python bench.py --precision $pol --also "" --no-cpu-baseline --no-roofline --steps 20 --warmup 5 > gpurun_out/r04_b6_${pol}_r$r.json 2> gpurun_out/r04_b6_${pol}_r$r.err
  echo "rows=$r $pol: $(python -c "import json;d=json.load(open('gpurun_out/r04_b6_${pol}_r$r.json'));print(d['ms_per_step'])")"
done; done
