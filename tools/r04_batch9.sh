#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -q -x -k "wgrad or conv2d" > gpurun_out/r04_b9_tests.log 2>&1; echo "ops tests rc=$?"; tail -2 gpurun_out/r04_b9_tests.log | cut -c1-160
for n in 0 1; do for cfg in "hrnet 8 512 21 20" "unet 8 256 2 30"; do
  echo "narrow64=$n $(PSEG_WGRAD_NARROW64=$n PSEG_PRECISION=fp32 PSEG_GRAPH=1 python3 tools/bench_model.py $cfg 2>&1 | grep -a 'ms/step')"
done; done
