#!/bin/bash
# round 4, GPU batch 1: new tests, AUTO graph mode, wgrad blocks-per-CU, counters of the fp16 conv kernels
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
python -m pytest tests/test_half_models_gpu.py tests/test_lanes_gpu.py tests/test_cli_gpu.py -q -x -s -k "fallback or eval_forward or graph_replay or lane or cli or default_trainer or entry_points or validation" > gpurun_out/r04_b1_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r04_b1_tests.log
for cfg in "hrnet 8 512 21 20" "unet 8 256 2 30"; do for pol in fp32 half; do
  echo "auto: $(PSEG_PRECISION=$pol PSEG_GRAPH_VERBOSE=1 python3 tools/bench_model.py $cfg 2>&1 | grep -a 'ms/step\|lane executor\|auto graph\|graph mode')"
done; done > gpurun_out/r04_b1_auto.txt 2>&1
echo "auto: $(PSEG_PRECISION=half PSEG_GRAPH_VERBOSE=1 python3 tools/bench_model.py deeplabv3plus 16 512 21 10 2>&1 | grep -a 'ms/step\|lane executor\|auto graph\|graph mode')" >> gpurun_out/r04_b1_auto.txt 2>&1
cat gpurun_out/r04_b1_auto.txt
for bpc in 0 1; do
  PSEG_WGRAD_BPC=$bpc python bench.py --precision half --also "" --no-cpu-baseline --no-roofline --steps 20 --warmup 5 > gpurun_out/r04_b1_half_bpc$bpc.json 2> gpurun_out/r04_b1_half_bpc$bpc.err
  echo "bpc=$bpc: $(python -c "import json;d=json.load(open('gpurun_out/r04_b1_half_bpc$bpc.json'));print(d['ms_per_step'])")"
done
bash tools/pmc_any.sh r04conv tools/bench_conv_half.py l4_3x3d2 l3_3x3 l3_1x1b l4_1x1b l1_1x1b aspp_d6
tail -5 gpurun_out/pmcany_r04conv.txt
