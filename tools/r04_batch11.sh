#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for cfg in "2 0" "3 0" "4 0" "2 32" "3 32" "4 32" "2 64" "3 64"; do set -- $cfg
  w=$(PSEG_HWGRAD_STAGES=$1 PSEG_HWGRAD_BKP=$2 python tools/bench_conv_half.py 2>/dev/null > gpurun_out/r04_b11_w_$1_$2.log; tail -1 gpurun_out/r04_b11_w_$1_$2.log)
  s=$(PSEG_HWGRAD_STAGES=$1 PSEG_HWGRAD_BKP=$2 python bench.py --precision half --also "" --no-cpu-baseline --no-roofline --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('%.3f' % d['ms_per_step'])")
  echo "stages=$1 bkp=$2: $w | step $s"
done
