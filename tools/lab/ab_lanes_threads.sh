#!/bin/bash
# A/B of the threaded lane issue (csrc/lanes.hip, round 6): the launch-bound configurations, replayed, one thread vs one per lane
set -o pipefail
out=gpurun_out/r06_lanes_threads.txt
: > $out
for rep in 1 2; do
for cfg in hrnet:half hrnet:fp32 unet:half unet:fp32; do
  for th in 0 1; do
    echo "== $cfg PSEG_LANES_THREADS=$th rep $rep" >> $out
    PSEG_LANES_THREADS=$th python bench.py --only-config $cfg --steps 40 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for k,v in d.items():
    if isinstance(v,dict): print(k, round(v['ms_per_step'],3),'ms', round(v['value'],1),'img/s', v['step_mode'], v['lane_executor'])
" >> $out 2>&1
  done
done
done
cat $out
