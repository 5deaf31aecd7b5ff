#!/bin/bash
# GPU_MAX_HW_QUEUES sweep (round 6): the runtime multiplexes a process's streams onto this many hardware queues (default 4)
O=gpurun_out
: > $O/r06_ab_hwq.txt
for q in 4 2 3 5 6; do
  for cfg in hrnet:half hrnet:fp32 unet:half; do
    GPU_MAX_HW_QUEUES=$q python bench.py --only-config $cfg --steps 40 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for k,v in d.items():
    if isinstance(v,dict): print('hwq=$q $cfg %.3f ms' % v['ms_per_step'])" >> $O/r06_ab_hwq.txt
  done
  GPU_MAX_HW_QUEUES=$q python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --also "" --configs "" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('hwq=$q deeplab fp32 %.3f ms' % d['ms_per_step'])" >> $O/r06_ab_hwq.txt
done
cat $O/r06_ab_hwq.txt
