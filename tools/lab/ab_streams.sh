#!/bin/bash
# which change of round 6 moved the replayed HRNet step (14.03 -> 14.55 ms fp32, 7.17 -> 7.7 ms -mp)?  PSEG_OWN_STREAMS=0: streams
# straight from torch's pool and torch's default capture stream, as up to round 5
O=gpurun_out
: > $O/r06_ab_streams.txt
for rep in 1 2; do
  for own in 1 0; do
    for cfg in hrnet:half; do
      PSEG_OWN_STREAMS=$own python bench.py --only-config $cfg --steps 40 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for k,v in d.items():
    if isinstance(v,dict): print('own_streams=$own rep $rep $cfg %.3f ms' % v['ms_per_step'])" >> $O/r06_ab_streams.txt
    done
  done
done
sort $O/r06_ab_streams.txt
