#!/bin/bash
# the stem's weight gradient on the main stream planned as a CONCURRENT launch (PSEG_STEM_WGRAD_HINT=1, one resident block
# per CU, half the slabs) or as one that runs alone (0, round 5)
O=gpurun_out
: > $O/r06_ab_stem_hint.txt
for rep in 1 2 3; do
  for h in 1 0; do
    PSEG_STEM_WGRAD_HINT=$h python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --also "" --configs "" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('hint=$h rep $rep fp32 %.3f ms' % d['ms_per_step'])" >> $O/r06_ab_stem_hint.txt
  done
done
sort $O/r06_ab_stem_hint.txt
