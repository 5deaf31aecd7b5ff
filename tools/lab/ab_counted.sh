#!/bin/bash
# A/B of the counted post-epilogue waits (round 6): fp16 persistent kernel (PSEG_HCONV_PERSIST_COUNTED) and fp32 persistent
# pointwise kernel (PSEG_CONV_PW_COUNTED), op level and step level
O=gpurun_out
for c in 1 0; do
  PSEG_HCONV_PERSIST_COUNTED=$c python tools/bench_conv_half.py > $O/r06_ab_half_conv_$c.txt 2>&1
done
paste -d'\n' <(grep -a "fwd" $O/r06_ab_half_conv_1.txt | sed 's/^/cnt1 /') <(grep -a "fwd" $O/r06_ab_half_conv_0.txt | sed 's/^/cnt0 /') > $O/r06_ab_half_conv.txt
: > $O/r06_ab_counted_steps.txt
for rep in 1 2; do
  for c in 1 0; do
    export PSEG_HCONV_PERSIST_COUNTED=$c PSEG_CONV_PW_COUNTED=$c
    for cfg in hrnet:half unet:half hrnet:fp32 unet:fp32; do
      python bench.py --only-config $cfg --steps 40 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for k,v in d.items():
    if isinstance(v,dict): print('counted=$c rep $rep $cfg %.3f ms' % v['ms_per_step'])" >> $O/r06_ab_counted_steps.txt
    done
    for pol in half fp32; do
      python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --also "" --configs "" --precision $pol 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('counted=$c rep $rep deeplab $pol %.3f ms' % d['ms_per_step'])" >> $O/r06_ab_counted_steps.txt
    done
  done
done
unset PSEG_HCONV_PERSIST_COUNTED PSEG_CONV_PW_COUNTED
sort $O/r06_ab_counted_steps.txt
