#!/bin/bash
# A/B of the round-6 epilogues (loads first, wait-free store loop; counted tile-boundary wait of the persistent pointwise kernel)
# against the round-5 library built as libpseg_amd_head.so
O=gpurun_out
HEAD=$PWD/pytorch_segmentation_amd/libpseg_amd_head.so
python tools/bench_conv.py fp32 > $O/r06_ab_conv_new.txt 2>&1
PSEG_LIB_PATH=$HEAD python tools/bench_conv.py fp32 > $O/r06_ab_conv_head.txt 2>&1
paste -d'\n' <(grep -a "GF" $O/r06_ab_conv_new.txt | sed 's/^/new  /') <(grep -a "GF" $O/r06_ab_conv_head.txt | sed 's/^/head /') > $O/r06_ab_conv.txt
grep -a "^total" $O/r06_ab_conv_new.txt | sed 's/^/new  /' >> $O/r06_ab_conv.txt
grep -a "^total" $O/r06_ab_conv_head.txt | sed 's/^/head /' >> $O/r06_ab_conv.txt
for rep in 1 2; do
  for lib in new head; do
    if [ $lib = head ]; then export PSEG_LIB_PATH=$HEAD; else unset PSEG_LIB_PATH; fi
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --also "" --configs "" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib rep $rep fp32 %.3f ms %.1f img/s' % (d['ms_per_step'], d['value']))" >> $O/r06_ab_step.txt
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --also "" --configs "" --precision half 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib rep $rep half %.3f ms %.1f img/s' % (d['ms_per_step'], d['value']))" >> $O/r06_ab_step.txt
  done
done
unset PSEG_LIB_PATH
cat $O/r06_ab_step.txt
