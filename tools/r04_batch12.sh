#!/bin/bash
# A/B on one box: ds_read_b64_tr_b16 as inline asm (default build) against the builtin (libpseg_amd_trbuiltin.so), interleaved
cd "$GRAFT_REPO_ROOT"
B=$GRAFT_REPO_ROOT/pytorch_segmentation_amd/libpseg_amd_trbuiltin.so
step() { python bench.py --precision half --also "" --no-cpu-baseline --no-roofline --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('%.3f' % d['ms_per_step'])"; }
for rep in 1 2 3; do
  echo "rep $rep: asm $(step)  builtin $(PSEG_LIB_PATH=$B step)"
done
python tools/bench_conv_half.py > gpurun_out/r04_b12_asm.log 2>&1; echo "asm: $(tail -1 gpurun_out/r04_b12_asm.log)"
PSEG_LIB_PATH=$B python tools/bench_conv_half.py > gpurun_out/r04_b12_builtin.log 2>&1; echo "builtin: $(tail -1 gpurun_out/r04_b12_builtin.log)"
echo "hrnet asm: $(PSEG_PRECISION=half python3 tools/bench_model.py hrnet 8 512 21 20 2>&1 | grep -a 'ms/step' | cut -c1-70)"
echo "hrnet builtin: $(PSEG_LIB_PATH=$B PSEG_PRECISION=half python3 tools/bench_model.py hrnet 8 512 21 20 2>&1 | grep -a 'ms/step' | cut -c1-70)"
