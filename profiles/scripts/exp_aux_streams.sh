#!/bin/bash
# how many auxiliary (weight-gradient) streams: step time per configuration, replayed.  usage: tools/exp_aux_streams.sh
cd "$GRAFT_REPO_ROOT"
for n in 1 2 3; do
  for cfg in "hrnet 8 512 21 20 fp32" "hrnet 8 512 21 20 half" "unet 8 256 2 30 fp32" "unet 8 256 2 30 half" "deeplabv3plus 16 512 21 12 fp32" "deeplabv3plus 16 512 21 12 half"; do
    set -- $cfg
    echo "aux=$n $(PSEG_AUX_STREAMS=$n PSEG_PRECISION=$6 PSEG_GRAPH=1 python3 tools/bench_model.py $1 $2 $3 $4 $5 2>&1 | grep -a 'ms/step' | cut -c1-90)"
  done
done
