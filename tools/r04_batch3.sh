#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
PSEG_HCONV_ABLATE=31 bash tools/pmc_any.sh r04abl31 tools/bench_conv_half.py l1_1x1b l3_1x1b l4_1x1b l3_3x3
bash tools/pmc_any.sh r04abl0 tools/bench_conv_half.py l1_1x1b l3_1x1b l4_1x1b l3_3x3
