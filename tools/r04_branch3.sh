#!/bin/bash
cd "$GRAFT_REPO_ROOT" && export TMPDIR=/tmp
run() { echo "$* : $(env "$@" timeout -k 10 300 python3 tools/bench_model.py $CFG 2>&1 | grep -a 'ms/step' | tr '\n' ' ' | cut -c1-130)"; }
CFG="hrnet 8 512 21 20"
for pol in half fp32; do
run PSEG_PRECISION=$pol PSEG_LANES_OVERFLOW=parent
run PSEG_PRECISION=$pol PSEG_LANES_OVERFLOW=light
run PSEG_PRECISION=$pol PSEG_LANES_OVERFLOW=parent
run PSEG_PRECISION=$pol PSEG_LANES_OVERFLOW=light
done
