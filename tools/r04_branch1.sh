#!/bin/bash
# branch lanes (ops.Branches): replay == eager, then HRNet step time with / without the forks.  usage: tools/r04_branch1.sh
cd "$GRAFT_REPO_ROOT" && export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_models_gpu.py -x -q -m gpu -k "replay_matches_eager" > $O/br1_tests.log 2>&1 || { echo tests failed; tail -30 $O/br1_tests.log; exit 1; }
tail -2 $O/br1_tests.log
for pol in half fp32; do
  for bs in 0 3; do
    echo "branch_streams=$bs $(PSEG_BRANCH_STREAMS=$bs PSEG_PRECISION=$pol PSEG_GRAPH=1 timeout -k 10 300 python3 tools/bench_model.py hrnet 8 512 21 20 2>&1 | grep -a 'ms/step\|lane executor')"
  done
done > $O/br1_bench.txt 2>&1
cat $O/br1_bench.txt
