#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for f in test_cli_gpu test_half_models_gpu test_bench_path_gpu test_dist_gpu test_lanes_gpu; do
  timeout -k 10 400 python -m pytest tests/$f.py "tests/test_models_gpu.py::test_trainer_graph_replay_matches_eager" -q -x -p no:cacheprovider > gpurun_out/r04_bis_$f.log 2>&1
  echo "$f + graph_replay: rc=$? $(grep -c Fatal gpurun_out/r04_bis_$f.log) fatal; $(tail -1 gpurun_out/r04_bis_$f.log | cut -c1-120)"
done
