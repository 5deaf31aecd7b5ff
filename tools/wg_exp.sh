cd $GRAFT_REPO_ROOT
for cfg in "32 2" "32 3" "32 4" "64 2"; do set -- $cfg; PSEG_HWGRAD_BKP=$1 PSEG_HWGRAD_STAGES=$2 timeout -k 10 200 python -m pytest tests/test_half_gpu.py -q -x -k conv2d_dgrad_wgrad 2>&1 | tail -1; done
for cfg in "64 2" "32 2" "32 3" "32 4"; do set -- $cfg; echo "wcfg=$1/$2"; PSEG_HWGRAD_BKP=$1 PSEG_HWGRAD_STAGES=$2 timeout -k 10 100 python tools/bench_conv_half.py 2>&1 | grep -v amdgpu; done
