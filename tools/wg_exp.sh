cd $GRAFT_REPO_ROOT
for st in 3 4; do PSEG_HWGRAD_STAGES=$st timeout -k 10 200 python -m pytest tests/test_half_gpu.py -q -x -k conv2d 2>&1 | tail -1; done
for st in 2 3 4; do echo "wstages=$st"; PSEG_HWGRAD_STAGES=$st timeout -k 10 100 python tools/bench_conv_half.py 2>&1 | grep -v amdgpu; done
