#!/bin/bash
# End-of-round measurement set (run on the GPU box via gpurun).  Two parts, each within one gpurun call's time limit:
#   tools/refresh_a.sh <tag>   bench line + kernel tables (two streams / one stream; fp32, half, mixed)
#   tools/refresh_b.sh <tag>   the launch-bound configurations, counters per kernel class, the forced 1-rank reducer
# then copy gpurun_out/<tag>_* into profiles/ (tracked) and regenerate profiles/<tag>_summary.md.
# usage: tools/refresh_profiles.sh <tag>     (both parts back to back, for boxes without a time limit)
cd "$GRAFT_REPO_ROOT" || exit 1
tools/refresh_a.sh "${1:-r03}" && tools/refresh_b.sh "${1:-r03}"
