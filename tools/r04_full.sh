#!/bin/bash
# the whole GPU suite + smoke, log under gpurun_out/.  usage: tools/r04_full.sh <tag>
cd "$GRAFT_REPO_ROOT" && export TMPDIR=/tmp
O=gpurun_out; T=${1:-x}
timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu > $O/full_$T.log 2>&1; rc=$?
tail -3 $O/full_$T.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke_$T.log 2>&1; rc=$?
tail -2 $O/smoke_$T.log; echo "smoke rc=$rc"; exit $rc
