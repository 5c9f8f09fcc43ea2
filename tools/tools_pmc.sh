#!/bin/bash
# usage: tools_pmc.sh <tag> <pass-name> "<counters>" [bench args...]  (GPU box)
set -u
TAG=$1; PASS=$2; CTRS=$3; shift 3
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG/pmc_$PASS
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 180 rocprofv3 --pmc $CTRS --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py "$@" --no-cpu > $OUT/bench.json 2> $OUT/err.txt
echo "pmc $PASS rc=$?"
