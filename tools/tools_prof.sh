#!/bin/bash
# usage: tools_prof.sh <tag> [bench args...]   (run on the GPU box via gpurun)
set -u
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py "$@" --no-cpu > $OUT/bench.json 2> $OUT/trace.err
echo "trace rc=$?"
