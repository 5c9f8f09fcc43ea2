#!/bin/bash
# GPU box: two PMC passes (FETCH_SIZE; WRITE_SIZE) over a process that only builds indexes.
# usage: tools_build_pmc.sh <tag> <n> <builds> [arrays]
set -u
TAG=$1; N=$2; B=$3; ARR=${4:-}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
timeout -s KILL 170 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $GRAFT_REPO_ROOT/tools/tools_build_pmc.py run $N $B $ARR > $OUT/fetch.log 2>&1 || { tail -3 $OUT/fetch.log; exit 1; }
timeout -s KILL 170 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $GRAFT_REPO_ROOT/tools/tools_build_pmc.py run $N $B $ARR > $OUT/write.log 2>&1 || { tail -3 $OUT/write.log; exit 1; }
cd $GRAFT_REPO_ROOT
python3 tools/tools_build_pmc.py sum $OUT $B > $OUT/${TAG}_traffic.json; cat $OUT/${TAG}_traffic.json
find $OUT -name "*.db" -delete; find $OUT -name "*counter_collection.csv" -size +8M -delete
