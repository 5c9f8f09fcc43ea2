#!/bin/bash
# GPU box: one rocprofv3 --pmc pass over a process that only builds indexes, summed per kernel.
# usage: [ARRAYS=arrays] tools_kernel_pmc.sh <tag> <n> <builds> <counter> [<counter> ...]   (at most 8 SQ_, 4 TCC_ ... per pass)
# ARRAYS=arrays: every build also makes the search-only arrays (position records, k-mer table)
set -u
TAG=$1; N=$2; B=$3; shift 3
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmc -- python3 $GRAFT_REPO_ROOT/tools/tools_build_pmc.py run $N $B ${ARRAYS:-} > $OUT/run.log 2>&1 || { tail -3 $OUT/run.log; exit 1; }
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for f in glob.glob("$OUT/pmc/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "bwtgpu::" not in k: continue
        short = k.split("bwtgpu::")[1].split("(")[0]
        acc[short][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[(short, r["Counter_Name"])] += 1
for k, c in sorted(acc.items()):
    print(k, {n: round(v / $B) for n, v in c.items()}, "launches/build", {n: calls[(k, n)] / $B for n in c})
PY
find $OUT -name "*.db" -delete; find $OUT -name "*counter_collection.csv" -size +8M -delete
