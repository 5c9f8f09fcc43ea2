#!/bin/bash
# GPU box: kernels and copies of ONE index build in the order and at the times the device ran them
# (rocprofv3 kernel + memory-copy trace of a process that only builds; the last build is printed).
# usage: tools_build_timeline.sh <tag> <n>
set -u
TAG=$1; N=$2
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
timeout -s KILL 120 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/tools/tools_build_pmc.py run $N 6 > $OUT/run.log 2>&1 || { tail -3 $OUT/run.log; exit 1; }
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv, glob
ev = []
for f in glob.glob("$OUT/trace/*/*_kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("bwtgpu::", "").replace("void ", "")))
for f in glob.glob("$OUT/trace/*/*_memory_copy_trace.csv"):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", "")))
ev.sort()
starts = [i for i, e in enumerate(ev) if "k_prepare_text" in e[2]]
a = starts[-1]
t0 = ev[a][0]; prev_end = t0
for s, e, name in ev[a:]:
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:6.1f}  {name[:60]}")
    prev_end = max(prev_end, e)
print("total %.1f us" % ((prev_end - t0) / 1e3))
PY
find $OUT -name "*.db" -delete; find $OUT -name "*_trace.csv" -size +4M -delete
