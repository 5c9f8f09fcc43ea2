#!/bin/bash
# GPU box: rocprofv3 kernel trace + stats of the plain `python3 bench.py` (the command the driver runs at N=1:
# the 46 Mbp legs and the 3.1 Gbp leg in one process); the stats table and the line it printed go to gpurun_out/<tag>/.
set -u
TAG=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py > $OUT/bench.json 2> $OUT/bench.err || { tail -3 $OUT/bench.err; exit 1; }
cd $GRAFT_REPO_ROOT
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_kernel_stats.csv
find $OUT -name "*.db" -delete; find $OUT -name "*_trace.csv" -delete
python3 - <<PY
import csv, json
d = json.loads([l for l in open("$OUT/bench.json") if l.startswith("{")][-1])
rows = {r["Name"].split("(")[0]: r for r in csv.DictReader(open("$OUT/${TAG}_kernel_stats.csv"))}
for name, ev in (("void bwtgpu::k_search<true, false, true, false>", d["roofline"]["kernel_ms"]),
                 ("void bwtgpu::k_search<true, false, true, true>", d["c4"]["roofline"]["kernel_ms"])):
    r = rows.get(name)
    print(name, "trace avg ms", float(r["AverageNs"]) / 1e6 if r else None, "calls", r["Calls"] if r else None, "| bench events ms", ev)
PY
