#!/bin/bash
# usage (GPU box): tools_run.sh <tag>  -> runs gpu tests then bench, prints a digest
TAG=$1; shift
timeout -k 10 800 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/$TAG.test.log 2>&1
RC=$?; tail -4 gpurun_out/$TAG.test.log
[ $RC -ne 0 ] && { grep -E "Error|error|assert|FAILED" gpurun_out/$TAG.test.log | head -20; exit $RC; }
timeout -k 10 300 python -u bench.py "$@" > gpurun_out/$TAG.bench.json 2> gpurun_out/$TAG.bench.err || { tail -5 gpurun_out/$TAG.bench.err; exit 1; }
python - <<PY
import json; d=json.load(open("gpurun_out/$TAG.bench.json"))
print("search reads/s %.4g  ms/step %.3f  frac %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["frac"]))
print("build Mbp/s %.5g wall_ms %.3f stats %s" % (d["sa_build"]["value"], d["sa_build"]["wall_ms"], d["sa_build"]["stats"]))
print("build kernels", {k:v for k,v in d["sa_build"]["kernels_ms"].items() if v>0.02})
print("search arrays", d["sa_build"]["search_arrays"]["wall_ms"], {k:v for k,v in d["sa_build"]["search_arrays"]["kernels_ms"].items() if v>0.02})
print("packed", d.get("packed_input")); print("locate", d["locate"]); print("bytes/read", d["roofline"]["bytes_per_read"], d["roofline"]["loads"], d["roofline"]["counting_pass_matches"]); print("rank_only", d["rank_only"])
if "cpu_baseline" in d: print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["build_mbp_s"], d["cpu_baseline"]["gpu_matches_cpu"])
PY
