cd $GRAFT_REPO_ROOT
timeout -k 10 400 python bench.py --no-cpu --no-callers --no-c5 > gpurun_out/r3_wide_bench.json 2> gpurun_out/r3_wide_bench.err || { tail -5 gpurun_out/r3_wide_bench.err; exit 1; }
BWT_MI355X_WIDE_RECORDS=0 timeout -k 10 400 python bench.py --no-cpu --no-callers --no-c5 > gpurun_out/r3_narrow_bench.json 2> gpurun_out/r3_narrow_bench.err || { tail -5 gpurun_out/r3_narrow_bench.err; exit 1; }
python - <<P
import json
for tag in ("wide","narrow"):
    d=json.load(open(f"gpurun_out/r3_{tag}_bench.json")); c=d["c4"]
    print(tag, "c2 value %.4g kernel_ms %.4f | c4 value %.4g ms/step %.3f kernel %.3f arrays_ms %s index_bytes %.1f GB loads %s checks %s" % (d["value"], d["roofline"]["kernel_ms"], c["value"], c["ms_per_step"], c["roofline"]["kernel_ms"], c["sa_build"].get("search_arrays_ms"), c["index_bytes"]/1e9, {k: round(v/1e8,3) for k,v in c["roofline"]["loads"].items() if isinstance(v,(int,float)) and v>1e6}, c["checks"]))
P
