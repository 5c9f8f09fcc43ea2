#!/usr/bin/env python3
"""Copies the digests of one profiling round from gpurun_out/ into profiles/ and refreshes
profiles/pmc_fm_search.json (what bench.py looks its `traffic` figures up in) from them.
usage: tools_publish_profiles.py <tag>     expects gpurun_out/<tag>/digest (tools_profile_r3.sh),
       gpurun_out/<tag>_c4 (tools_c4.sh) and gpurun_out/<tag>_build46 (tools_build_pmc.sh)"""
import csv, glob, json, os, re, shutil, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
go, pr = os.path.join(root, "gpurun_out"), os.path.join(root, "profiles")
for f in glob.glob(f"{go}/{tag}/digest/*"):
    shutil.copy(f, pr)
c4 = f"{go}/{tag}_c4"
shutil.copy(f"{c4}/c4_bench.json", f"{pr}/{tag}_c4_bench.json")
shutil.copy(f"{c4}/c4_kernel_stats.csv", f"{pr}/{tag}_c4_kernel_stats.csv")
log = open(f"{go}/{tag}_c4.log").read()
open(f"{pr}/{tag}_c4_pmc_summary.txt", "w").write(log[log.index("== pmc_fetch"):] if "== pmc_fetch" in log else log)
shutil.copy(f"{go}/{tag}_build46/{tag}_build46_traffic.json", f"{pr}/{tag}_build46_traffic.json")

rec = json.load(open(f"{pr}/pmc_fm_search.json"))
pmc = json.load(open(f"{pr}/{tag}_pmc.json"))
s = pmc["fm_search"]
rec["46000000bp_10000000x50"] = s["hbm_bytes_per_launch"]
rec["46000000bp_10000000x50_l2_misses"] = s["TCC_MISS_sum"]
rec["_l2_hit_rate"] = s["l2_hit_rate"]
rec["_source"] = (f"profiles/{tag}_pmc.json: (FETCH_SIZE + WRITE_SIZE) * 1024 per launch of k_search<true,false,true> "
                  f"({s['launches']} launches), separate rocprofv3 --pmc passes of `bench.py --steps 20 --warmup 3 --no-c5 "
                  f"--no-c4 --no-callers --no-cpu` (tools/tools_profile_r3.sh {tag}); FETCH_SIZE taken as reported (random 4-16 "
                  "byte loads, 64-byte requests; the factor 2 of wide streaming reads not applied)")
def per_launch(kernel, counter):
    m = re.search(r"== pmc_(?:fetch|write)\n(?:.*\n)*?\s+%s \(launches (\d+)\): \{([^}]*)\}" % re.escape(kernel), log)
    for blk in re.finditer(r"%s \(launches (\d+)\): \{([^}]*)\}" % re.escape(kernel), log):
        d = dict((k.strip("' "), float(v)) for k, v in (kv.split(":") for kv in blk.group(2).split(",")))
        if counter in d:
            return d[counter]
    return None
f, w, miss = per_launch("k_search", "FETCH_SIZE"), per_launch("k_search", "WRITE_SIZE"), per_launch("k_search", "TCC_MISS_sum")
if f and w:
    rec["3100000000bp_100000000x100"] = (f + w) * 1024
    rec["3100000000bp_100000000x100_l2_misses"] = miss
    rec["_source_c4"] = (f"profiles/{tag}_c4_pmc_summary.txt: rocprofv3 --pmc passes (FETCH_SIZE; WRITE_SIZE TCC_HIT_sum "
                         f"TCC_MISS_sum) of `bench.py --config c4 --steps 3 --warmup 1` on one MI355X (tools/tools_c4.sh {tag}_c4); "
                         "per launch of k_search = (FETCH_SIZE + WRITE_SIZE) * 1024 averaged over its launches (the one-call form "
                         "that also writes positions, one search with rows, one of the counting form); FETCH_SIZE as reported")
b = json.load(open(f"{pr}/{tag}_build46_traffic.json"))["per_build_bytes"]
rec["46000000bp_build"] = b["sum_with_fetch_doubled"]
rec["46000000bp_build_as_reported"] = b["sum"]
rec["_source_build"] = (f"profiles/{tag}_build46_traffic.json (tools/tools_build_pmc.sh: a process that only runs 5 make_fmi builds "
                        "of the 46 Mbp text, one rocprofv3 --pmc pass for FETCH_SIZE and one for WRITE_SIZE, every kernel summed, "
                        "divided by 5): WRITE_SIZE * 1024 + 2 * FETCH_SIZE * 1024 -- the build's loads are streaming and gfx950 "
                        "tallies their 128-byte requests at 64 bytes; ..._as_reported: without the doubling")
json.dump(rec, open(f"{pr}/pmc_fm_search.json", "w"), indent=1)
print({k: v for k, v in rec.items() if not k.startswith("_")})
