#!/usr/bin/env python3
"""Digest of tools_profile_all.sh output: kernel stats table + PMC per-launch
averages for the named kernels, written under gpurun_out/<tag>/digest/."""
import collections, csv, glob, json, os, sys
root, tag = sys.argv[1], sys.argv[2]
out = os.path.join(root, "digest")
os.makedirs(out, exist_ok=True)
# kernel stats
stats = glob.glob(root + "/trace/**/*kernel_stats.csv", recursive=True)
if stats:
    rows = list(csv.DictReader(open(stats[0])))
    with open(os.path.join(out, f"{tag}_kernel_stats.csv"), "w") as f:
        w = csv.DictWriter(f, fieldnames=rows[0].keys())
        w.writeheader()
        w.writerows(rows)
bench = os.path.join(root, "bench.json")
if os.path.exists(bench):
    txt = open(bench).read().strip().splitlines()
    open(os.path.join(out, f"{tag}_bench.json"), "w").write(txt[-1] + "\n")
want = {"fm_search": "k_search<true, false, true", "fm_search_rank_only": "k_search<false, false, true",
        "fm_search_unfused": "k_search<true, false, false", "isa_partition": "k_isa_deal<1>", "isa_partition2": "k_isa_deal<2>", "isa_fill": "k_isa_fill",
        "pack_reads": "k_pack_reads", "leaf_sort": "k_leaf_sort", "part_scatter": "k_part_scatter",
        "part_hist": "k_part_hist", "inverse_sa": "k_inverse_sa", "kmer_level": "k_kmer_level",
        "locate_walk": "k_locate(", "locate_sa": "k_locate_sa"}
pmc = {}
for f in sorted(glob.glob(root + "/pmc_*/**/*_counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        short = next((s for s, pat in want.items() if pat in k), None)
        if not short:
            continue
        acc[short][r["Counter_Name"]] += float(r["Counter_Value"])
        n[short].add(r["Dispatch_Id"])
    for k in acc:
        pmc.setdefault(k, {"launches": len(n[k])})
        for c, v in acc[k].items():
            pmc[k][c] = v / len(n[k])
for k, d in pmc.items():
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        d["hbm_bytes_per_launch"] = (d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024
    if "TCC_HIT_sum" in d:
        d["l2_hit_rate"] = d["TCC_HIT_sum"] / max(d["TCC_HIT_sum"] + d["TCC_MISS_sum"], 1)
json.dump(pmc, open(os.path.join(out, f"{tag}_pmc.json"), "w"), indent=1, sort_keys=True)
for k, d in pmc.items():
    print(k, {c: (round(v, 4) if isinstance(v, float) else v) for c, v in d.items()})
