#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs: per kernel name, counter sums / launches."""
import csv, glob, sys, collections
root = sys.argv[1]
want = sys.argv[2:] or ["k_search", "k_locate", "k_leaf_sort", "k_part_scatter", "k_part_hist", "k_emit_planes", "k_isa_init"]
for f in sorted(glob.glob(root + "/pmc_*/*/*_counter_collection.csv")):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        short = next((w for w in want if w in k), None)
        if not short:
            continue
        acc[short][r["Counter_Name"]] += float(r["Counter_Value"])
        n[short].add(r["Dispatch_Id"])
    print("==", f.split("/")[-3])
    for k in acc:
        print(f"  {k} (launches {len(n[k])}):", {c: round(v / len(n[k])) for c, v in acc[k].items()})
