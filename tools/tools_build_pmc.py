#!/usr/bin/env python3
"""HBM-side traffic of ONE index build (GPU box).

  tools_build_pmc.py run <n> <builds> [arrays]   builds of an n-base text, nothing else; with
                                                 `arrays` each build also makes the search-only arrays
  tools_build_pmc.py sum <dir> <builds>          adds up FETCH_SIZE / WRITE_SIZE of every kernel in the
                                                 rocprofv3 --pmc CSVs under <dir>/pmc_*/ and divides by
                                                 the number of builds (warm-up build included in it)
"""
import csv, glob, json, os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if sys.argv[1] == "run":
    import torch
    from bwt_amd import gpu as G, synth
    n, builds = int(sys.argv[2]), int(sys.argv[3])
    arrays = len(sys.argv) > 4
    G.init(0)
    t = G.gen_genome(n, synth.GENOME_SEED)
    for _ in range(builds):
        ix = G.build_index_dev(t, n)
        if arrays:
            ix.prepare_search()
        ix.close()
    torch.cuda.synchronize()
    print("built", builds)
else:
    root, builds = sys.argv[2], int(sys.argv[3])
    out = {}
    for f in sorted(glob.glob(root + "/pmc_*/*/*_counter_collection.csv")):
        acc = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "bwtgpu::" not in k or "k_gen" in k:
                continue
            short = k.split("bwtgpu::")[1].split("(")[0].split("<")[0]
            acc[short][r["Counter_Name"]] += float(r["Counter_Value"])
        for k, c in acc.items():
            for name, v in c.items():
                if name in ("FETCH_SIZE", "WRITE_SIZE"):
                    out.setdefault(k, {})[name] = v * 1024 / builds
    tot_f = sum(v.get("FETCH_SIZE", 0) for v in out.values())
    tot_w = sum(v.get("WRITE_SIZE", 0) for v in out.values())
    rep = {"per_build_bytes": {"fetch_as_reported": tot_f, "write": tot_w, "sum": tot_f + tot_w,
                               "sum_with_fetch_doubled": 2 * tot_f + tot_w},
           "per_kernel_MB": {k: {a: round(b / 1e6, 1) for a, b in v.items()} for k, v in sorted(out.items())}}
    print(json.dumps(rep, indent=1))
