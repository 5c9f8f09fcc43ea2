#!/usr/bin/env python3
"""Two ways to give each of G GPUs the suffixes of its bucket range before the sort, priced from
kernels timed on this one GPU (the exchange itself cannot be timed here and is priced at a range of
bandwidths):

  scan      (what bwt_amd.dist does) every rank reads the whole replicated text and keeps the
            suffixes of its own range: level 0 of bwtgpu_sa_build_range.  Nothing travels.
  exchange  (ids to bucket owners) every rank takes a slice of n/G positions, histograms and scatters
            them by owner -- level 0 of a build of an n/G-base text does exactly that work on n/G
            positions -- sends each owner its part (all-to-all), and the owner then either gathers the
            text for the ids it got (4 B per id on the wire, a random gather of n/G ids over the text)
            or regroups (id, word) pairs that came with their 15 bases (8 B per id on the wire, one more
            scatter pass over n/G pairs: priced as the level-1 scatter of the shard).

Everything after that (levels 1.., leaves, the exchange of index products, the assembly) is the same."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bwt_amd import gpu as G, synth, dist as D
G.init(0)
lib = G._lib.load()
LEVEL0 = ("sa_prepare_text", "sa_part_hist_l0", "sa_part_scatter_l0", "sa_restrict_level0")


def timed_kernels(fn):
    fn()
    G.timing_enable(True); G.timing_reset()
    torch.cuda.synchronize()
    fn()
    torch.cuda.synchronize()
    rep = G.timing_report(); G.timing_enable(False)
    return {k: v[1] for k, v in rep.items()}


out = {}
for n in [int(x) for x in (sys.argv[1:] or ["250000000", "3100000000"])]:
    res = {}
    for world in (2, 4, 8):
        t = G.gen_genome(n, synth.GENOME_SEED)
        counts = G.sa_bucket_counts(t, n)
        ranges = D.plan_bucket_ranges(counts, world); rows_of = D.shard_rows(counts, ranges)
        g = world // 2
        (lo, hi), (b, r) = ranges[g], rows_of[g]
        ws = torch.empty(int(lib.bwtgpu_sa_workspace_bytes(n)), dtype=torch.uint8, device="cuda")
        rows_buf = torch.empty(r, dtype=torch.int32, device="cuda"); bw_buf = torch.empty(r, dtype=torch.uint8, device="cuda")
        k = timed_kernels(lambda: G.sa_build_range(t, n, lo, hi, rows_buf, bw_buf, ws))
        scan_ms = sum(v for name, v in k.items() if name in LEVEL0)
        hist_all = k.get("sa_part_hist", 0.0)
        regroup_ms = k.get("sa_part_scatter_l1", 0.0)
        shard_ms = sum(k.values())
        del ws, rows_buf, bw_buf, t
        lib.bwtgpu_release_cache()
        # the slice-local work of the exchange variant: level 0 of a build over n/G positions
        ns = n // world
        ts = G.gen_genome(ns, synth.GENOME_SEED + 1)
        ks = timed_kernels(lambda: G.sa_build(ts, ns))
        slice_ms = sum(v for name, v in ks.items() if name in LEVEL0)
        slice_hist = ks.get("sa_part_hist", 0.0)
        del ts
        lib.bwtgpu_release_cache()
        # random gather of n/G ids over the packed text (n/4 bytes)
        lines_per_s = G.gather_rate(max(n / 4 / (1 << 30), 1 / 64), loads=ns)
        gather_ms = ns / lines_per_s * 1e3
        sent = (world - 1) / world * ns  # ids a rank sends (and receives)
        row = {"scan_level0_ms": scan_ms, "scan_hist_all_levels_ms": hist_all, "shard_kernels_ms": shard_ms,
               "slice_level0_ms": slice_ms, "slice_hist_all_levels_ms": slice_hist,
               "gather_ms": gather_ms, "gather_lines_per_s": lines_per_s, "regroup_ms": regroup_ms,
               "ids_on_the_wire_bytes": 4 * sent, "pairs_on_the_wire_bytes": 8 * sent}
        for bw in (100, 300, 450):
            row[f"exchange_ids_at_{bw}GBs"] = slice_ms + 4 * sent / (bw * 1e6) + gather_ms
            row[f"exchange_pairs_at_{bw}GBs"] = slice_ms + 8 * sent / (bw * 1e6) + regroup_ms
        # bandwidth (GB/s per rank, all links together) at which the cheaper exchange form equals the scan
        budget_ids = scan_ms - slice_ms - gather_ms
        budget_pairs = scan_ms - slice_ms - regroup_ms
        row["break_even_GBs_ids"] = 4 * sent / (budget_ids * 1e6) if budget_ids > 0 else None
        row["break_even_GBs_pairs"] = 8 * sent / (budget_pairs * 1e6) if budget_pairs > 0 else None
        res[f"G{world}"] = row
        print(n, world, json.dumps({k2: (round(v, 3) if isinstance(v, float) else v) for k2, v in row.items()}), flush=True)
    out[str(n)] = res
print(json.dumps(out))
