#!/usr/bin/env python3
"""Per-kernel times of ONE shard of an 8-rank build (what a rank does before the exchange)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bwt_amd import gpu as G, synth, dist as D
G.init(0)
lib = G._lib.load()
for n in [int(x) for x in (sys.argv[1:] or ["250000000", "3100000000"])]:
    t = G.gen_genome(n, synth.GENOME_SEED)
    counts = G.sa_bucket_counts(t, n)
    world = 8
    ranges = D.plan_bucket_ranges(counts, world); rows_of = D.shard_rows(counts, ranges)
    ws = torch.empty(int(lib.bwtgpu_sa_workspace_bytes(n)), dtype=torch.uint8, device="cuda")
    (lo, hi), (b, r) = ranges[3], rows_of[3]
    rows_buf = torch.empty(r, dtype=torch.int32, device="cuda"); bw_buf = torch.empty(r, dtype=torch.uint8, device="cuda")
    G.sa_build_range(t, n, lo, hi, rows_buf, bw_buf, ws)
    G.timing_enable(True); G.timing_reset()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    G.sa_build_range(t, n, lo, hi, rows_buf, bw_buf, ws)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
    rep = G.timing_report(); G.timing_enable(False)
    print(f"n={n} shard 3 of 8: {dt:.2f} ms wall (timing on);", {k: round(v[1], 3) for k, v in sorted(rep.items()) if v[1] > 0.005}, flush=True)
    del ws, rows_buf, bw_buf, t
    lib.bwtgpu_release_cache()
