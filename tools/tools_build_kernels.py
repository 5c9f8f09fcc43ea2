#!/usr/bin/env python3
"""Per-kernel times of one whole index build (make_fmi on device) and of the search-only arrays.
usage: tools_build_kernels.py [n ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bwt_amd import gpu as G, synth
G.init(0)
for n in [int(x) for x in (sys.argv[1:] or ["46000000", "3100000000"])]:
    t = G.gen_genome(n, synth.GENOME_SEED)
    ix = G.build_index_dev(t, n); ix.close()
    G.timing_enable(False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ix = G.build_index_dev(t, n)
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) * 1e3
    ix.close()
    G.timing_enable(True); G.timing_reset()
    ix = G.build_index_dev(t, n)
    torch.cuda.synchronize()
    rep = G.timing_report(); G.timing_reset()
    ix.prepare_search(); torch.cuda.synchronize()
    rep2 = G.timing_report(); G.timing_enable(False)
    print(f"n={n}: build {wall:.3f} ms wall (timing off); kernels (ms):",
          {k: round(v[1], 3) for k, v in sorted(rep.items(), key=lambda kv: -kv[1][1])}, flush=True)
    print(f"   search arrays:", {k: round(v[1], 3) for k, v in sorted(rep2.items(), key=lambda kv: -kv[1][1])}, flush=True)
    print("   stats", G.sa_last_stats()[:5], flush=True)
    ix.close(); del t
    G._lib.load().bwtgpu_release_cache()
