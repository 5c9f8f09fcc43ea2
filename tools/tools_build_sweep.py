#!/usr/bin/env python3
"""Build time (make_fmi on device, wall, timing off) over a range of text lengths: where the
children of the last bucket level fall relative to the leaf sorts' limits changes with n."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bwt_amd import gpu as G, synth
G.init(0)
for n in [int(float(x)) for x in (sys.argv[1:] or "1e6 4e6 16e6 30e6 46e6 52e6 60e6 70e6 100e6 180e6 250e6 400e6".split())]:
    t = G.gen_genome(n, synth.GENOME_SEED)
    best = 1e9
    for rep in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ix = G.build_index_dev(t, n)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
        if rep: best = min(best, dt)
        st = G.sa_last_stats()
        ix.close()
    print(f"n={n:>11d}  build {best:8.3f} ms  {n / best / 1e6:7.2f} Gbp/s  levels {st[0]} leaves {st[1]}", flush=True)
    del t
    G._lib.load().bwtgpu_release_cache()
