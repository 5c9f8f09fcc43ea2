#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer search (what fmi_search_batch costs a C caller)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from bwt_amd import gpu as G, synth, _lib
G.init(0)
n, R, m = 46_000_000, 10_000_000, 50
t = G.gen_genome(n, synth.GENOME_SEED)
ix = G.build_index_dev(t, n)
reads, _ = G.gen_reads(t, n, R, m, synth.READS_SEED, want_offsets=False)
h = reads[: R * m].cpu().numpy()
sp = np.empty(R, np.uint32); ep = np.empty(R, np.uint32); pos = np.empty(R, np.uint32)
lib = _lib.load()
vp = lambda a: a.ctypes.data_as(C.c_void_p)
for with_pos in (False, True):
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        _lib.check(lib.bwtgpu_search_host(ix.h, vp(h), m, m, R, vp(sp), vp(ep), None, vp(pos) if with_pos else None))
        best = min(best, time.perf_counter() - t0)
    print(f"host buffers, locate={with_pos}: {best*1e3:.1f} ms = {R/best/1e6:.1f} M reads/s "
          f"({(R*m + R*(12 if with_pos else 8))/best/1e9:.1f} GB/s over PCIe)")
