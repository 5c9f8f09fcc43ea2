#!/usr/bin/env python3
"""An index loaded from its stored parts: what the recovery of suffix array and text costs on the
first search, and the search rate before (lean) and after.  usage: tools_loaded_rate.py [n] [R] [m]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bwt_amd import gpu as G, synth
G.init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 46_000_000
R = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
m = int(sys.argv[3]) if len(sys.argv) > 3 else 50
d_text = G.gen_genome(n, synth.GENOME_SEED)
built = G.build_index_dev(d_text, n)
bwt, idxs = built.export()
endloc = built.endloc
d_reads, _ = G.gen_reads(d_text, n, R, m, synth.READS_SEED, want_offsets=False)
outs = {}
def rate(ix, label):
    sp, ep, lr = (torch.empty(R, dtype=torch.int32, device="cuda") for _ in range(3))
    ix.search(d_reads, m, R, want_locrow=True, out=(sp, ep, lr))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        ix.search(d_reads, m, R, want_locrow=True, out=(sp, ep, lr))
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pos = ix.locate_rows(lr)
    torch.cuda.synchronize(); dl = time.perf_counter() - t0
    outs[label] = (sp.clone(), ep.clone(), pos.clone())
    print(f"{label}: search {dt*1e3:.2f} ms = {R/dt/1e9:.2f} G reads/s; locate {dl*1e3:.2f} ms; "
          f"index {ix.device_bytes/1e6:.0f} MB, k-mer table {ix.kmer_table}", flush=True)
built.prepare_search()
rate(built, "built from the text")
built.close()
G._lib.load().bwtgpu_release_cache()
for lean in (True, False):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ix = G.index_from_parts(bwt, n, endloc, idxs, lean=lean)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    G.timing_enable(True); G.timing_reset()
    ix.prepare_search()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    rep = G.timing_report(); G.timing_enable(False)
    label = "loaded, kept lean" if lean else "loaded, arrays recovered"
    print(f"{label}: load {1e3*(t1-t0):.1f} ms (host parts -> rank blocks, samples), first-search preparation "
          f"{1e3*(t2-t1):.1f} ms: " + ", ".join(f"{k} {v[1]:.2f}" for k, v in sorted(rep.items())), flush=True)
    rate(ix, label)
    ix.close()
a, b, c = outs["built from the text"], outs["loaded, kept lean"], outs["loaded, arrays recovered"]
print("same answers:", all(torch.equal(x, y) and torch.equal(x, z) for x, y, z in zip(a, b, c)))
