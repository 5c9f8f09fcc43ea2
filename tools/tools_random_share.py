#!/usr/bin/env python3
"""What the 1 % of random reads in the search batch costs: the same batch with those reads
replaced by their neighbours (all reads occur in the text).  46 Mbp, 10 M x 50 bp."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bwt_amd import gpu as G, synth
G.init(0)
n, R, m = 46_000_000, 10_000_000, 50
t = G.gen_genome(n, synth.GENOME_SEED)
ix = G.build_index_dev(t, n)
ix.prepare_search()
reads, offs = G.gen_reads(t, n, R, m, synth.READS_SEED)
out = tuple(torch.empty(R, dtype=torch.int32, device="cuda") for _ in range(3))
def rate(d, label):
    for _ in range(3):
        ix.search(d, m, R, want_locrow=True, out=out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        ix.search(d, m, R, want_locrow=True, out=out)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(f"{label}: {dt*1e3:.3f} ms = {R/dt/1e9:.2f} G reads/s", flush=True)
rate(reads, "1 % random reads")
r2 = reads[: R * m].view(R, m).clone()
rnd = torch.nonzero(offs == -1).squeeze(1)
src = torch.clamp(rnd - 1, min=0)
r2[rnd] = r2[src]
still = int((offs[src] == -1).sum())
flat = torch.cat([r2.reshape(-1), torch.zeros(64, dtype=torch.uint8, device="cuda")])
rate(flat, f"every read from the text ({len(rnd)} replaced, {still} of the replacements random themselves)")
