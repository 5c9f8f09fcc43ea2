#!/usr/bin/env python3
"""Large-genome run on the GPU box: build, search, locate, self-checks."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bwt_amd import gpu as G, synth
G.init(0)
n = int(float(sys.argv[1])); R = int(float(sys.argv[2])); m = int(sys.argv[3])
t0 = time.perf_counter()
t = G.gen_genome(n, synth.GENOME_SEED); torch.cuda.synchronize()
print(f"genome {n} bp generated in {time.perf_counter()-t0:.3f}s", flush=True)
G.timing_enable(True); G.timing_reset()
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ix = G.build_index_dev(t, n)
    torch.cuda.synchronize(); tb = time.perf_counter() - t0
    print(f"build rep{rep}: {tb*1e3:.1f} ms = {n/tb/1e6:.1f} Mbp/s stats={G.sa_last_stats()[:5]} C={ix.C} endloc={ix.endloc} index_GB={ix.device_bytes/1e9:.2f}", flush=True)
    if rep == 0: ix.close()
rep = G.timing_report()
print({k: round(v[1]/2, 2) for k, v in rep.items() if v[1] > 0.5}, flush=True)
assert ix.C[4] == n + 1
reads, offs = G.gen_reads(t, n, R, m, synth.READS_SEED)
torch.cuda.synchronize()
sp = ep = lr = None
for rep_i in range(2):
    G.timing_reset(); torch.cuda.synchronize(); t0 = time.perf_counter()
    sp, ep, lr = ix.search(reads, m, R, want_locrow=True)
    torch.cuda.synchronize(); ts = time.perf_counter() - t0
    t0 = time.perf_counter()
    pos, steps = ix.locate_rows(lr, count_steps=True)
    torch.cuda.synchronize(); tl = time.perf_counter() - t0
    print(f"search {R}x{m}: {ts*1e3:.2f} ms = {R/ts/1e6:.1f} Mreads/s ; locate {tl*1e3:.2f} ms = {R/tl/1e6:.1f} Mrows/s steps/row={steps/R:.2f}", flush=True)
spn, epn, offn, posn = (G.to_numpy_u32(x) for x in (sp, ep, offs, pos))
uniq = (epn - spn == 1) & (offn != 0xFFFFFFFF)
print("unique frac", uniq.mean(), "random frac", (offn == 0xFFFFFFFF).mean(), "nonempty among genome reads", ((epn > spn) | (offn == 0xFFFFFFFF)).mean())
bad = int((posn[uniq] != offn[uniq]).sum())
print("locate mismatches among unique reads:", bad)
assert bad == 0 and uniq.mean() > 0.9
print("OK")
