#!/usr/bin/env python3
"""The largest text the 32-bit row numbers of the reference's layout allow, nearly: build, device
certificate of suffix array and index, search + locate of reads cut from the text, and the same
index loaded from its stored parts.  usage: tools_max_size.py [n] [R] [m]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bwt_amd import gpu as G, synth
G.init(0)
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 4_200_000_000
R = int(float(sys.argv[2])) if len(sys.argv) > 2 else 20_000_000
m = int(sys.argv[3]) if len(sys.argv) > 3 else 100
t = G.gen_genome(n, synth.GENOME_SEED)
torch.cuda.synchronize(); t0 = time.perf_counter()
ix = G.build_index_dev(t, n)
torch.cuda.synchronize(); tb = time.perf_counter() - t0
print(f"n = {n}: build {tb*1e3:.1f} ms (first build, workspace allocation included), stats {G.sa_last_stats()[:4]}, C[4] == n + 1: {ix.C[4] == n + 1}", flush=True)
rep = ix.certify(t)
print("certificate:", rep, flush=True)
assert all(v == 0 for v in rep.values())
G._lib.load().bwtgpu_release_cache()
reads, offs = G.gen_reads(t, n, R, m, synth.READS_SEED)
sp, ep, pos = ix.search_locate(reads, m, R)
torch.cuda.synchronize(); t0 = time.perf_counter()
sp, ep, pos = ix.search_locate(reads, m, R)
torch.cuda.synchronize(); ts = time.perf_counter() - t0
spn, epn, offn, posn = (G.to_numpy_u32(x) for x in (sp, ep, offs, pos))
cut = offn != 0xFFFFFFFF
uniq = cut & (epn - spn == 1)
print(f"search + locate {R} x {m}: {ts*1e3:.2f} ms; index {ix.device_bytes/1e9:.1f} GB, table {ix.kmer_table}; cut reads found: {bool((epn[cut] > spn[cut]).all())}; "
      f"unique {uniq.mean():.4f}; located off their offset: {int((posn[uniq] != offn[uniq]).sum())}", flush=True)
assert (epn[cut] > spn[cut]).all() and (posn[uniq] == offn[uniq]).all()
a, b, lr = ix.search(reads, m, R, want_locrow=True)
p2 = ix.locate_rows(lr)
print("two calls equal the one:", bool(torch.equal(a, sp) and torch.equal(b, ep) and torch.equal(p2, pos)), flush=True)
ix.compact()
a, b, lr = ix.search(reads[: 2_000_000 * m], m, 2_000_000, want_locrow=True)
p2 = ix.locate_rows(lr)
print("rank blocks alone, 2 M reads, equal:", bool(torch.equal(a, sp[:2_000_000]) and torch.equal(b, ep[:2_000_000]) and torch.equal(p2, pos[:2_000_000])), flush=True)
print("OK")
