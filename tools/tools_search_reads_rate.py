#!/usr/bin/env python3
"""Rate of the search_reads anchor-loop kernel and of mms on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bwt_amd import gpu as G, synth, _lib
G.init(0)
n, R, m = 46_000_000, 2_000_000, 100
t = G.gen_genome(n, synth.GENOME_SEED)
ix = G.build_index_dev(t, n)
reads, _ = G.gen_reads(t, n, R, m, 77, want_offsets=False)
r = reads[: R * m].view(R, m).clone()
# two substitutions per read in 30 % of the reads, an N in 5 %
g = torch.Generator(device="cuda"); g.manual_seed(5)
sel = torch.rand(R, generator=g, device="cuda") < 0.3
for _ in range(2):
    pos = torch.randint(0, m, (R,), generator=g, device="cuda")
    rows = torch.nonzero(sel).squeeze(1)
    r[rows, pos[rows]] = (r[rows, pos[rows]] + 1) & 3
seln = torch.rand(R, generator=g, device="cuda") < 0.05
rows = torch.nonzero(seln).squeeze(1)
r[rows, torch.randint(0, m, (R,), generator=g, device="cuda")[rows]] = 5
flat = r.reshape(-1).contiguous()
rev = torch.flip(r, dims=[1]).reshape(-1).contiguous()
offs = (torch.arange(R + 1, dtype=torch.int64, device="cuda") * m)
out = torch.empty(R * 4, dtype=torch.int32, device="cuda")
lib = _lib.load()
res = {}
for name in ("full", "compact"):
    if name == "compact":
        ix.compact()
    for rep in range(2):
        G.timing_enable(True); G.timing_reset()
        _lib.check(lib.bwtgpu_search_reads(ix.h, flat.data_ptr(), rev.data_ptr(), offs.data_ptr(), R, 20, out.data_ptr(), None))
        torch.cuda.synchronize()
        ms = G.timing_report()["fm_search_reads"][1]
    res[name] = out.clone()
    anchored = int((out.view(R, 4)[:, 0] > 0).sum())
    print(f"search_reads {name}: {R} x {m} bp in {ms:.2f} ms = {R/ms/1e3:.1f} M reads/s (forward-anchored {anchored})", flush=True)
    got = torch.empty(R, dtype=torch.int32, device="cuda"); sp = torch.empty_like(got); ep = torch.empty_like(got)
    G.timing_reset()
    _lib.check(lib.bwtgpu_mms(ix.h, flat.data_ptr(), m, m, R, got.data_ptr(), sp.data_ptr(), ep.data_ptr(), None))
    torch.cuda.synchronize()
    ms = G.timing_report()["fm_mms"][1]
    print(f"mms {name}: {ms:.2f} ms = {R/ms/1e3:.1f} M reads/s", flush=True)
assert torch.equal(res["full"], res["compact"])
print("same output on both index forms")
