#!/usr/bin/env python3
"""Per-kernel split (HIP events) of the suffix-array build of the 100 Mbp tandem-repeat texts."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bwt_amd import gpu as G, synth
G.init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
for label, args in (("period 7", (7, 11, 0, 0)), ("period 1000", (1000, 11, 0, 0)),
                    ("period 1000 + point mutations", (1000, 11, 5000, 3))):
    text = synth.gen_tandem(n, *args)
    t = torch.from_numpy(np.concatenate([text, np.zeros(64, np.uint8)])).cuda()
    G.timing_enable(False)
    G.sa_build(t, n)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sa = G.sa_build(t, n)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    st = G.sa_last_stats()
    print(f"{label}: {dt*1e3:.1f} ms levels={st[0]} leaves={st[1]} unresolved={st[2]} rounds={st[3]}", flush=True)
    G.timing_enable(True); G.timing_reset()
    sa = G.sa_build(t, n)
    torch.cuda.synchronize()
    rep = G.timing_report()
    tot = sum(v[1] for v in rep.values())
    for k, v in sorted(rep.items(), key=lambda kv: -kv[1][1]):
        print(f"   {k:28s} {v[0]:5d} launches {v[1]:9.3f} ms", flush=True)
    print(f"   sum {tot:.2f} ms", flush=True)
    G.timing_enable(False)
    print("   certificate (out of range, not a permutation, out of order):", G.sa_certify(t, n, sa), flush=True)
    del sa, t
