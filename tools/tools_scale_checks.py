#!/usr/bin/env python3
"""Full-size self-checks on the GPU box: the tandem-repeat texts (prefix-doubling
path) and the 250 Mbp genome built in 8 shards, each suffix array certified on
the host (permutation + order, O(n))."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bwt_amd import gpu as G, synth, dist as D
from oracle import load_oracle, load_reference
orc = load_oracle()
ref = load_reference()  # None when oracle/_ref was not built
G.init(0)

def build(t, n, label):
    G.sa_build(t, n)  # warm-up: workspace
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sa = G.sa_build(t, n)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    st = G.sa_last_stats()
    print(f"{label}: n={n} SA in {dt*1e3:.1f} ms = {n/dt/1e6:.0f} Mbp/s; levels={st[0]} leaves={st[1]} "
          f"unresolved_rows={st[2]} doubling_rounds={st[3]}", flush=True)
    return sa

def certify(text_host, n, sa, label):
    t0 = time.perf_counter()
    bad = orc.check_sa(text_host, n, G.to_numpy_u32(sa))
    print(f"{label}: host certification {'OK' if bad == 0 else 'FAILED code %d' % bad} ({time.perf_counter()-t0:.1f} s)", flush=True)
    assert bad == 0

which = sys.argv[1:] or ["c5", "c3"]
if "c5" in which:
    n = 100_000_000
    for label, args in (("c5 period 7", (7, 11, 0, 0)), ("c5 period 1000", (1000, 11, 0, 0)),
                        ("c5 period 1000 + point mutations", (1000, 11, 5000, 3))):
        text = synth.gen_tandem(n, *args)
        t = torch.from_numpy(np.concatenate([text, np.zeros(64, np.uint8)])).cuda()
        sa = build(t, n, label)
        certify(text, n, sa, label)
        if ref is not None and args == (1000, 11, 5000, 3):
            t0 = time.perf_counter()
            sa_ref = ref.csuff_arr(text, n)
            dt = time.perf_counter() - t0
            same = bool((np.asarray(sa_ref, np.uint32) == G.to_numpy_u32(sa)).all())
            print(f"{label}: reference csuff_arr (SACA-K, 1 thread) {dt:.1f} s = {n/dt/1e6:.1f} Mbp/s; "
                  f"same suffix array: {same}", flush=True)
            assert same
        del sa, t
if "c3" in which:
    n = 250_000_000
    t = G.gen_genome(n, synth.GENOME_SEED)
    text = t[: n // 4 + 1].cpu().numpy()
    whole = build(t, n, "c3 whole")
    counts = G.sa_bucket_counts(t, n)
    ranges = D.plan_bucket_ranges(counts, 8)
    rows_of = D.shard_rows(counts, ranges)
    sa = torch.empty(n + 1, dtype=torch.int32, device="cuda"); sa[0] = n
    bw = torch.empty(n + 1, dtype=torch.uint8, device="cuda")
    ws = torch.empty(int(G._lib.load().bwtgpu_sa_workspace_bytes(n)), dtype=torch.uint8, device="cuda")
    ms = []
    for (lo, hi), (b, r) in zip(ranges, rows_of):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        G.sa_build_range(t, n, lo, hi, sa[b:b + r], bw[b:b + r], ws)
        torch.cuda.synchronize(); ms.append((time.perf_counter() - t0) * 1e3)
    print("c3 8 shards, ms each:", [round(x, 2) for x in ms], "rows:", [r for _, r in rows_of], flush=True)
    assert torch.equal(sa, whole), "shards do not concatenate to the whole-build suffix array"
    print("c3: 8 shards concatenate to the whole-build suffix array", flush=True)
    certify(text, n, sa, "c3")
print("OK")
