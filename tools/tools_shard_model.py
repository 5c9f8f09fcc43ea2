#!/usr/bin/env python3
"""Inputs of the multi-GPU build cost model (GPU box, one GPU): for each text
size and rank count G, the time of the slowest logical shard (what one rank
sorts), of packing its symbols, of assembling the index from every shard's
products, of attaching the gathered rows, and the bytes a rank receives in the
two exchanges.  The one-GPU whole build is timed beside them."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bwt_amd import gpu as G, synth, dist as D

G.init(0)
lib = G._lib.load()
out = {}


def timed(fn, reps=2):
    best = None
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
        best = dt if best is None or dt < best else best
    return best, r


for n in [int(x) for x in (sys.argv[1:] or ["250000000", "3100000000"])]:
    t = G.gen_genome(n, synth.GENOME_SEED)
    rec = {}
    ix = G.build_index_dev(t, n); ix.close()
    ms, ix = timed(lambda: G.build_index_dev(t, n), 1)
    rec["one_gpu_index_ms"] = ms
    ms2, _ = timed(lambda: ix.prepare_search(), 1)
    rec["one_gpu_search_arrays_ms"] = ms2
    ix.close()
    counts = G.sa_bucket_counts(t, n)
    ws = torch.empty(int(lib.bwtgpu_sa_workspace_bytes(n)), dtype=torch.uint8, device="cuda")
    for world in (2, 4, 8):
        ranges = D.plan_bucket_ranges(counts, world)
        rows_of = D.shard_rows(counts, ranges)
        longest = max(r for _, r in rows_of)
        rows_buf = torch.empty(longest, dtype=torch.int32, device="cuda")
        bw_buf = torch.empty(longest, dtype=torch.uint8, device="cuda")
        shard_ms, pack_ms = [], []
        packed, offsets, at = [], [], 0
        samples = torch.empty(n // 8 + 1, dtype=torch.int32, device="cuda")
        sa = torch.empty(n + 1, dtype=torch.int32, device="cuda")
        endloc = 0
        for (lo, hi), (b, r) in zip(ranges, rows_of):
            G.sa_build_range(t, n, lo, hi, rows_buf, bw_buf, ws)
            ms, res = timed(lambda: G.sa_build_range(t, n, lo, hi, rows_buf, bw_buf, ws), 1)
            shard_ms.append(ms); endloc += res[2]
            ms, p = timed(lambda: G.pack_bw2(bw_buf, r), 1)
            pack_ms.append(ms); packed.append(p); offsets.append(at); at += (r + 3) // 4
            first, i0, cnt = D.sample_span(b, r)
            samples[i0:i0 + cnt] = rows_buf[first:r:8]
            sa[b:b + r] = rows_buf[:r]
        samples[0] = n - (1 << 32) if n >= (1 << 31) else n
        sa[0] = n - (1 << 32) if n >= (1 << 31) else n
        parts = D.ShardedParts(endloc, rows_of, torch.cat(packed + [torch.zeros(16, dtype=torch.uint8, device="cuda")]), offsets, samples)
        del packed
        G.build_index_from_shards(t, n, parts).close()
        ms_asm, ixs = timed(lambda: G.build_index_from_shards(t, n, parts), 1)
        if world == 8:
            G.timing_enable(True); G.timing_reset()
            G.build_index_from_shards(t, n, parts).close()
            rep = G.timing_report(); G.timing_enable(False)
            print(n, "assemble kernels", {k: round(v[1], 3) for k, v in sorted(rep.items()) if v[1] > 0.005}, flush=True)
        ms_att, _ = timed(lambda: G.check(lib.bwtgpu_index_attach_sa(ixs.h, t.data_ptr(), sa.data_ptr(), None)), 1)
        ok = all(v == 0 for v in ixs.certify(t).values())
        ixs.close()
        rec[f"G{world}"] = {"max_shard_ms": max(shard_ms), "mean_shard_ms": sum(shard_ms) / world, "pack_ms": max(pack_ms),
                            "assemble_ms": ms_asm, "attach_rows_ms": ms_att, "certified": ok,
                            "recv_products_bytes": (n // 4 + (n // 8) * 4) * (world - 1) // world,
                            "recv_rows_bytes": 4 * n * (world - 1) // world,
                            "max_over_mean_rows": longest / (n / world)}
        del rows_buf, bw_buf, samples, sa, parts
        print(n, world, json.dumps(rec[f"G{world}"]), flush=True)
    out[str(n)] = rec
    del ws, t
    lib.bwtgpu_release_cache()
print(json.dumps(out))
