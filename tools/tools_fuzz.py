#!/usr/bin/env python3
"""Randomised differential run on the GPU box: many small texts of different
kinds, every product of the build and every search form against the CPU oracle.
usage: tools_fuzz.py <seconds> [first_seed] [max_n]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from bwt_amd import gpu as G, synth, dist as D, _lib
from oracle import load_oracle
orc = load_oracle()

def make_text(rng, n):
    kind = rng.integers(0, 6)
    if kind == 0:
        bases = rng.integers(0, 4, n)
    elif kind == 1:   # tandem repeat, maybe mutated
        p = int(rng.integers(1, max(2, min(n, 2000))))
        bases = np.tile(rng.integers(0, 4, p), n // p + 1)[:n].copy()
        if rng.integers(0, 2):
            hits = rng.random(n) < 0.002
            bases[hits] = (bases[hits] + 1) & 3
    elif kind == 2:   # skewed composition
        bases = rng.choice(4, n, p=[0.85, 0.05, 0.05, 0.05])
    elif kind == 3:   # one base only, or two alternating
        bases = np.full(n, rng.integers(0, 4)) if rng.integers(0, 2) else (np.arange(n) & 1) * int(rng.integers(1, 4))
    elif kind == 4:   # random with a long duplicated segment
        bases = rng.integers(0, 4, n)
        if n > 100:
            L = int(rng.integers(10, n // 2)); a = int(rng.integers(0, n - L)); b = int(rng.integers(0, n - L))
            bases[b:b + L] = bases[a:a + L].copy()
    else:             # runs of As with islands
        bases = np.zeros(n, np.int64)
        for _ in range(int(rng.integers(1, 20))):
            a = int(rng.integers(0, n)); L = int(rng.integers(1, 200))
            bases[a:a + L] = rng.integers(0, 4, len(bases[a:a + L]))
    return kind, synth.pack(bases.astype(np.uint8)), bases.astype(np.uint8)

def reads_for(rng, bases, n, R, m, with_n):
    reads = np.empty((R, m), np.uint8)
    for r in range(R):
        k = rng.integers(0, 5)
        if n >= m and k != 4:
            j = int(rng.integers(0, n - m + 1)); reads[r] = bases[j:j + m]
            if k == 1:
                q = int(rng.integers(0, m)); reads[r, q] = (reads[r, q] + 1 + rng.integers(0, 3)) & 3
            elif k == 2:
                kk = int(rng.integers(1, min(m, n) + 1)); reads[r, :m - kk] = rng.integers(0, 4, m - kk); reads[r, m - kk:] = bases[:kk]
        else:
            reads[r] = rng.integers(0, 4, m)
        if with_n and m >= 2 and rng.integers(0, 4) == 0:  # a pattern of Ns only is undefined in the reference
            reads[r, rng.integers(0, m)] = 5
    return reads

def mms_gpu(ix, pats):
    R, m = pats.shape
    d = torch.from_numpy(np.ascontiguousarray(pats).reshape(-1)).cuda()
    got = torch.empty(R, dtype=torch.int32, device="cuda"); sp = torch.empty_like(got); ep = torch.empty_like(got)
    _lib.check(_lib.load().bwtgpu_mms(ix.h, d.data_ptr(), m, m, R, got.data_ptr(), sp.data_ptr(), ep.data_ptr(), None))
    return list(zip(got.cpu().numpy().tolist(), G.to_numpy_u32(sp).tolist(), G.to_numpy_u32(ep).tolist()))

def main(budget, seed0=1, nmax=300000, max_cases=None, quiet=False):
    G.init(0)
    t_end = time.time() + budget
    it = 0
    seed = seed0
    while time.time() < t_end and (max_cases is None or it < max_cases):
        rng = np.random.default_rng(seed)
        n = int(np.exp(rng.uniform(0, np.log(nmax)))) + 1
        kind, text, bases = make_text(rng, n)
        label = f"seed={seed} n={n} kind={kind}"
        try:
            sa_o = orc.sa_doubling(text, n)
            fo = orc.make_fmi(text, n, sa_o)
            k_env = int(rng.integers(0, 12))
            if rng.integers(0, 2):
                os.environ["BWT_MI355X_KMER"] = str(k_env)
            else:
                os.environ.pop("BWT_MI355X_KMER", None)
            if rng.integers(0, 3) == 0:  # fewer size bits in table entries (0: the form of texts of 2^30 bases and more)
                os.environ["BWT_MI355X_KMER_SIZEBITS"] = "0234"[int(rng.integers(0, 4))]
            else:
                os.environ.pop("BWT_MI355X_KMER_SIZEBITS", None)
            if rng.integers(0, 3) == 0:
                os.environ["BWT_MI355X_ISA_TWO_STEP"] = "12"[int(rng.integers(0, 2))]
            else:
                os.environ.pop("BWT_MI355X_ISA_TWO_STEP", None)
            os.environ["BWT_MI355X_WIDE_RECORDS"] = "01"[int(rng.integers(0, 2))]  # position records of either kind
            t = torch.from_numpy(np.concatenate([text, np.zeros(64, np.uint8)])).cuda()
            sa = G.sa_build(t, n)
            assert np.array_equal(G.to_numpy_u32(sa), sa_o), "SA"
            full = G.build_index_dev(t, n)
            lean = G.index_from_parts(fo.bwt, n, fo.endloc, fo.idxs, lean=True)
            loaded = G.index_from_parts(fo.bwt, n, fo.endloc, fo.idxs)  # recomputes suffix array and text on first search
            if rng.integers(0, 4):  # mostly: the search arrays made now, under the forced sizes; else by the first search
                full.prepare_search()
                lean.prepare_search()
                loaded.prepare_search()
            assert full.endloc == fo.endloc and full.C == fo.C, "endloc/C"
            bwt, idxs = full.export()
            assert np.array_equal(bwt, fo.bwt) and np.array_equal(idxs, fo.idxs), "export"
            assert np.array_equal(full.rank_rows16(), fo.rank_rows), "rank rows"
            rows = torch.arange(0, n + 1, dtype=torch.int64, device="cuda").to(torch.int32)
            for ix in (full, lean, loaded):
                assert np.array_equal(G.to_numpy_u32(ix.locate_rows(rows)), sa_o), "locate"
            # sharded
            if n > 4096:
                world = int(rng.integers(2, 9))
                counts = G.sa_bucket_counts(t, n)
                ranges = D.plan_bucket_ranges(counts, world); rows_of = D.shard_rows(counts, ranges)
                sh = torch.full((n + 1,), -1, dtype=torch.int32, device="cuda"); sh[0] = n
                ok = True
                for (lo, hi), (b, r) in zip(ranges, rows_of):
                    if r:
                        try:
                            G.sa_build_range(t, n, lo, hi, sh[b:b + r])
                        except G.BwtGpuError as e:
                            assert e.code == -6, str(e); ok = False; break
                if ok:
                    assert torch.equal(sh, sa), "shards"
            for m in sorted(set(int(x) for x in rng.integers(1, 120, 3))):
                R = 400
                reads = reads_for(rng, bases, n, R, m, False)
                sp_o, ep_o = orc.search_batch(fo, reads, m, m, R)
                want_loc = [orc.locate(fo, reads[r]) for r in range(R)]
                d = torch.from_numpy(reads.reshape(-1)).cuda()
                for name, ix in (("full", full), ("lean", lean), ("loaded", loaded)):
                    sp, ep, lr = (G.to_numpy_u32(x) for x in ix.search(d, m, R, want_locrow=True))
                    assert np.array_equal(sp, sp_o) and np.array_equal(ep, ep_o), f"search {name} m={m}"
                    pos = G.to_numpy_u32(ix.locate_rows(torch.from_numpy(lr.view(np.int32)).cuda())).astype(np.int64)
                    pos[pos == 0xFFFFFFFF] = -1
                    assert pos.tolist() == want_loc, f"locate() {name} m={m}"
                    a, b, c = (G.to_numpy_u32(x) for x in ix.search_locate(d, m, R))  # the one-call form
                    c = c.astype(np.int64); c[c == 0xFFFFFFFF] = -1
                    assert np.array_equal(a, sp_o) and np.array_equal(b, ep_o) and c.tolist() == want_loc, \
                        f"search_locate {name} m={m}"
                readsn = reads_for(rng, bases, n, 200, m, True)
                want = [orc.mms(fo, readsn[r]) for r in range(200)]
                for name, ix in (("full", full), ("lean", lean), ("loaded", loaded)):
                    assert mms_gpu(ix, readsn) == want, f"mms {name} m={m}"
            # search_reads anchor loop on reads of mixed lengths, with substitutions and N
            enc = []
            for _ in range(60):
                L = int(rng.integers(1, 160))
                if n >= L and rng.integers(0, 6):
                    j = int(rng.integers(0, n - L + 1)); e = bases[j:j + L].copy()
                    for _ in range(int(rng.integers(0, 3))):
                        q = int(rng.integers(0, L)); e[q] = (e[q] + 1) & 3
                else:
                    e = rng.integers(0, 4, L).astype(np.uint8)
                if L >= 2 and rng.integers(0, 5) == 0:
                    e[rng.integers(0, L)] = 5
                if e[-1] == 5 and (e == 5).all():
                    e[0] = 0
                enc.append(e.astype(np.uint8))
            anchor = int(rng.integers(1, 30))
            want = np.array([orc.search_reads_one(fo, e, anchor) for e in enc], np.uint32)
            offs = np.zeros(len(enc) + 1, np.uint64); offs[1:] = np.cumsum([len(e) for e in enc])
            d_f = torch.from_numpy(np.concatenate(enc)).cuda()
            d_r = torch.from_numpy(np.concatenate([e[::-1] for e in enc])).cuda()
            d_o = torch.from_numpy(offs.view(np.int64)).cuda()
            for name, ix in (("full", full), ("lean", lean), ("loaded", loaded)):
                out = torch.empty(len(enc) * 4, dtype=torch.int32, device="cuda")
                _lib.check(_lib.load().bwtgpu_search_reads(ix.h, d_f.data_ptr(), d_r.data_ptr(), d_o.data_ptr(),
                                                           len(enc), anchor, out.data_ptr(), None))
                assert np.array_equal(G.to_numpy_u32(out).reshape(-1, 4), want), f"search_reads {name} anchor={anchor}"
            # align_read_anchored (single_align.c:163-298), reads of 25..150 bases with edits and N; on
            # repetitive texts it walks every row of large intervals (as the reference does), so those stay small
            if n > 400 and (kind in (0, 4) or n <= 4000):
                al = []
                for _ in range(80):
                    L = int(rng.integers(25, min(150, n // 2)))
                    j = int(rng.integers(0, n - L)); e = list(bases[j:j + L])
                    for _ in range(int(rng.integers(0, 5))):
                        q = int(rng.integers(12, len(e))); w = int(rng.integers(0, 4))
                        if w == 0: e[q] = (e[q] + 1 + int(rng.integers(0, 3))) & 3
                        elif w == 1: del e[q:q + int(rng.integers(1, 4))]
                        elif w == 2: e[q:q] = list(rng.integers(0, 4, int(rng.integers(1, 4))))
                        else: e[q] = 5
                    if rng.integers(0, 8) == 0: e = list(rng.integers(0, 4, L))
                    al.append(np.array(e, np.uint8))
                a_len = int(rng.choice([8, 12, 16, 20]))
                want = [orc.align_read_anchored(fo, text, e, a_len) for e in al]
                offs = np.zeros(len(al) + 1, np.int64); offs[1:] = np.cumsum([len(e) for e in al])
                d_al = torch.from_numpy(np.concatenate(al + [np.zeros(1, np.uint8)])).cuda()
                d_ao = torch.from_numpy(offs).cuda()
                for name, ix in (("full", full), ("lean", lean), ("loaded", loaded)):
                    pos, ops, nops = G.align_reads(ix, t, d_al, d_ao, len(al), a_len, 160)
                    pos_h, ops_h, nops_h = pos.cpu().numpy(), ops.cpu().numpy(), G.to_numpy_u32(nops)
                    for r in range(len(al)):
                        o = 5 * int(offs[r]) + 64 * r
                        got = (int(pos_h[r]), ops_h[o:o + int(nops_h[r])].tobytes())
                        assert got == want[r], (f"align {name} anchor={a_len} read {r} len {len(al[r])}: {got[0]} vs {want[r][0]}; "
                                                f"ops ({len(got[1])}) {got[1][:120].hex()} vs ({len(want[r][1])}) {want[r][1][:120].hex()}; "
                                                f"read {bytes(al[r]).hex()}")
            full.close(); lean.close(); loaded.close()
        except AssertionError as e:
            raise AssertionError(f"MISMATCH {label} kmer_env={os.environ.get('BWT_MI355X_KMER')} "
                                 f"sizebits_env={os.environ.get('BWT_MI355X_KMER_SIZEBITS')} "
                                 f"wide_records={os.environ.get('BWT_MI355X_WIDE_RECORDS')}: {e}") from e
        finally:
            os.environ.pop("BWT_MI355X_KMER", None)
            os.environ.pop("BWT_MI355X_KMER_SIZEBITS", None)
            os.environ.pop("BWT_MI355X_WIDE_RECORDS", None)
            os.environ.pop("BWT_MI355X_ISA_TWO_STEP", None)
        it += 1
        seed += 1
        if it % 5 == 0 and not quiet:
            print(f"{it} cases ok (last {label})", flush=True)
    print(f"fuzz OK: {it} cases, seeds {seed0}..{seed - 1}", flush=True)
    return it


if __name__ == "__main__":
    main(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, int(sys.argv[2]) if len(sys.argv) > 2 else 1,
         int(float(sys.argv[3])) if len(sys.argv) > 3 else 300000)
