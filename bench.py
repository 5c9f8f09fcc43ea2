#!/usr/bin/env python3
"""Benchmark of the two hot paths on MI355X.

Workload at N=1 (BASELINE.json configs[1]): a 46 Mbp synthetic genome,
FM-index build on device (SA + BWT + sampled SA + rank blocks), and 10 M
50-bp exact-match reads searched per step.  A "step" is one backward-search
pass over the rank's whole read batch with the reads already in HBM.  At N>1
the index is replicated per GPU (each rank builds it) and every rank searches
its own batch of the same size (weak scaling, no data-path collective).

`--config c4` is BASELINE.json configs[3] in the form it names: a 3.1 Gbp genome,
the index replicated on every GPU, ONE set of 100 M x 100 bp reads partitioned
over the ranks (strong scaling), locate included in the step.

Prints ONE JSON line on rank 0.  `value` = reads/s over all ranks; the SA/BWT
build rate, the locate rate, the roofline figures of the dominant kernel and
the CPU baseline ride in the same line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from bwt_amd import gpu as G  # noqa: E402
from bwt_amd import synth  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak


def rank_only_bytes_per_read(m, locate_steps=None):
    """Algorithmic HBM bytes per read of a search made of rank queries alone:
    m/4 packed pattern + one 32-byte index block per rank query (2 per
    backward-search step) + 8 bytes of (sp, ep); with locate, one block per
    LF step + 4 bytes of position."""
    b = m / 4 + 32.0 * 2 * (m - 1) + 8
    if locate_steps is not None:
        b += 32.0 * locate_steps + 4
    return b


def roofline_extras(m, R, k_ms, missed_lines, gather_lines, locate_steps=None):
    """The two other ways to price the search kernel, beside bytes requested / HBM peak.

    survey: the byte count of a search made of rank queries alone (one 32-byte block per query,
    2 (m - 1) queries a read, seqindex.c:280-309 + :91-111; with locate, one block per LF step of
    unc_sa, :244-255) divided by this kernel's time.  The kernel answers loc_search / locate
    without most of those queries (one table lookup for the read's last K bases, one comparison
    with the text for the rest), so this comes out above the HBM peak: it says how many times
    the rank-query formulation's traffic the kernel avoids, not how fast memory is.
    line_rate: what does bound it -- 64-byte lines fetched past the L2 per second (PMC, from the
    committed passes) against plain random 4-byte gathers over 8 GiB measured in this run."""
    bpr = rank_only_bytes_per_read(m, locate_steps)
    ach = bpr * R / (k_ms * 1e-3) / 1e9 if k_ms else 0.0
    out = {"survey": {"bytes_per_read": bpr, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                      "frac": ach / HBM_PEAK_GBS,
                      "what": "m/4 + 32 B x 2(m-1) rank queries" +
                              (f" + 32 B x {locate_steps} LF steps + 4" if locate_steps is not None else "") +
                              " + 8, over this kernel's time; above 1 because the kernel answers without "
                              "most of those queries (k-mer table start + text comparison), results "
                              "bit-identical to the rank-only search of the same batch"}}
    if missed_lines and k_ms:
        lps = missed_lines / (k_ms * 1e-3)
        out["line_rate"] = {"l2_missed_lines_per_launch": missed_lines, "l2_missed_lines_per_s": lps,
                            "lines_per_read": missed_lines / R,
                            "random_gather_lines_per_s": gather_lines,
                            "frac": lps / gather_lines if gather_lines else None,
                            "what": "64-byte lines that miss the L2 per second (TCC_MISS_sum of the committed PMC "
                                    "pass / kernel time of this run) over the rate of plain random 4-byte "
                                    "gathers measured in this run: the bound of a kernel whose loads have no "
                                    "locality; above 1 where records and rank blocks are found in the last-level cache"}
    return out


def time_callers(ix, d_text, n, R=1_000_000, m=100):
    """Kernel rates of the functions that call the search path (seqindex.c:314-351 mms,
    search_reads.c:115-153, single_align.c:163-298) on one batch of edited reads."""
    lib = G._lib.load()
    reads, _ = G.gen_reads(d_text, n, R, m, synth.READS_SEED + 99, want_offsets=False)
    r = reads[: R * m].view(R, m).clone()
    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    rows = torch.nonzero(torch.rand(R, generator=g, device="cuda") < 0.3).squeeze(1)
    for _ in range(2):
        p = torch.randint(12, m, (R,), generator=g, device="cuda")
        r[rows, p[rows]] = (r[rows, p[rows]] + 1) & 3
    rows = torch.nonzero(torch.rand(R, generator=g, device="cuda") < 0.05).squeeze(1)
    r[rows, torch.randint(12, m, (R,), generator=g, device="cuda")[rows]] = 5
    flat = torch.cat([r.reshape(-1), torch.zeros(64, dtype=torch.uint8, device="cuda")])
    rev = torch.cat([torch.flip(r, dims=[1]).reshape(-1), torch.zeros(64, dtype=torch.uint8, device="cuda")])
    offs = torch.arange(R + 1, dtype=torch.int64, device="cuda") * m
    out = {"reads": R, "read_len": m, "what": "kernel time of one batch, HIP events"}

    def timed(name, fn):
        for _ in range(2):
            G.timing_enable(True)
            G.timing_reset()
            fn()
            torch.cuda.synchronize()
        ms = G.timing_report()[name][1]
        return {"value": R / (ms * 1e-3), "unit": "reads/s", "kernel_ms": ms}

    got, sp, ep = (torch.empty(R, dtype=torch.int32, device="cuda") for _ in range(3))
    out["mms"] = timed("fm_mms", lambda: G._lib.check(lib.bwtgpu_mms(
        ix.h, flat.data_ptr(), m, m, R, got.data_ptr(), sp.data_ptr(), ep.data_ptr(), None)))
    res = torch.empty(R * 4, dtype=torch.int32, device="cuda")
    out["search_reads"] = timed("fm_search_reads", lambda: G._lib.check(lib.bwtgpu_search_reads(
        ix.h, flat.data_ptr(), rev.data_ptr(), offs.data_ptr(), R, 20, res.data_ptr(), None)))
    out["search_reads"]["anchored_forward"] = int((res.view(R, 4)[:, 0] > 0).sum())
    if n <= 0x7FFFFFFF:  # the aligner returns positions as int
        holder = {}

        def run_align():
            holder["r"] = G.align_reads(ix, d_text, flat, offs, R, 12, m)
        out["align_reads"] = timed("align_reads", run_align)
        out["align_reads"]["aligned"] = int((holder["r"][0] != 0).sum())
    return out


def measured_traffic(workload_key):
    """HBM bytes per fm_search launch from the committed PMC passes of this
    workload (profiles/pmc_fm_search.json), or None."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_fm_search.json")) as f:
            rec = json.load(f)
        return rec.get(workload_key)
    except (OSError, ValueError):
        return None


def cpu_baseline(args, d_text, n, d_reads, R, m, gpu_sp, gpu_ep):
    """The reference (oracle/_ref, when built) or the oracle port, timed on the
    host cores on a bounded sample of the same workload."""
    from oracle import load_oracle, load_reference

    ref = None if args.cpu_port else load_reference()
    orc = load_oracle()
    text = d_text[: n // 4 + 1].cpu().numpy()
    sample = min(R, args.cpu_reads)
    reads = d_reads[: sample * m].cpu().numpy()
    out = {"cores": 4, "kind": "reference" if ref else "port"}
    t0 = time.perf_counter()
    fmi = ref.make_fmi(text, n) if ref else orc.make_fmi(text, n)
    t_build = time.perf_counter() - t0
    impl = ref if ref else orc
    t0 = time.perf_counter()
    sp, ep = impl.search_batch(fmi, reads, m, m, sample, threads=4)
    t_search = time.perf_counter() - t0
    ok = bool((sp == gpu_sp[:sample]).all() and (ep == gpu_ep[:sample]).all())
    # the loop over reads is embarrassingly parallel: the same sample on this box's share of cores
    many = min(16, os.cpu_count() or 1)
    t0 = time.perf_counter()
    sp2, ep2 = impl.search_batch(fmi, reads, m, m, sample, threads=many)
    t_many = time.perf_counter() - t0
    ok = ok and bool((sp2 == sp).all() and (ep2 == ep).all())
    out.update({
        "value": sample / t_search,
        "unit": "reads/s",
        "sample": f"{sample} of the {R} reads ({m} bp), 4 threads as fmitest.c:70; "
                  f"index built by make_fmi on the same {n} bp genome (4 threads, histsortcomp.c:117)",
        "build_mbp_s": n / t_build / 1e6,
        "build_s": t_build,
        "search_s": t_search,
        "all_cores": {"cores": many, "value": sample / t_many, "unit": "reads/s", "search_s": t_many},
        "host_cpus": os.cpu_count(),
        "gpu_matches_cpu": ok,
    })
    return out


def c5_leg(args):
    """BASELINE.json configs[4]: a 100 Mbp tandem repeat with rare point mutations, the input the
    reference's histsort cannot sort (README.md:76-77): the prefix-doubling path, certified on
    the device, next to the reference's csuff_arr (csacak.c:521-528) on one host core."""
    from oracle import load_oracle, load_reference

    n = 100_000_000
    orc = load_oracle()
    text = orc.gen_tandem(n, 1000, 11, 5000, 3)
    d_text = torch.from_numpy(np.concatenate([text, np.zeros(64, np.uint8)])).cuda()
    G.timing_enable(False)
    G.sa_build(d_text, n)  # warm-up: workspace
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sa = G.sa_build(d_text, n)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = G.sa_last_stats()
    cert = G.sa_certify(d_text, n, sa)
    out = {"workload": "100 Mbp tandem repeat, period 1000, a point mutation every 5000 bases on average",
           "value": n / dt / 1e6, "unit": "Mbp/s", "wall_ms": dt * 1e3,
           "levels": st[0], "unresolved_rows_after_bucket_sort": st[2], "doubling_rounds": st[3],
           "certified_on_device": cert == (0, 0, 0),
           "roofline": {"bound": "hbm", "bytes_per_bp": 4.25, "achieved": 4.25 * n / dt / 1e9, "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": 4.25 * n / dt / 1e9 / HBM_PEAK_GBS}}
    ref = None if (args.no_cpu or args.cpu_port) else load_reference()
    sa_same = None
    if ref is not None:
        t0 = time.perf_counter()
        sa_ref = ref.csuff_arr(text, n)
        dc = time.perf_counter() - t0
        sa_same = bool((np.asarray(sa_ref, np.uint32) == G.to_numpy_u32(sa)).all())
        out["cpu_baseline"] = {"value": n / dc / 1e6, "unit": "Mbp/s", "cores": 1, "kind": "reference",
                               "sample": "the whole 100 Mbp text through csuff_arr (SACA-K is sequential)",
                               "seconds": dc, "gpu_matches_cpu": sa_same}
        del sa_ref
    # ---- backward search on this text: wide intervals, where the k-mer table and the text comparison
    # help least (seqindex.c:298-306 goes on with rank steps while an interval holds several rows) ----
    try:
        R, m = 10_000_000, 50
        sa_h = G.to_numpy_u32(sa) if not args.no_cpu else None
        del sa
        ix = G.build_index_dev(d_text, n)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ix.prepare_search()
        torch.cuda.synchronize()
        prep_ms = (time.perf_counter() - t0) * 1e3
        d_reads, d_offs = G.gen_reads(d_text, n, R, m, synth.READS_SEED + 50)
        sp, ep, pos = (torch.empty(R, dtype=torch.int32, device="cuda") for _ in range(3))
        for _ in range(2):
            ix.search_locate(d_reads, m, R, out=(sp, ep, pos))
        G.timing_enable(True)
        G.timing_reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.c5_steps):
            ix.search_locate(d_reads, m, R, out=(sp, ep, pos))
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.c5_steps
        k_n, k_ms = G.timing_report().get("fm_search", (0, 0.0))
        G.timing_enable(False)
        loads, sp_c, ep_c = ix.search_loads(d_reads, m, R)
        cut = d_offs != -1
        width = (ep - sp).to(torch.int64)
        # locate() of a read gives the text position of the first row of its interval: the read must stand there
        lines = (loads["rank_blocks_16B"] + loads["text_windows_16B"] + loads["sa_isa_entries_4B"] / 2
                 + loads["kmer_entries"]) / R
        leg = {"workload": f"{R} x {m} bp reads cut from this text (1% random), search + locate of the first row, one kernel",
               "value": R / dt, "unit": "reads/s", "ms_per_step": dt * 1e3, "kernel_ms": k_ms / max(k_n, 1),
               "search_arrays_ms": prep_ms,
               "rows_per_interval_mean": float(width[cut].double().mean()),
               "reads_with_one_row": float((width[cut] == 1).double().mean()),
               "requested_lines_per_read": lines, "rank_blocks_per_read": loads["rank_blocks_16B"] / R,
               "counting_pass_matches": bool(torch.equal(sp_c, sp) and torch.equal(ep_c, ep)),
               "every_cut_read_found": bool((width[cut] > 0).all())}
        if sa_h is not None:
            # bit-exact against the oracle's loc_search / locate on a sample, on the index the oracle derives
            # from the suffix array compared with csuff_arr above
            fo = orc.make_fmi(text, n, sa_h)
            S = 20_000
            pick = torch.arange(0, R, R // S, device="cuda")[:S]
            reads_h = d_reads[: R * m].view(R, m)[pick].cpu().numpy()
            sp_o, ep_o = orc.search_batch(fo, reads_h, m, m, S)
            sp_g, ep_g, pos_g = (G.to_numpy_u32(x[pick]) for x in (sp, ep, pos))
            found = ep_o > sp_o
            pos_o, _ = orc.locate_rows(fo, np.where(found, sp_o, 0).astype(np.uint32))
            leg["sample_matches_oracle"] = bool((sp_g == sp_o).all() and (ep_g == ep_o).all()
                                                and (pos_g[found] == pos_o[found]).all())
            leg["sample"] = f"{S} reads, oracle loc_search + unc_sa of the first row"
            del fo
        # the same batch on rank blocks alone
        ix.compact()
        sp2, ep2, lr2 = ix.search(d_reads, m, R, want_locrow=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2):
            ix.search(d_reads, m, R, want_locrow=True, out=(sp2, ep2, lr2))
        torch.cuda.synchronize()
        ro = (time.perf_counter() - t0) / 2
        leg["rank_only"] = {"value": R / ro, "unit": "reads/s", "ms_per_step": ro * 1e3,
                            "same_intervals": bool(torch.equal(sp2, sp) and torch.equal(ep2, ep))}
        out["search"] = leg
        ix.close()
        del d_reads, d_offs, sp, ep, pos, sp2, ep2, lr2
    except Exception as e:  # noqa: BLE001 - the build figure does not depend on this
        out["search"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    del d_text
    G._lib.load().bwtgpu_release_cache()
    return out


def run_c4(args, rank, world, dist, barrier, red_dev, n, R_all, m, steps, warmup):
    """BASELINE.json configs[3]: 3.1 Gbp, index replicated, ONE set of reads partitioned over the
    ranks (the static split of fmitest.c:67-91), each read searched and located.  Strong scaling:
    value = reads of the whole set / time of the slowest rank.  Returns the result on rank 0,
    None on the others."""
    from bwt_amd import dist as D

    d_text = G.gen_genome(n, synth.GENOME_SEED)
    torch.cuda.synchronize()
    G.timing_enable(False)
    build_ms, prep_ms = [], []
    ix = None
    for rep in range(2):  # the second build finds its buffers in the pool
        if ix is not None:
            ix.close()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ix = G.build_index_dev(d_text, n)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        ix.prepare_search()
        torch.cuda.synchronize()
        build_ms.append((t1 - t0) * 1e3)
        prep_ms.append((time.perf_counter() - t1) * 1e3)
    sa_stats = G.sa_last_stats()
    G._lib.load().bwtgpu_release_cache()  # the build workspace (about 17 B/base) goes back before the reads come

    lo, hi = D.slice_bounds(R_all, world, rank)
    R = hi - lo
    d_reads, d_offs = G.gen_reads(d_text, n, R, m, synth.READS_SEED, first=lo)
    sp, ep, lr, pos = (torch.empty(R, dtype=torch.int32, device="cuda") for _ in range(4))

    def step():  # loc_search and locate() of every read: one call, one kernel on an index with position records
        ix.search_locate(d_reads, m, R, out=(sp, ep, pos))

    for _ in range(warmup):
        step()
    G.timing_enable(True)
    G.timing_reset()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    kernels = G.timing_report()
    # the same slice as two calls, search then locate of the rows it names: same answers, and the
    # locate kernel's time beside the one-call step
    G.timing_reset()
    a, b, c = ix.search(d_reads, m, R, want_locrow=True)
    p = ix.locate_rows(c)
    torch.cuda.synchronize()
    two_calls = G.timing_report()
    G.timing_enable(False)
    two_calls_same = bool(torch.equal(a, sp) and torch.equal(b, ep) and torch.equal(p, pos))
    del a, b, c, p
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # what the text itself says about the answers: every read cut from it is found, every unique one
    # located where it was cut; and a position-weighted checksum of this slice's three result arrays
    cut = d_offs != -1

    def u32(x):
        return x.to(torch.int64) & 0xFFFFFFFF

    found_all = bool((u32(ep[cut]) > u32(sp[cut])).all())
    uniq = cut & ((ep - sp) == 1)
    located_ok = bool((pos[uniq] == d_offs[uniq]).all())
    weight = torch.arange(lo + 1, hi + 1, dtype=torch.int64, device="cuda")
    sums = torch.stack([(x.to(torch.int64) * weight).sum() for x in (sp, ep, pos)])
    same_output = None
    if dist is not None:
        all_sums = [torch.zeros(3, dtype=torch.int64, device=red_dev) for _ in range(world)]
        dist.all_gather(all_sums, sums.to(red_dev))
        flags = torch.tensor([int(found_all), int(located_ok)], dtype=torch.int64, device=red_dev)
        dist.all_reduce(flags, op=dist.ReduceOp.MIN)
        found_all, located_ok = bool(flags[0]), bool(flags[1])
        if rank == 0:
            # the single-rank answers for every other rank's slice, computed here on the same index
            same_output = True
            for g in range(1, world):
                glo, ghi = D.slice_bounds(R_all, world, g)
                rg, _ = G.gen_reads(d_text, n, ghi - glo, m, synth.READS_SEED, want_offsets=False, first=glo)
                a, b, c = ix.search(rg, m, ghi - glo, want_locrow=True)
                p = ix.locate_rows(c)
                w = torch.arange(glo + 1, ghi + 1, dtype=torch.int64, device="cuda")
                want = torch.stack([(x.to(torch.int64) * w).sum() for x in (a, b, p)])
                same_output = same_output and bool(torch.equal(want.cpu(), all_sums[g].cpu()))
                del rg, a, b, c, p, w
    if rank != 0:
        return None

    ns, ms_s = kernels.get("fm_search", (0, 0.0))
    nl, ms_l = two_calls.get("fm_locate", (0, 0.0))
    ns2, ms_s2 = two_calls.get("fm_search", (0, 0.0))
    k_ms, l_ms = ms_s / max(ns, 1), ms_l / max(nl, 1)
    loads, sp_c, ep_c = ix.search_loads(d_reads, m, R)
    counted_ok = bool(torch.equal(sp_c, sp) and torch.equal(ep_c, ep))
    bpr = (loads["bytes"] - 4 * loads["pattern_words_4B"]) / R + m + 12.0
    achieved = bpr * R / (k_ms * 1e-3) / 1e9 if k_ms else 0.0
    traffic = measured_traffic(f"{n}bp_{R_all}x{m}") if world == 1 else None
    missed = measured_traffic(f"{n}bp_{R_all}x{m}_l2_misses") if world == 1 else None
    try:
        gather_lines = G.gather_rate(8) if world == 1 else None
    except Exception:  # noqa: BLE001 - a probe, not a result
        gather_lines = None
    extras = roofline_extras(m, R, k_ms, missed, gather_lines, locate_steps=15.5)
    build_s, prep_s = build_ms[-1] / 1e3, prep_ms[-1] / 1e3
    result = {
        "metric": "SA build Mbp/s; FM backward-search reads/s (1/2/4/8 GPU); % HBM peak",
        "value": R_all * steps / elapsed,
        "unit": "reads/s",
        "n_gpus": world,
        "steps": steps,
        "warmup": warmup,
        "ms_per_step": elapsed / steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "u32",
        "data": "synthetic",
        "config": {
            "workload": f"{n / 1e6:g} Mbp splitmix64 genome, FM-index built on every GPU (replicated), ONE set of "
                        f"{R_all} x {m} bp reads (1% random) split into contiguous slices over the GPUs, "
                        f"backward search + locate of every read per step",
            "genome_bp": n, "reads_total": R_all, "reads_this_rank": R, "read_len": m,
        },
        "checks": {"every_cut_read_found": found_all, "unique_reads_located_at_their_offset": located_ok,
                   "same_output_as_single_rank": same_output,
                   "one_call_equals_search_then_locate": two_calls_same},
        "roofline": {
            "kernel": "fm_search", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
            "traffic_source": "profiles/pmc_fm_search.json (builder's rocprofv3 --pmc passes; looked up)"
            if traffic is not None else None,
            "bytes_per_read": bpr, "loads": loads, "counting_pass_matches": counted_ok, "kernel_ms": k_ms,
            "requested_lines_per_read": (loads["rank_blocks_16B"] + loads["text_windows_16B"]
                                         + loads["sa_isa_entries_4B"] / 2 + loads["kmer_entries"]) / R,
            "what": "rank 0's slice; the step is ONE kernel (bwtgpu_search_locate: intervals and positions); bytes "
                    "counted by the kernel's counting build, which leaves out the suffix-array entry a repeated "
                    "read's position costs",
            **extras,
        },
        "search_then_locate": {"search_kernel_ms": ms_s2 / max(ns2, 1), "locate_kernel_ms": l_ms,
                               "locate_value": R / (l_ms * 1e-3) if l_ms else None, "unit": "rows/s",
                               "what": "the same slice as two calls (bwtgpu_search with rows, bwtgpu_locate_rows: one "
                                       "SA entry per row); not part of the timed step"},
        "sa_build": {
            "value": n / build_s / 1e6, "unit": "Mbp/s", "wall_ms": build_s * 1e3,
            "what": "make_fmi on device, whole text on this GPU (every rank builds its replica)",
            "search_arrays_ms": prep_s * 1e3,
            "with_search_arrays": {"wall_ms": (build_s + prep_s) * 1e3, "value": n / (build_s + prep_s) / 1e6,
                                   "unit": "Mbp/s"},
            "roofline": {"bound": "hbm", "bytes_per_bp": 5.625, "achieved": 5.625 * n / build_s / 1e9,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": 5.625 * n / build_s / 1e9 / HBM_PEAK_GBS},
            "stats": {"levels": sa_stats[0], "leaves": sa_stats[1], "unresolved_rows": sa_stats[2],
                      "doubling_rounds": sa_stats[3]},
        },
        "cpu_baseline": None,
        "cpu_baseline_note": "the reference cannot run this configuration: its lengths are int "
                             "(seqindex.h:14-22, csacak.c:14) and 3.1e9 > 2^31 - 1; see the c2 line for the "
                             "reference timed on this box",
        "index_bytes": ix.device_bytes,
    }
    ix.close()
    del d_reads, d_offs, sp, ep, lr, pos, d_text
    G._lib.load().bwtgpu_release_cache()
    return result


def bench_c4(args, rank, world, dist, barrier, red_dev):
    result = run_c4(args, rank, world, dist, barrier, red_dev, args.genome_bp, args.reads, args.read_len,
                    args.steps, args.warmup)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", choices=["c2", "c4"], default="c2",
                    help="c2: 46 Mbp, 10 M x 50 bp per GPU (weak); c4: 3.1 Gbp, 100 M x 100 bp over all GPUs (strong)")
    ap.add_argument("--genome-bp", type=int, default=None)
    ap.add_argument("--reads", type=int, default=None, help="c2: reads per GPU per step; c4: reads in the one set")
    ap.add_argument("--read-len", type=int, default=None)
    ap.add_argument("--no-c5", action="store_true", help="skip the tandem-repeat build leg (c2, one GPU)")
    ap.add_argument("--no-c4", action="store_true",
                    help="skip the 3.1 Gbp / 100 M x 100 bp leg of the default line (c2, one GPU)")
    ap.add_argument("--c4-steps", type=int, default=100, help="timed steps of that leg")
    ap.add_argument("--c5-steps", type=int, default=5, help="timed search steps on the tandem-repeat text")
    ap.add_argument("--build-reps", type=int, default=3)
    ap.add_argument("--cpu-reads", type=int, default=4_000_000)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-callers", action="store_true", help="skip the mms / search_reads / align figures")
    ap.add_argument("--sharded-timeout", type=float, default=240.0,
                    help="seconds the multi-rank sharded build may take before it is reported as stuck")
    ap.add_argument("--sharded-genome-bp", type=int, default=250_000_000,
                    help="N>1: genome of the second sharded-build leg (BASELINE.json configs[2]); 0 skips it")
    ap.add_argument("--cpu-port", action="store_true", help="time the oracle port even if oracle/_ref exists")
    args = ap.parse_args()
    dflt = {"c2": (46_000_000, 10_000_000, 50), "c4": (3_100_000_000, 100_000_000, 100)}[args.config]
    args.genome_bp = args.genome_bp or dflt[0]
    args.reads = args.reads or dflt[1]
    args.read_len = args.read_len or dflt[2]

    if os.environ.get("BWT_BENCH_TRACE_AFTER"):  # debugging aid: dump every thread's stack and exit
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["BWT_BENCH_TRACE_AFTER"]), exit=True)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if os.environ.get("BWT_BENCH_REHEARSAL"):
            # rehearsal of the multi-rank code on a one-GPU box: every rank on device 0, gloo
            local_rank = 0
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    G.init(local_rank)

    n, R, m = args.genome_bp, args.reads, args.read_len
    # timing words travel as device tensors over RCCL, as host tensors over gloo
    red_dev = "cpu" if os.environ.get("BWT_BENCH_REHEARSAL") else "cuda"

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if args.config == "c4":
        return bench_c4(args, rank, world, dist, barrier, red_dev)

    # ---- inputs, generated in HBM (same genome on every rank, own reads) ----
    d_text = G.gen_genome(n, synth.GENOME_SEED)
    torch.cuda.synchronize()

    # ---- HP1: FM-index build from the HBM-resident text ----
    # wall time over build_reps builds with launch timing off (two HIP events per
    # launch are not free), then one more build with it on for the per-kernel split
    # The build proper ends when what make_fmi produces is in HBM; the arrays that only
    # speed searches up (position records, k-mer table) are made on first search and are
    # timed here as their own item.
    G.timing_enable(False)
    ix = None
    build_ms, prep_ms = [], []
    for rep in range(args.build_reps + 1):
        if ix is not None:
            ix.close()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ix = G.build_index_dev(d_text, n)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        ix.prepare_search()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        if rep >= 1 or args.build_reps == 0:
            build_ms.append((t1 - t0) * 1e3)
            prep_ms.append((t2 - t1) * 1e3)
    ix.close()
    G.timing_enable(True)
    G.timing_reset()
    ix = G.build_index_dev(d_text, n)
    torch.cuda.synchronize()
    build_kernels = G.timing_report()
    G.timing_reset()
    ix.prepare_search()
    torch.cuda.synchronize()
    prep_kernels = G.timing_report()
    sa_stats = G.sa_last_stats()
    build_s = float(np.median(build_ms)) / 1e3
    prep_s = float(np.median(prep_ms)) / 1e3

    sharded = None
    if dist is None:
        # one GPU: the eight shards of an 8-rank build, one after the other
        try:
            from bwt_amd import dist as D
            counts = G.sa_bucket_counts(d_text, n)
            ranges = D.plan_bucket_ranges(counts, 8)
            rows_of = D.shard_rows(counts, ranges)
            sa_t = torch.empty(n + 1, dtype=torch.int32, device="cuda")
            bw_t = torch.empty(n + 1, dtype=torch.uint8, device="cuda")
            ws_t = torch.empty(int(G._lib.load().bwtgpu_sa_workspace_bytes(n)), dtype=torch.uint8, device="cuda")
            ms = []
            for rep in range(2):
                ms = []
                for (lo_b, hi_b), (b0, r0) in zip(ranges, rows_of):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    G.sa_build_range(d_text, n, lo_b, hi_b, sa_t[b0:b0 + r0], bw_t[b0:b0 + r0], ws_t)
                    torch.cuda.synchronize()
                    ms.append((time.perf_counter() - t0) * 1e3)
            sharded = {"logical_shards": 8, "shard_ms": [round(x, 3) for x in ms], "max_shard_ms": max(ms),
                       "shard_rows": [r for _, r in rows_of],
                       "what": "one GPU sorting each of the 8 bucket ranges in turn: the per-rank work of an "
                               "8-GPU build, without the exchange"}
            del sa_t, bw_t, ws_t
        except Exception as e:  # noqa: BLE001
            sharded = {"error": f"{type(e).__name__}: {e}"[:400]}

    d_reads, _ = G.gen_reads(d_text, n, R, m, synth.READS_SEED + rank, want_offsets=False)
    sp = torch.empty(R, dtype=torch.int32, device="cuda")
    ep = torch.empty(R, dtype=torch.int32, device="cuda")
    lr = torch.empty(R, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()

    # ---- HP2: backward search, timed region ----
    G.timing_enable(False)
    for _ in range(args.warmup):
        ix.search(d_reads, m, R, want_locrow=True, out=(sp, ep, lr))
    G.timing_enable(True)
    G.timing_reset()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ix.search(d_reads, m, R, want_locrow=True, out=(sp, ep, lr))
    barrier()
    elapsed = time.perf_counter() - t0
    search_kernels = G.timing_report()
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- locate (unc_sa) of every read's first row: reported beside, not in value ----
    def time_locate():
        G.timing_reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pos, steps = ix.locate_rows(lr, count_steps=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        launches, ms = G.timing_report().get("fm_locate", (0, 0.0))
        return pos, steps, dt, ms / max(launches, 1)

    pos, _, locate_s, loc_ms = time_locate()
    located = int((pos != -1).sum().item())

    phase = {"at": "start"}

    def sharded_build():
        """HP1 sharded: one bucket range of the suffix array sorted per rank; the ranks all-gather
        the index products of their rows (2-bit BWT symbols + every 8th row), every rank assembles
        its index.  Timed with and without the second exchange (the rows themselves, for the
        search-only position records).  Twice: on the genome of this run, and on BASELINE.json
        configs[2] -- a 250 Mbp genome, the SA build sharded by bucket range -- with the whole
        build of the same text on one GPU timed beside it."""

        def one_text(text, nn, whole_ix, tag):
            res = {}
            for key, want_sa in (("index", False), ("index_and_suffix_array", True)):
                phase["at"] = f"sharded build {tag}({key}) warm-up"
                ixs, how = G.build_index_sharded(text, nn, want_sa=want_sa)  # workspace, RCCL channels
                ixs.close()
                phase["at"] = f"sharded build {tag}({key}) barrier"
                barrier()
                t0 = time.perf_counter()
                phase["at"] = f"sharded build {tag}({key}) timed run"
                ixs, how = G.build_index_sharded(text, nn, want_sa=want_sa)
                barrier()
                ts = time.perf_counter() - t0
                same = ixs.C == whole_ix.C and ixs.endloc == whole_ix.endloc
                if same:
                    probe, _ = G.gen_reads(text, nn, 100000, m, 4242, want_offsets=False)
                    a = whole_ix.search(probe, m, 100000, want_locrow=True)
                    b = ixs.search(probe, m, 100000, want_locrow=True)
                    same = all(torch.equal(x, y) for x, y in zip(a, b))
                    if want_sa:
                        rep = ixs.certify(text)
                        same = same and all(v == 0 for v in rep.values())
                ixs.close()
                phase["at"] = f"sharded build {tag}({key}) all-reduce of the time"
                tt = torch.tensor([ts], dtype=torch.float64, device=red_dev)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                t_max = float(tt.item())
                res[key] = {"how": how, "wall_ms": t_max * 1e3, "value": nn / t_max / 1e6, "unit": "Mbp/s",
                            "same_as_whole_build": bool(same),
                            "roofline": {"bound": "hbm", "bytes_per_bp": 5.625,
                                         "achieved": 5.625 * nn / t_max / 1e9, "peak": HBM_PEAK_GBS * world,
                                         "unit": "GB/s", "frac": 5.625 * nn / t_max / 1e9 / (HBM_PEAK_GBS * world)}}
            return res

        def exchange_probe():
            """Measured bandwidth of the two collectives a sharded build could use: an all-gather
            (what it uses) and an all-to-all (the id exchange it does not), 64 MB per rank."""
            per = 64 << 20
            dev = "cpu" if os.environ.get("BWT_BENCH_REHEARSAL") else "cuda"
            if dev == "cpu":
                per = 4 << 20
            per -= per % world
            src = torch.ones(per, dtype=torch.uint8, device=dev)
            gathered = torch.empty(per * world, dtype=torch.uint8, device=dev)
            dealt = torch.empty(per, dtype=torch.uint8, device=dev)
            res = {"bytes_per_rank": per}
            for name, fn in (("all_gather", lambda: dist.all_gather_into_tensor(gathered, src)),
                             ("all_to_all", lambda: dist.all_to_all_single(dealt, src))):
                phase["at"] = f"exchange probe: {name}"
                try:
                    fn()
                    barrier()
                    t0 = time.perf_counter()
                    for _ in range(5):
                        fn()
                    barrier()
                    dtp = (time.perf_counter() - t0) / 5
                    tt = torch.tensor([dtp], dtype=torch.float64, device=red_dev)
                    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                    recv = per * (world - 1) if name == "all_gather" else per * (world - 1) // world
                    res[name] = {"ms": float(tt.item()) * 1e3, "received_bytes_per_rank": recv,
                                 "gb_per_s_per_rank": recv / float(tt.item()) / 1e9}
                except Exception as e:  # noqa: BLE001 - a probe
                    res[name] = {"error": f"{type(e).__name__}: {e}"[:200]}
            return res

        try:
            out = one_text(d_text, n, ix, "")
            out["exchange_probe"] = exchange_probe()
            out["what"] = ("text replicated; each rank sorts one bucket range and the ranks all-gather 0.75 B/base "
                           "of index products over RCCL (index); then 4 B/base of rows for the position records "
                           "(index_and_suffix_array)")
            n3 = args.sharded_genome_bp
            if n3:
                phase["at"] = "250 Mbp genome: whole build on one GPU"
                text3 = G.gen_genome(n3, synth.GENOME_SEED)
                whole_ms = []
                whole = None
                for _ in range(3):
                    if whole is not None:
                        whole.close()
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    whole = G.build_index_dev(text3, n3)
                    torch.cuda.synchronize()
                    whole_ms.append((time.perf_counter() - t0) * 1e3)
                big = one_text(text3, n3, whole, "of the 250 Mbp genome ")
                whole.close()
                big["genome_bp"] = n3
                big["whole_build_one_gpu_ms"] = min(whole_ms[1:])
                big["index_ready_vs_one_gpu"] = big["index"]["wall_ms"] / big["whole_build_one_gpu_ms"]
                big["what"] = ("BASELINE.json configs[2]: SA build of a 250 Mbp genome sharded by bucket range over "
                               "the ranks; whole_build_one_gpu_ms is make_fmi of the same text on this rank alone")
                out["c3_250mbp"] = big
                del text3
            phase["at"] = "done"
            return out
        except Exception as e:  # noqa: BLE001 - the search figures do not depend on this
            return {"error": f"{type(e).__name__}: {e}"[:400], "phase": phase["at"]}

    import threading
    printed = threading.Lock()

    if rank != 0:
        # a collective that never returns must not keep this rank alive for ever; a hang is a failure
        def stuck():
            print(f"rank {rank}: no progress for {args.sharded_timeout} s in: {phase['at']}", file=sys.stderr, flush=True)
            os._exit(3)

        dog = threading.Timer(args.sharded_timeout, stuck)
        dog.daemon = True
        dog.start()
        sharded_build()
        dog.cancel()
        dist.destroy_process_group()
        return

    n_launch, ms_total = search_kernels.get("fm_search", (0, 0.0))
    k_ms = ms_total / max(n_launch, 1)
    # bytes the search loads and stores, counted by the kernel itself on this batch
    G.timing_enable(False)
    loads, sp_c, ep_c = ix.search_loads(d_reads, m, R)
    counted_ok = bool(torch.equal(sp_c, sp) and torch.equal(ep_c, ep))
    # + sp, ep, locate row written; when the kernel packs the reads itself (no pack_reads
    # launch) it reads their m bytes once instead of the packed words
    fused = "pack_reads" not in search_kernels
    bpr = (loads["bytes"] - (4 * loads["pattern_words_4B"] if fused else 0)) / R + (m if fused else 0) + 12.0
    achieved = bpr * R / (k_ms * 1e-3) / 1e9 if k_ms else 0.0
    index_bytes = ix.device_bytes
    traffic = measured_traffic(f"{n}bp_{R}x{m}")
    try:
        gather_lines = G.gather_rate(8) if world == 1 else None
    except Exception:  # noqa: BLE001 - a probe, not a result
        gather_lines = None

    # ---- the callers of the search path, beside the headline: mms, the search_reads anchor loop,
    # align_read_anchored; 10^6 reads of 100 bases, 30 % with two substitutions, 5 % with an N ----
    callers = None
    if world == 1 and not args.no_callers:
        try:
            callers = time_callers(ix, d_text, n)
        except Exception as e:  # noqa: BLE001 - the headline does not depend on these
            callers = {"error": f"{type(e).__name__}: {e}"[:300]}
    G.timing_enable(False)

    # ---- the same batch handed over already packed to 2 bits per base (bwtgpu_search_packed):
    # beside the headline, whose reads are one base per byte as the reference's callers hold them ----
    side_steps = min(args.steps, 50)  # the legs beside the headline
    packed_leg = None
    if world == 1:
        try:
            d_words = G.pack_reads_dev(d_reads, m, R)
            sp3, ep3, lr3 = (torch.empty_like(sp) for _ in range(3))
            ix.search_packed(d_words, m, R, want_locrow=True, out=(sp3, ep3, lr3))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(side_steps):
                ix.search_packed(d_words, m, R, want_locrow=True, out=(sp3, ep3, lr3))
            torch.cuda.synchronize()
            pk = time.perf_counter() - t0
            packed_leg = {"what": "same reads resident in HBM as 2-bit words (4 words a read); same kernel",
                          "value": R * side_steps / pk, "unit": "reads/s", "ms_per_step": pk / side_steps * 1e3,
                          "same_output": bool(torch.equal(sp3, sp) and torch.equal(ep3, ep) and torch.equal(lr3, lr))}
            del d_words, sp3, ep3, lr3
        except Exception as e:  # noqa: BLE001 - the headline does not depend on this
            packed_leg = {"error": f"{type(e).__name__}: {e}"[:300]}

    # ---- the same batch on an index loaded from its stored parts (what read_index hands over,
    # fileio.c:24-65): its first search recomputes suffix array and text from BWT and sampled rows ----
    loaded_leg = None
    if world == 1:
        try:
            bwt_h, idxs_h = ix.export()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ixl = G.index_from_parts(bwt_h, n, ix.endloc, idxs_h)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            ixl.prepare_search()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            sp4, ep4, lr4 = (torch.empty_like(sp) for _ in range(3))
            ixl.search(d_reads, m, R, want_locrow=True, out=(sp4, ep4, lr4))
            torch.cuda.synchronize()
            t3 = time.perf_counter()
            for _ in range(side_steps):
                ixl.search(d_reads, m, R, want_locrow=True, out=(sp4, ep4, lr4))
            torch.cuda.synchronize()
            ld = time.perf_counter() - t3
            loaded_leg = {"what": "index exported to the stored parts (BWT bytes, every 32nd row) and loaded again; "
                                  "suffix array and text recomputed on the device from those, then the same arrays "
                                  "and the same kernel as the built index",
                          "load_ms": (t1 - t0) * 1e3, "recover_and_search_arrays_ms": (t2 - t1) * 1e3,
                          "value": R * side_steps / ld, "unit": "reads/s", "ms_per_step": ld / side_steps * 1e3,
                          "same_output": bool(torch.equal(sp4, sp) and torch.equal(ep4, ep) and torch.equal(lr4, lr)),
                          "index_bytes": ixl.device_bytes}
            ixl.close()
            # the third form: loaded and kept lean (rank blocks, sampled rows, a one-word table of the
            # k-mer intervals' starts; no suffix array, text or position records)
            ixn = G.index_from_parts(bwt_h, n, ix.endloc, idxs_h, lean=True)
            ixn.prepare_search()
            ixn.search(d_reads, m, R, want_locrow=True, out=(sp4, ep4, lr4))
            torch.cuda.synchronize()
            lean_steps = min(side_steps, 10)
            t4 = time.perf_counter()
            for _ in range(lean_steps):
                ixn.search(d_reads, m, R, want_locrow=True, out=(sp4, ep4, lr4))
            torch.cuda.synchronize()
            ln = time.perf_counter() - t4
            loaded_leg["kept_lean"] = {
                "what": "the same parts loaded with lean=True: rank blocks, sampled rows and a one-word k-mer table",
                "value": R * lean_steps / ln, "unit": "reads/s", "ms_per_step": ln / lean_steps * 1e3,
                "same_output": bool(torch.equal(sp4, sp) and torch.equal(ep4, ep) and torch.equal(lr4, lr)),
                "index_bytes": ixn.device_bytes}
            ixn.close()
            del sp4, ep4, lr4
        except Exception as e:  # noqa: BLE001 - the headline does not depend on this
            loaded_leg = {"error": f"{type(e).__name__}: {e}"[:300]}

    # ---- the same batch on rank blocks alone (an index kept lean, or one without room for the arrays) ----
    ix.compact()
    sp2, ep2, lr2 = (torch.empty_like(sp) for _ in range(3))
    ix.search(d_reads, m, R, want_locrow=True, out=(sp2, ep2, lr2))
    G.timing_enable(True)
    G.timing_reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(side_steps):
        ix.search(d_reads, m, R, want_locrow=True, out=(sp2, ep2, lr2))
    torch.cuda.synchronize()
    ro_elapsed = time.perf_counter() - t0
    ro_launch, ro_total = G.timing_report().get("fm_search", (0, 0.0))
    ro_ms = ro_total / max(ro_launch, 1)
    ro_same = bool(torch.equal(sp2, sp) and torch.equal(ep2, ep) and torch.equal(lr2, lr))
    pos2, lf_steps, ro_locate_s, ro_loc_ms = time_locate()
    ro_same = ro_same and bool(torch.equal(pos2, pos))
    ro_bpr = rank_only_bytes_per_read(m)
    ro_achieved = ro_bpr * R / (ro_ms * 1e-3) / 1e9 if ro_ms else 0.0
    loc_bytes = 32.0 * lf_steps + 8.0 * located
    sa_ms = sum(v[1] for k, v in build_kernels.items() if k.startswith(("sa_", "dbl_", "rs_", "scan_")))
    result = {
        "metric": "SA build Mbp/s; FM backward-search reads/s (1/2/4/8 GPU); % HBM peak",
        "value": world * R * args.steps / elapsed,
        "unit": "reads/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u32",
        "data": "synthetic",
        "config": {
            "workload": f"{n / 1e6:g} Mbp splitmix64 genome, FM-index built on device, "
                        f"{R} x {m} bp exact-match reads per GPU per step (1% random), index replicated",
            "genome_bp": n, "reads_per_gpu": R, "read_len": m,
        },
        "roofline": {
            "kernel": "fm_search",
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic,
            "traffic_source": ("profiles/pmc_fm_search.json: rocprofv3 --pmc passes of this workload taken by "
                               "the builder (TCC_EA0_RDREQ/WRREQ families, corrected as the microarchitecture guide "
                               "says); looked up, not measured in this run") if traffic is not None else None,
            # the kernel is bound by how many cache lines per second leave L2, not by bytes
            "l2_miss_lines_per_s_from_profiles": (measured_traffic(f"{n}bp_{R}x{m}_l2_misses") or 0) / (k_ms * 1e-3)
            if k_ms and traffic is not None else None,
            # plain random 4-byte gathers over 8 GiB, measured in this run (bwtgpu_diag_gather_rate):
            # the ceiling for loads with no locality
            "random_gather_lines_per_s": gather_lines,
            # distinct lines a read asks for, from the counting build: rank blocks, record lines (the
            # text window and the ISA entry share one), SA entries, table entries
            "requested_lines_per_read": (loads["rank_blocks_16B"] + loads["text_windows_16B"]
                                         + loads["sa_isa_entries_4B"] / 2 + loads["kmer_entries"]) / R,
            "bytes_per_read": bpr,
            "loads": loads,
            "counting_pass_matches": counted_ok,
            "kernel_ms": k_ms,
            "what": "bytes the kernel loads and stores, counted per batch by its counting build "
                    "(16 B rank blocks, 16 B text windows, 4 B SA/ISA entries and further table starts, 8 B table entry, "
                    "the read itself, 12 B out)",
            "reads_packed_in_kernel": fused,
            **roofline_extras(m, R, k_ms, measured_traffic(f"{n}bp_{R}x{m}_l2_misses"), gather_lines),
        },
        "packed_input": packed_leg,
        "loaded_index": loaded_leg,
        "rank_only": {
            "what": "same batch after dropping SA/ISA/text from the index: every step is two rank queries",
            "value": R * side_steps / ro_elapsed,
            "unit": "reads/s",
            "kernel_ms": ro_ms,
            "same_output": ro_same,
            "roofline": {"bound": "hbm", "bytes_per_read": ro_bpr, "achieved": ro_achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ro_achieved / HBM_PEAK_GBS},
            "locate": {"value": R / ro_locate_s, "unit": "rows/s",
                       "lf_steps_per_row": lf_steps / max(located, 1), "kernel_ms": ro_loc_ms,
                       "achieved_gbs": loc_bytes / (ro_loc_ms * 1e-3) / 1e9 if ro_loc_ms else 0.0},
            "index_bytes": ix.device_bytes,
        },
        "sa_build": {
            "value": n / build_s / 1e6,
            "unit": "Mbp/s",
            "wall_ms": build_s * 1e3,
            "what": "make_fmi on device: text in HBM -> SA, BWT, sampled SA, rank blocks, C[] "
                    "(whole SA and word-packed text kept in HBM)",
            "sa_kernels_ms": sa_ms,
            "search_arrays": {
                "what": "position records (inverse SA + text) and k-mer table, made once on first search",
                "wall_ms": prep_s * 1e3,
                "kernels_ms": {k: round(v[1], 4) for k, v in sorted(prep_kernels.items())},
            },
            "with_search_arrays": {"wall_ms": (build_s + prep_s) * 1e3,
                                   "value": n / (build_s + prep_s) / 1e6, "unit": "Mbp/s"},
            "roofline": {
                "bound": "hbm", "bytes_per_bp": 5.625,
                "achieved": 5.625 * n / build_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": 5.625 * n / build_s / 1e9 / HBM_PEAK_GBS,
                # HBM bytes one build moves over all its kernels (a sort makes several passes), from the
                # committed PMC passes of this workload
                "traffic": measured_traffic(f"{n}bp_build"),
                "traffic_source": "profiles/pmc_fm_search.json (builder's rocprofv3 --pmc passes; looked up, "
                                  "not measured in this run)",
            },
            "stats": {"levels": sa_stats[0], "leaves": sa_stats[1], "unresolved_rows": sa_stats[2],
                      "doubling_rounds": sa_stats[3]},
            "kernels_ms": {k: round(v[1], 4) for k, v in sorted(build_kernels.items())},
        },
        "locate": {
            "value": R / locate_s,
            "unit": "rows/s",
            "what": "one SA entry per row (the index keeps the whole suffix array)",
            "kernel_ms": loc_ms,
            "achieved_gbs": 12.0 * located / (loc_ms * 1e-3) / 1e9 if loc_ms else 0.0,
        },
        "sa_build_sharded": sharded,
        "callers": callers,
        "search_kernels_ms": {k: round(v[1] / max(v[0], 1), 4) for k, v in sorted(search_kernels.items())},
        "index_bytes": index_bytes,
    }
    if world == 1 and not args.no_c5:
        try:
            result["c5_repeats"] = c5_leg(args)
        except Exception as e:  # noqa: BLE001 - the headline does not depend on this
            result["c5_repeats"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    if not args.no_cpu and world == 1:
        gs = G.to_numpy_u32(sp[: args.cpu_reads])
        ge = G.to_numpy_u32(ep[: args.cpu_reads])
        result["cpu_baseline"] = cpu_baseline(args, d_text, n, d_reads, R, m, gs, ge)
    if world == 1 and not args.no_c4:
        # ---- BASELINE.json configs[3] on this one GPU, after everything of the 46 Mbp legs has been released ----
        try:
            ix.close()
            del ix, d_text, d_reads, sp, ep, lr, sp2, ep2, lr2, pos, pos2
            G._lib.load().bwtgpu_release_cache()
            torch.cuda.empty_cache()
            t0 = time.perf_counter()
            c4 = run_c4(args, 0, 1, None, barrier, red_dev, 3_100_000_000, 100_000_000, 100, args.c4_steps, 2)
            c4["leg_wall_s"] = time.perf_counter() - t0
            for k in ("metric", "higher_is_better", "vs_baseline", "data", "n_gpus"):
                c4.pop(k, None)
            result["c4"] = c4
        except Exception as e:  # noqa: BLE001 - the headline does not depend on this
            result["c4"] = {"error": f"{type(e).__name__}: {e}"[:400]}
    if dist is not None:
        # the sharded build comes last, on a timer: if one of its collectives never
        # returns, the line goes out without it
        def give_up():
            if printed.acquire(blocking=False):
                result["sa_build_sharded"] = {"error": f"no result within {args.sharded_timeout} s",
                                              "stuck_in": phase["at"], "rank": rank}
                print(json.dumps(result), flush=True)
                os._exit(3)  # the line is out, but a hang is not a successful run

        dog = threading.Timer(args.sharded_timeout, give_up)
        dog.daemon = True
        dog.start()
        out = sharded_build()
        dog.cancel()
        if not printed.acquire(blocking=False):
            return  # the timer is printing
        result["sa_build_sharded"] = out
    print(json.dumps(result), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
