#!/usr/bin/env python3
"""Benchmark of the two hot paths on MI355X.

Workload at N=1 (BASELINE.json configs[1]): a 46 Mbp synthetic genome,
FM-index build on device (SA + BWT + sampled SA + rank blocks), and 10 M
50-bp exact-match reads searched per step.  A "step" is one backward-search
pass over the rank's whole read batch with the reads already in HBM.  At N>1
the index is replicated per GPU (each rank builds it) and every rank searches
its own batch of the same size (weak scaling, no data-path collective).

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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--genome-bp", type=int, default=46_000_000)
    ap.add_argument("--reads", type=int, default=10_000_000, help="reads per GPU per step")
    ap.add_argument("--read-len", type=int, default=50)
    ap.add_argument("--build-reps", type=int, default=3)
    ap.add_argument("--cpu-reads", type=int, default=4_000_000)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-callers", action="store_true", help="skip the mms / search_reads / align figures")
    ap.add_argument("--sharded-timeout", type=float, default=240.0,
                    help="seconds the multi-rank sharded build may take before it is reported as stuck")
    ap.add_argument("--cpu-port", action="store_true", help="time the oracle port even if oracle/_ref exists")
    args = ap.parse_args()

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

    def sharded_build():
        """HP1 sharded: one shard of the suffix array per rank, all-gathered into place."""
        try:
            ixs, how = G.build_index_sharded(d_text, n)  # warm-up (workspace, RCCL channels)
            ixs.close()
            barrier()
            t0 = time.perf_counter()
            ixs, how = G.build_index_sharded(d_text, n)
            barrier()
            ts = time.perf_counter() - t0
            same = ixs.C == ix.C and ixs.endloc == ix.endloc
            if same:
                probe, _ = G.gen_reads(d_text, n, 100000, m, 4242, want_offsets=False)
                a = ix.search(probe, m, 100000, want_locrow=True)
                b = ixs.search(probe, m, 100000, want_locrow=True)
                same = all(torch.equal(x, y) for x, y in zip(a, b))
            ixs.close()
            tt = torch.tensor([ts], dtype=torch.float64, device=red_dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            t_max = float(tt.item())
            return {"how": how, "wall_ms": t_max * 1e3, "value": n / t_max / 1e6,
                    "unit": "Mbp/s", "same_as_whole_build": bool(same),
                    "roofline": {"bound": "hbm", "bytes_per_bp": 5.625, "achieved": 5.625 * n / t_max / 1e9,
                                 "peak": HBM_PEAK_GBS * world, "unit": "GB/s",
                                 "frac": 5.625 * n / t_max / 1e9 / (HBM_PEAK_GBS * world)},
                    "what": "text replicated, one bucket range of the SA sorted per rank, shards all-gathered "
                            "over RCCL, every rank finishes its own index"}
        except Exception as e:  # noqa: BLE001 - the search figures do not depend on this
            return {"error": f"{type(e).__name__}: {e}"[:400]}

    import threading
    printed = threading.Lock()

    if rank != 0:
        # a collective that never returns must not keep this rank alive for ever
        dog = threading.Timer(args.sharded_timeout, lambda: os._exit(0))
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

    # ---- the callers of the search path, beside the headline: mms, the search_reads anchor loop,
    # align_read_anchored; 10^6 reads of 100 bases, 30 % with two substitutions, 5 % with an N ----
    callers = None
    if world == 1 and not args.no_callers:
        try:
            callers = time_callers(ix, d_text, n)
        except Exception as e:  # noqa: BLE001 - the headline does not depend on these
            callers = {"error": f"{type(e).__name__}: {e}"[:300]}
    G.timing_enable(False)

    # ---- the same batch on rank blocks alone (what an index loaded from a file has) ----
    ix.compact()
    sp2, ep2, lr2 = (torch.empty_like(sp) for _ in range(3))
    ix.search(d_reads, m, R, want_locrow=True, out=(sp2, ep2, lr2))
    G.timing_enable(True)
    G.timing_reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
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
            "traffic": measured_traffic(f"{n}bp_{R}x{m}"),
            # the kernel is bound by how many cache lines per second leave L2, not by bytes
            "l2_miss_lines_per_s": (measured_traffic(f"{n}bp_{R}x{m}_l2_misses") or 0) / (k_ms * 1e-3) if k_ms else None,
            # what plain random 4-byte gathers over 24 GiB reach on this GPU (tools_probe/gather_probe.hip,
            # profiles/r1_gather_probe.txt): the ceiling for loads with no locality
            "random_gather_lines_per_s": 46e9,
            "bytes_per_read": bpr,
            "loads": loads,
            "counting_pass_matches": counted_ok,
            "kernel_ms": k_ms,
            "what": "bytes the kernel loads and stores, counted per batch by its counting build "
                    "(16 B rank blocks, 16 B text windows, 4 B SA/ISA entries, 8 B table entry, the read itself, "
                    "12 B out)",
            "reads_packed_in_kernel": fused,
        },
        "rank_only": {
            "what": "same batch after dropping SA/ISA/text from the index: every step is two rank queries",
            "value": R * args.steps / ro_elapsed,
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
    if not args.no_cpu and world == 1:
        gs = G.to_numpy_u32(sp[: args.cpu_reads])
        ge = G.to_numpy_u32(ep[: args.cpu_reads])
        result["cpu_baseline"] = cpu_baseline(args, d_text, n, d_reads, R, m, gs, ge)
    if dist is not None:
        # the sharded build comes last, on a timer: if one of its collectives never
        # returns, the line goes out without it
        def give_up():
            if printed.acquire(blocking=False):
                result["sa_build_sharded"] = {"error": f"no result within {args.sharded_timeout} s"}
                print(json.dumps(result), flush=True)
                os._exit(0)

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
