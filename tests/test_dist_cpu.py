"""The multi-rank paths on CPU: two processes over gloo run the same
orchestration code the GPU ranks run (bwt_amd.dist), with the oracle standing
in for the device work of each rank."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from bwt_amd import dist as D  # noqa: E402
from bwt_amd import synth  # noqa: E402


def test_sample_span_lists_every_eighth_row():
    for begin, rows in ((1, 0), (1, 5), (1, 7), (1, 8), (8, 1), (9, 7), (9, 8), (16, 17), (12345, 1000)):
        first, i0, cnt = D.sample_span(begin, rows)
        want = [r for r in range(begin, begin + rows) if r % 8 == 0]
        assert cnt == len(want)
        if want:
            assert begin + first == want[0] and i0 == want[0] // 8


def test_slice_bounds_cover_everything_once():
    for total in (0, 1, 7, 100, 10_000_001):
        for world in (1, 2, 3, 8):
            edges = [D.slice_bounds(total, world, r) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == total
            for (a, b), (c, d) in zip(edges, edges[1:]):
                assert b == c and b >= a
            sizes = [b - a for a, b in edges]
            assert max(sizes) - min(sizes) <= 1


def test_plan_bucket_ranges_balances_and_covers():
    rng = np.random.default_rng(3)
    for world in (1, 2, 4, 8):
        counts = rng.integers(1000, 2000, 256).astype(np.uint64)
        ranges = D.plan_bucket_ranges(counts, world)
        assert ranges[0][0] == 0 and ranges[-1][1] == 256
        assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
        rows = D.shard_rows(counts, ranges)
        assert rows[0][0] == 1 and sum(r for _, r in rows) == int(counts.sum())
        assert max(r for _, r in rows) < 1.1 * counts.sum() / world + 2000
    # one bucket outweighs everything: ranges stay ordered, some are empty
    skew = np.zeros(256, np.uint64)
    skew[0] = 10 ** 6
    skew[200] = 5
    ranges = D.plan_bucket_ranges(skew, 8)
    assert ranges[0][0] == 0 and ranges[-1][1] == 256
    assert all(lo <= hi for lo, hi in ranges)
    assert sum(r for _, r in D.shard_rows(skew, ranges)) == 10 ** 6 + 5


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _digits(bases, n):
    padded = np.concatenate([bases, np.zeros(3, np.uint8)]).astype(np.int64)
    return ((padded[:-3] << 6) | (padded[1:-2] << 4) | (padded[2:-1] << 2) | padded[3:])[:n]


def _worker(rank, world, port, n, R, m, fail_rank, q, skewed=False):
    try:
        from oracle import load_oracle

        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
        orc = load_oracle()
        if skewed:  # one leading string outweighs a share: shards of very different sizes
            rng = np.random.default_rng(11)
            text = synth.pack(rng.choice(4, n, p=[0.91, 0.03, 0.03, 0.03]).astype(np.uint8))
        else:
            text = orc.gen_genome(n, synth.GENOME_SEED)
        bases = synth.unpack(text, n)
        sa_o = orc.sa_doubling(text, n) if skewed else orc.histsort(text, n)
        counts = np.bincount(_digits(bases, n), minlength=256).astype(np.uint64)
        bw_o = np.zeros(n + 1, np.uint8)
        nz = sa_o > 0
        bw_o[nz] = bases[sa_o[nz].astype(np.int64) - 1]
        endloc_o = int(np.nonzero(sa_o == 0)[0][0])

        def build_range(lo, hi, rows_out, bw_out):
            if rank == fail_rank:
                raise D.NeedsWholeText("asked to fail")
            begin, rows = D.shard_rows(counts, [(lo, hi)])[0]
            rows_out.copy_(torch.from_numpy(sa_o[begin:begin + rows].view(np.int32)))
            bw_out.copy_(torch.from_numpy(bw_o[begin:begin + rows]))
            e = endloc_o if begin <= endloc_o < begin + rows else 0
            return begin, rows, e

        sa = torch.full((n + 1,), -7, dtype=torch.int32)
        bw = torch.zeros(n + 1, dtype=torch.uint8)
        if fail_rank >= 0:
            try:
                D.build_sa_sharded(n, counts, build_range, sa, bw)
                q.put((rank, "no error"))
            except D.NeedsWholeText:
                q.put((rank, "ok"))
            return
        endloc = D.build_sa_sharded(n, counts, build_range, sa, bw)
        assert endloc == endloc_o
        np.testing.assert_array_equal(sa.numpy().view(np.uint32), sa_o)
        np.testing.assert_array_equal(bw.numpy()[1:], bw_o[1:])

        # the index-products exchange: 2-bit symbols and every 8th row of each rank's own rows
        def pack_bw2(bw_t, rows):
            b = np.concatenate([bw_t.numpy()[:rows], np.zeros((-rows) % 4, np.uint8)]).reshape(-1, 4)
            return torch.from_numpy((b[:, 0] | (b[:, 1] << 2) | (b[:, 2] << 4) | (b[:, 3] << 6)).astype(np.uint8))

        mine = D.shard_rows(counts, D.plan_bucket_ranges(counts, world))[rank][1]
        rows_buf = torch.full((max(mine, 1),), -7, dtype=torch.int32)
        bw_buf = torch.zeros(max(mine, 1), dtype=torch.uint8)
        parts = D.build_index_parts_sharded(n, counts, build_range, pack_bw2, rows_buf, bw_buf)
        assert parts.endloc == endloc_o
        np.testing.assert_array_equal(parts.samples.numpy().view(np.uint32), sa_o[::8])
        packed = parts.bw2.numpy()
        for g, (b, r) in enumerate(parts.rows_of):
            j = np.arange(r)
            sym = (packed[parts.bw2_offsets[g] + (j >> 2)] >> (2 * (j & 3))) & 3
            np.testing.assert_array_equal(sym, bw_o[b:b + r])
        sa2 = torch.full((n + 1,), -7, dtype=torch.int32)
        D.gather_suffix_array(n, parts.rows_of, rows_buf, sa2)
        np.testing.assert_array_equal(sa2.numpy().view(np.uint32), sa_o)

        # reads: every rank searches its slice with the oracle; rank 0 gets them all in order
        fo = orc.make_fmi(text, n, sa_o)
        reads, _ = orc.gen_reads(text, n, R, m, synth.READS_SEED)

        def search_slice(lo, hi):
            sp, ep = orc.search_batch(fo, reads[lo:hi], m, m, hi - lo)
            pos = np.array([orc.locate(fo, reads[r]) for r in range(lo, hi)], np.int64)
            return (torch.from_numpy(sp.view(np.int32).copy()), torch.from_numpy(ep.view(np.int32).copy()),
                    torch.from_numpy(pos))

        lo, hi, sp, ep, pos = D.search_sharded(R, search_slice)
        assert (lo, hi) == D.slice_bounds(R, world, rank) and len(sp) == hi - lo
        got = D.search_sharded(R, search_slice, gather_to=0)
        if rank == 0:
            sp_all, ep_all = orc.search_batch(fo, reads, m, m, R)
            np.testing.assert_array_equal(got[0].numpy().view(np.uint32), sp_all)
            np.testing.assert_array_equal(got[1].numpy().view(np.uint32), ep_all)
            assert got[2][lo:hi].tolist() == pos.tolist()
        else:
            assert got == (None, None, None)
        q.put((rank, "ok"))
    except Exception as e:  # report instead of hanging the peer
        import traceback
        q.put((rank, "".join(traceback.format_exception(type(e), e, e.__traceback__))))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.parametrize("fail_rank,skewed", [(-1, False), (1, False), (-1, True)],
                         ids=["even shards", "a rank needs the whole text", "lopsided shards"])
def test_two_ranks_over_gloo(fail_rank, skewed):
    world, n, R, m = 2, 60000, 501, 24
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, R, m, fail_rank, q, skewed))
             for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(results) == [(0, "ok"), (1, "ok")], results
