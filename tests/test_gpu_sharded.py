"""Sharded suffix-array build on one device: the shards of 1..8 logical ranks,
built one after the other, concatenate to the whole array (bit-exact against
the oracle), and the index assembled from them equals make_fmi's."""
import numpy as np
import pytest
import torch

from bwt_amd import dist as D
from bwt_amd import synth

pytestmark = pytest.mark.gpu


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_bucket_counts_match_the_text(oracle, gpu):
    for n in (5, 4097, 100003):
        text = oracle.gen_genome(n, 31 + n)
        bases = np.concatenate([synth.unpack(text, n), np.zeros(3, np.uint8)])
        digit = (bases[:-3].astype(np.int64) << 6) | (bases[1:-2] << 4) | (bases[2:-1] << 2) | bases[3:]
        want = np.bincount(digit[:n], minlength=256).astype(np.uint64)
        t = _dev(np.concatenate([text, np.zeros(64, np.uint8)]))
        np.testing.assert_array_equal(gpu.sa_bucket_counts(t, n), want)


@pytest.mark.parametrize("n,world", [(100003, 2), (1000000, 3), (1000000, 8), (5000, 4)])
def test_shards_concatenate_to_the_suffix_array(oracle, gpu, n, world):
    text = oracle.gen_genome(n, synth.GENOME_SEED + world)
    sa_o = oracle.histsort(text, n)
    fo = oracle.make_fmi(text, n, sa_o)
    t = _dev(np.concatenate([text, np.zeros(64, np.uint8)]))
    counts = gpu.sa_bucket_counts(t, n)
    assert int(counts.sum()) == n
    ranges = D.plan_bucket_ranges(counts, world)
    rows_of = D.shard_rows(counts, ranges)
    assert ranges[0][0] == 0 and ranges[-1][1] == 256
    sa = torch.full((n + 1,), -1, dtype=torch.int32, device="cuda")
    bw = torch.zeros(n + 1, dtype=torch.uint8, device="cuda")
    sa[0] = n
    endloc = 0
    for (lo, hi), (begin, rows) in zip(ranges, rows_of):
        if rows == 0:
            continue
        b, r, e = gpu.sa_build_range(t, n, lo, hi, sa[begin:begin + rows], bw[begin:begin + rows])
        assert (b, r) == (begin, rows)
        endloc += e
    np.testing.assert_array_equal(gpu.to_numpy_u32(sa), sa_o)
    assert endloc == fo.endloc
    if world > 1 and n > 50000:  # balanced on a random genome
        sizes = [r for _, r in rows_of]
        assert max(sizes) < 1.3 * n / world
    ix = gpu.build_index_from_sa(t, n, sa, bw[1:], endloc)
    assert ix.endloc == fo.endloc and ix.C == fo.C
    bwt, idxs = ix.export()
    np.testing.assert_array_equal(bwt, fo.bwt)
    np.testing.assert_array_equal(idxs, fo.idxs)
    np.testing.assert_array_equal(ix.rank_rows16(), fo.rank_rows)
    # and without the carried BWT symbols
    ix2 = gpu.build_index_from_sa(t, n, sa)
    assert ix2.endloc == fo.endloc
    np.testing.assert_array_equal(ix2.export()[0], fo.bwt)


@pytest.mark.parametrize("n,world", [(100003, 2), (1000000, 3), (1000000, 8), (300000, 64), (5000, 4)])
def test_index_from_every_shards_own_products(oracle, gpu, n, world):
    """What the ranks of a sharded build exchange: each shard's BWT symbols packed
    to 2 bits and every 8th of its rows.  The index assembled from those equals
    make_fmi's (BWT, samples, rank rows, C, the $ row); with the rows attached
    afterwards it searches like an index built whole."""
    text = oracle.gen_genome(n, synth.GENOME_SEED + 7 * world)
    sa_o = oracle.histsort(text, n)
    fo = oracle.make_fmi(text, n, sa_o)
    t = _dev(np.concatenate([text, np.zeros(64, np.uint8)]))
    counts = gpu.sa_bucket_counts(t, n)
    ranges = D.plan_bucket_ranges(counts, world)
    rows_of = D.shard_rows(counts, ranges)
    packed, offsets, at = [], [], 0
    samples = torch.full((n // 8 + 1,), -1, dtype=torch.int32, device="cuda")
    samples[0] = n
    sa = torch.full((n + 1,), -1, dtype=torch.int32, device="cuda")
    sa[0] = n
    endloc = 0
    for (lo, hi), (begin, rows) in zip(ranges, rows_of):
        offsets.append(at)
        if rows == 0:
            continue
        rows_buf = torch.empty(rows, dtype=torch.int32, device="cuda")
        bw_buf = torch.empty(rows, dtype=torch.uint8, device="cuda")
        b, r, e = gpu.sa_build_range(t, n, lo, hi, rows_buf, bw_buf)
        assert (b, r) == (begin, rows)
        endloc += e
        packed.append(gpu.pack_bw2(bw_buf, rows))
        at += (rows + 3) // 4
        first, i0, cnt = D.sample_span(begin, rows)
        if cnt:
            samples[i0:i0 + cnt] = rows_buf[first::8]
        sa[begin:begin + rows] = rows_buf
    parts = D.ShardedParts(endloc, rows_of, torch.cat(packed + [torch.zeros(16, dtype=torch.uint8, device="cuda")]),
                           offsets, samples)
    ix = gpu.build_index_from_shards(t, n, parts)
    assert ix.endloc == fo.endloc and ix.C == fo.C
    bwt, idxs = ix.export()
    np.testing.assert_array_equal(bwt, fo.bwt)
    np.testing.assert_array_equal(idxs, fo.idxs)
    np.testing.assert_array_equal(ix.rank_rows16(), fo.rank_rows)
    R, m = 3000, 40
    reads, _ = oracle.gen_reads(text, n, R, m, 17)
    sp_o, ep_o = oracle.search_batch(fo, reads, m, m, R)
    d = _dev(reads.reshape(-1))
    small = ix.device_bytes
    for attached in (False, True):
        if attached:
            gpu.check(gpu._lib.load().bwtgpu_index_attach_sa(ix.h, t.data_ptr(), sa.data_ptr(), None))
            torch.cuda.synchronize()
        sp, ep, lr = ix.search(d, m, R, want_locrow=True)
        np.testing.assert_array_equal(gpu.to_numpy_u32(sp), sp_o)
        np.testing.assert_array_equal(gpu.to_numpy_u32(ep), ep_o)
        pos = gpu.to_numpy_u32(ix.locate_rows(lr))
        rows = gpu.to_numpy_u32(lr)
        ok = rows <= n
        np.testing.assert_array_equal(pos[ok], sa_o[rows[ok]])
    assert ix.device_bytes > small + 8 * n  # rows, text words, position records
    rep = ix.certify(t)
    assert all(v == 0 for v in rep.values()), rep
    ix.close()


def test_index_from_shards_rejects_inconsistent_tables(oracle, gpu):
    n = 5000
    text = oracle.gen_genome(n, 3)
    t = _dev(np.concatenate([text, np.zeros(64, np.uint8)]))
    z8 = torch.zeros(n, dtype=torch.uint8, device="cuda")
    z32 = torch.zeros(n // 8 + 1, dtype=torch.int32, device="cuda")
    for rows_of, endloc in (([(1, n - 1)], 5), ([(2, n)], 5), ([(1, 100), (102, n - 100)], 5), ([(1, n)], 0),
                            ([(1, n)], n + 1)):
        with pytest.raises(gpu.BwtGpuError):
            gpu.build_index_from_shards(t, n, D.ShardedParts(endloc, rows_of, z8, [0] * len(rows_of), z32))


def test_shard_with_long_repeats_asks_for_a_whole_build(oracle, gpu):
    n = 200000
    text = oracle.gen_tandem(n, 7, 11)
    t = _dev(np.concatenate([text, np.zeros(64, np.uint8)]))
    rows = torch.empty(n + 1, dtype=torch.int32, device="cuda")
    with pytest.raises(gpu.BwtGpuError) as ei:
        gpu.sa_build_range(t, n, 0, 256, rows)
    assert ei.value.code == -6
    # the same text builds whole
    np.testing.assert_array_equal(gpu.to_numpy_u32(gpu.sa_build(t, n)), oracle.sa_doubling(text, n))


def test_range_build_rejects_bad_ranges(gpu):
    n = 10000
    t = gpu.gen_genome(n, 5)
    rows = torch.empty(16, dtype=torch.int32, device="cuda")
    for lo, hi in ((5, 5), (9, 3), (0, 257)):
        with pytest.raises(gpu.BwtGpuError):
            gpu.sa_build_range(t, n, lo, hi, rows)
    with pytest.raises(gpu.BwtGpuError):  # room for 16 rows only
        gpu.sa_build_range(t, n, 0, 128, rows)


def test_single_rank_group_builds_sharded(oracle, gpu):
    """The multi-rank driver over RCCL with a world of one rank."""
    import torch.distributed as dist
    if not dist.is_initialized():
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29631", rank=0, world_size=1,
                                device_id=torch.device("cuda", 0))
    try:
        n = 300000
        text = oracle.gen_genome(n, 99)
        fo = oracle.make_fmi(text, n)
        t = _dev(np.concatenate([text, np.zeros(64, np.uint8)]))
        ix, how = gpu.build_index_sharded(t, n)
        assert how == "sharded"
        bwt, idxs = ix.export()
        np.testing.assert_array_equal(bwt, fo.bwt)
        np.testing.assert_array_equal(idxs, fo.idxs)
        rep = ix.certify(t)  # the rows were gathered and attached
        assert all(v == 0 for v in rep.values()), rep
        ix3, _ = gpu.build_index_sharded(t, n, want_sa=False)
        np.testing.assert_array_equal(ix3.export()[0], fo.bwt)
        with pytest.raises(gpu.BwtGpuError, match="does not keep"):
            ix3.certify(t)
        tt = _dev(np.concatenate([oracle.gen_tandem(n, 7, 11), np.zeros(64, np.uint8)]))
        ix2, how2 = gpu.build_index_sharded(tt, n)
        assert how2 == "whole"
    finally:
        dist.destroy_process_group()
