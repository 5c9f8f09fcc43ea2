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
        np.testing.assert_array_equal(ix.export()[0], fo.bwt)
        tt = _dev(np.concatenate([oracle.gen_tandem(n, 7, 11), np.zeros(64, np.uint8)]))
        ix2, how2 = gpu.build_index_sharded(tt, n)
        assert how2 == "whole"
    finally:
        dist.destroy_process_group()
