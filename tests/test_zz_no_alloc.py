"""bwtgpu_search_static (the entry point that allocates nothing) against
bwtgpu_search through the C-ABI: same intervals and locate rows, bit for bit,
on two calls in a row, the second with a smaller batch, for reads that use the
static scratch (too long to be packed inside the kernel) and for reads that do
not; batches beyond the scratch are refused with a clear error."""
import numpy as np
import pytest
import torch

from bwt_amd import synth

pytestmark = pytest.mark.gpu


def _reads(oracle, text, n, R, m, seed):
    reads, _ = oracle.gen_reads(text, n, R, m, seed)
    flat = np.concatenate([reads.reshape(-1), np.zeros(64, np.uint8)])
    return torch.from_numpy(flat).cuda()


def _same(gpu, a, b):
    for x, y in zip(a, b):
        np.testing.assert_array_equal(gpu.to_numpy_u32(x), gpu.to_numpy_u32(y))


@pytest.mark.parametrize("built_from_text", [False, True])
def test_static_search_equals_search_twice_in_a_row(oracle, gpu, built_from_text):
    n = 200_003
    text = oracle.gen_genome(n, synth.GENOME_SEED)
    fo = oracle.make_fmi(text, n)
    ix = gpu.build_index(text, n) if built_from_text else gpu.index_from_parts(fo.bwt, n, fo.endloc, fo.idxs, lean=True)
    # m = 2000, 3000: the packed reads go through the scratch; m = 50, 600: packed inside the kernel
    # (600-base reads in parts: the bytes of 8 reads at a time pass through LDS)
    for m, R_big, R_small in ((2000, 3000, 701), (50, 20000, 4099), (3000, 513, 64), (600, 3000, 701)):
        big = _reads(oracle, text, n, R_big, m, synth.READS_SEED + m)
        small = _reads(oracle, text, n, R_small, m, synth.READS_SEED + m + 1)
        want_big = ix.search(big, m, R_big, want_locrow=True)
        want_small = ix.search(small, m, R_small, want_locrow=True)
        got_big = ix.search_static(big, m, R_big, want_locrow=True)
        got_small = ix.search_static(small, m, R_small, want_locrow=True)  # leftovers of the first call must not matter
        torch.cuda.synchronize()
        _same(gpu, got_big, want_big)
        _same(gpu, got_small, want_small)
        # and against the oracle's loc_search on the small batch
        host = small.cpu().numpy()[: R_small * m].reshape(R_small, m)
        sp_o, ep_o = oracle.search_batch(fo, host, m, m, R_small)
        np.testing.assert_array_equal(gpu.to_numpy_u32(got_small[0]), sp_o)
        np.testing.assert_array_equal(gpu.to_numpy_u32(got_small[1]), ep_o)
    ix.close()


def test_static_search_refuses_batches_beyond_its_scratch(oracle, gpu):
    n = 50_000
    text = oracle.gen_genome(n, synth.GENOME_SEED)
    ix = gpu.build_index(text, n)
    m, R = 3000, 30_000  # 188 words a read: 22.6 MB of packed reads > 16 MiB
    m2, R2 = 40, 3_000_000
    d = torch.zeros(max(R * m, R2 * m2) + 64, dtype=torch.uint8, device="cuda")
    with pytest.raises(gpu.BwtGpuError, match="bytes of scratch"):
        ix.search_static(d, m, R)
    # the same reads in two halves... still too many; a tenth fits
    sp, ep = ix.search_static(d, m, R // 10)
    sp2, ep2 = ix.search(d, m, R // 10)
    _same(gpu, (sp, ep), (sp2, ep2))
    # short reads are not limited: they never touch the scratch
    sp, ep = ix.search_static(d, m2, R2)
    sp2, ep2 = ix.search(d, m2, R2)
    _same(gpu, (sp, ep), (sp2, ep2))
    ix.close()
