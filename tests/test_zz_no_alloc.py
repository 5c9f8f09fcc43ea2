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


@pytest.mark.parametrize("form", ["built", "loaded"])
def test_static_search_on_an_index_no_search_has_prepared(oracle, gpu, form):
    """The first thing done with the index is the call that allocates nothing: it searches on
    the rank blocks (no position records, no k-mer table), and locate_rows walks LF or reads
    the suffix array the build left; after an ordinary search has made the search-only arrays
    the same call gives the same answers from them."""
    n = 300_007
    text = oracle.gen_genome(n, synth.GENOME_SEED + 5)
    fo = oracle.make_fmi(text, n)
    ix = gpu.build_index(text, n) if form == "built" else gpu.index_from_parts(fo.bwt, n, fo.endloc, fo.idxs)
    for m, R in ((50, 5000), (2000, 300)):
        d = _reads(oracle, text, n, R, m, synth.READS_SEED + 7 * m)
        host = d.cpu().numpy()[: R * m].reshape(R, m)
        sp_o, ep_o = oracle.search_batch(fo, host, m, m, R)
        for attempt in range(2):
            if attempt == 1:
                ix.search(d, m, R)  # makes the arrays
            sp, ep, lr = ix.search_static(d, m, R, want_locrow=True)
            np.testing.assert_array_equal(gpu.to_numpy_u32(sp), sp_o)
            np.testing.assert_array_equal(gpu.to_numpy_u32(ep), ep_o)
            rows = gpu.to_numpy_u32(lr)
            ok = rows <= n
            pos = gpu.to_numpy_u32(ix.locate_rows(lr))
            want, _ = oracle.locate_rows(fo, np.where(ok, rows, 0).astype(np.uint32))
            np.testing.assert_array_equal(pos[ok], want[ok])
        if m == 50:  # back to an index nobody has searched, for the long reads
            ix.close()
            ix = gpu.build_index(text, n) if form == "built" else gpu.index_from_parts(fo.bwt, n, fo.endloc, fo.idxs)
    ix.close()


def test_static_search_of_reads_at_an_odd_address(oracle, gpu):
    """Patterns that start 3 bytes into a buffer: the staging of the kernel that packs its own
    reads and the separate pack kernel both have to deal with the misalignment."""
    n = 120_001
    text = oracle.gen_genome(n, synth.GENOME_SEED + 6)
    fo = oracle.make_fmi(text, n)
    ix = gpu.build_index(text, n)
    for m, R in ((50, 4097), (37, 1000), (2000, 257), (600, 900)):
        d = _reads(oracle, text, n, R, m, synth.READS_SEED + m)
        shifted = torch.cat([torch.full((3,), 2, dtype=torch.uint8, device="cuda"), d])[3:]
        assert shifted.data_ptr() % 16 == 3
        host = d.cpu().numpy()[: R * m].reshape(R, m)
        sp_o, ep_o = oracle.search_batch(fo, host, m, m, R)
        for fn in (ix.search_static, ix.search):
            sp, ep = fn(shifted, m, R)
            np.testing.assert_array_equal(gpu.to_numpy_u32(sp), sp_o)
            np.testing.assert_array_equal(gpu.to_numpy_u32(ep), ep_o)
    ix.close()


@pytest.mark.parametrize("form", ["built", "loaded"])
def test_static_search_beside_another_threads_first_search(oracle, gpu, form):
    """One thread's first search makes the search-only arrays (for a loaded index: suffix array and
    text first) while another keeps calling the entry points that do not wait for them; every answer
    of the latter must be the right one, whichever arrays it found."""
    import threading

    n = 6_000_011
    text = oracle.gen_genome(n, synth.GENOME_SEED + 8)
    ref_ix = gpu.build_index(text, n)
    bwt, idxs = ref_ix.export()
    m, R = 40, 20000
    d = _reads(oracle, text, n, R, m, synth.READS_SEED + 3)
    want = [gpu.to_numpy_u32(x) for x in ref_ix.search(d, m, R, want_locrow=True)]
    want_pos = gpu.to_numpy_u32(ref_ix.locate_rows(ref_ix.search(d, m, R, want_locrow=True)[2]))
    found = want[2] <= n
    for _ in range(3):
        ix = gpu.build_index(text, n) if form == "built" else gpu.index_from_parts(bwt, n, ref_ix.endloc, idxs)
        errors = []

        def first_search():
            try:
                torch.cuda.set_device(0)
                got = ix.search(d, m, R, want_locrow=True)
                torch.cuda.synchronize()
                for g, w in zip(got, want):
                    np.testing.assert_array_equal(gpu.to_numpy_u32(g), w)
            except Exception as e:  # noqa: BLE001 - reported by the main thread
                errors.append(e)

        t = threading.Thread(target=first_search)
        t.start()
        rounds = 0
        while t.is_alive() or rounds < 3:
            got = ix.search_static(d, m, R, want_locrow=True)
            pos = ix.locate_rows(got[2])
            torch.cuda.synchronize()
            for g, w in zip(got, want):
                np.testing.assert_array_equal(gpu.to_numpy_u32(g), w)
            np.testing.assert_array_equal(gpu.to_numpy_u32(pos)[found], want_pos[found])
            rounds += 1
        t.join()
        assert not errors, errors
        ix.close()
    ref_ix.close()
