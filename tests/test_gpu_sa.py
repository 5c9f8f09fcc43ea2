"""HP1 parity on the GPU: suffix array, BWT, sampled SA, rank rows and C[]
built by the HIP kernels against the CPU oracle (bit-exact), plus the device
primitives the build stands on."""
import ctypes as C

import numpy as np
import pytest
import torch

from bwt_amd import _lib, synth

pytestmark = pytest.mark.gpu


def _dev_text(text):
    """packed text -> CUDA tensor with slack past the last byte"""
    t = torch.zeros(len(text) + 64, dtype=torch.uint8, device="cuda")
    t[: len(text)] = torch.from_numpy(np.ascontiguousarray(text))
    return t


# ---- device primitives ----

@pytest.mark.parametrize("count", [1, 63, 2048, 2049, 100000, 3000001])
def test_device_scans(gpu, count):
    rng = np.random.default_rng(count)
    a = rng.integers(0, 1000, count, dtype=np.uint32)
    d = torch.from_numpy(a.view(np.int32)).cuda()
    out = torch.empty_like(d)
    tot = torch.zeros(1, dtype=torch.int32, device="cuda")
    lib = _lib.load()
    _lib.check(lib.bwtgpu_diag_scan(0, d.data_ptr(), out.data_ptr(), count, tot.data_ptr(), None))
    want = np.concatenate([[0], np.cumsum(a, dtype=np.uint64)[:-1]]).astype(np.uint32)
    np.testing.assert_array_equal(gpu.to_numpy_u32(out), want)
    assert int(tot.item()) & 0xFFFFFFFF == int(a.sum(dtype=np.uint64)) & 0xFFFFFFFF
    # inclusive max scan, in place
    b = np.where(rng.random(count) < 0.05, rng.integers(0, 2**31, count), 0).astype(np.uint32)
    d = torch.from_numpy(b.view(np.int32)).cuda()
    _lib.check(lib.bwtgpu_diag_scan(1, d.data_ptr(), d.data_ptr(), count, None, None))
    np.testing.assert_array_equal(gpu.to_numpy_u32(d), np.maximum.accumulate(b))


@pytest.mark.parametrize("count,bits", [(1, 8), (100, 16), (4096, 24), (4097, 40), (250000, 54), (1000003, 64)])
def test_device_radix_sort_is_stable(gpu, count, bits):
    rng = np.random.default_rng(bits)
    keys = rng.integers(0, 2**63, count, dtype=np.uint64) >> np.uint64(64 - bits) if bits < 64 else \
        rng.integers(0, 2**63, count, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, count, dtype=np.uint64)
    if count > 1000:
        keys[: count // 2] = keys[count // 2: count // 2 * 2]  # plenty of duplicates
    vals = np.arange(count, dtype=np.uint32)
    dk = torch.from_numpy(keys.view(np.int64)).cuda()
    dv = torch.from_numpy(vals.view(np.int32)).cuda()
    _lib.check(_lib.load().bwtgpu_diag_radix_sort(dk.data_ptr(), dv.data_ptr(), count, bits, None))
    order = np.argsort(keys, kind="stable")
    np.testing.assert_array_equal(dk.cpu().numpy().view(np.uint64), keys[order])
    np.testing.assert_array_equal(gpu.to_numpy_u32(dv), order.astype(np.uint32))


# ---- suffix array ----

def _check_sa(oracle, gpu, text, n, want=None):
    sa = gpu.to_numpy_u32(gpu.sa_build(_dev_text(text), n))
    if want is None:
        want = oracle.histsort(text, n, threads=1)
    bad = np.nonzero(sa != want)[0]
    assert len(bad) == 0, f"n={n}: first mismatch at row {bad[0]}: {sa[bad[0]]} vs {want[bad[0]]}; stats={gpu.sa_last_stats()}"
    return sa


@pytest.mark.parametrize("n", [1, 2, 3, 13, 64, 777, 4095, 4096, 4097, 5000, 70000, 100003, 1000000, 3000017])
def test_sa_random_genome(oracle, gpu, n):
    text = oracle.gen_genome(n, synth.GENOME_SEED)
    _check_sa(oracle, gpu, text, n)


def test_sa_ignores_bits_after_the_last_base(oracle, gpu):
    n = 100003
    text = oracle.gen_genome(n, 5)
    want = oracle.histsort(text, n)
    dirty = text.copy()
    dirty[n // 4] |= 0xFF >> (2 * (n % 4))
    _check_sa(oracle, gpu, dirty, n, want)


@pytest.mark.parametrize("n,period", [(20000, 7), (200000, 7), (200000, 1000), (300001, 1), (65536, 2), (1000000, 31)])
def test_sa_tandem_repeat(oracle, gpu, n, period):
    """The reference's pathological case (README.md:76-77): histsort recurses
    to the LCP; the device build must take the prefix-doubling path."""
    text = oracle.gen_tandem(n, period, 11)
    want = oracle.sa_doubling(text, n)
    _check_sa(oracle, gpu, text, n, want)
    st = gpu.sa_last_stats()
    assert st[2] > 0 and st[3] > 0, st  # unresolved rows went through doubling rounds


@pytest.mark.parametrize("form", ["groups", "lsd"])
def test_sa_repeat_with_point_mutations(oracle, gpu, form, monkeypatch):
    """form lsd: the rounds as one radix sort of (group, key) each, kept beside the
    group-by-group rounds for comparison (BWT_MI355X_DOUBLING)"""
    if form == "lsd":
        monkeypatch.setenv("BWT_MI355X_DOUBLING", "lsd")
    n = 500000
    text = oracle.gen_tandem(n, 1000, 11, mut_every=997, mut_seed=3)
    want = oracle.sa_doubling(text, n)
    _check_sa(oracle, gpu, text, n, want)
    text = oracle.gen_tandem(3_000_000, 31, 5, mut_every=40000, mut_seed=9)  # a few groups of 10^5 rows
    _check_sa(oracle, gpu, text, 3_000_000, oracle.sa_doubling(text, 3_000_000))


@pytest.mark.parametrize("bases", ["A", "T", "AC", "T" * 34, "ACGT" * 50, "A" * 5000, "T" * 5000 + "A" * 5000,
                                   "G" * 4097],
                         ids=["A", "T", "AC", "T34", "ACGTx50", "A5000", "T5000A5000", "G4097"])
def test_sa_degenerate_texts(oracle, gpu, bases):
    code = {"A": 0, "C": 1, "G": 2, "T": 3}
    b = np.array([code[c] for c in bases], np.uint8)
    text = synth.pack(b)
    n = len(b)
    want = oracle.sa_doubling(text, n)
    assert oracle.check_sa(text, n, want) == 0
    _check_sa(oracle, gpu, text, n, want)


def test_sa_reference_compiled(reference, oracle, gpu):
    """Where the compiled reference is present: histsort and csuff_arr
    themselves (histsortcomp.c:161, csacak.c:521) against the device build."""
    n = 1000000
    text = oracle.gen_genome(n, 77)
    want = reference.histsort(text, n)
    np.testing.assert_array_equal(reference.csuff_arr(text, n), want)
    _check_sa(oracle, gpu, text, n, want)
    t2 = oracle.gen_tandem(2000000, 1000, 5)
    _check_sa(oracle, gpu, t2, 2000000, reference.csuff_arr(t2, 2000000))


# ---- BWT + index ----

@pytest.mark.parametrize("n", [1, 5, 13, 64, 777, 4096, 100003, 1000000])
def test_index_build_matches_make_fmi(oracle, gpu, n):
    text = oracle.gen_genome(n, synth.GENOME_SEED)
    fo = oracle.make_fmi(text, n)
    ix = gpu.build_index(text, n)
    assert (ix.n, ix.endloc, ix.C) == (n, fo.endloc, fo.C)
    bwt, idxs = ix.export()
    np.testing.assert_array_equal(bwt, fo.bwt)
    np.testing.assert_array_equal(idxs, fo.idxs)
    np.testing.assert_array_equal(ix.rank_rows16(), fo.rank_rows)
    # sprintcbwt's own byte format straight from an SA
    dt = _dev_text(text)
    sa = gpu.sa_build(dt, n)
    b2, e2 = gpu.bwt_from_sa(dt, sa, n)
    assert e2 == fo.endloc
    np.testing.assert_array_equal(b2.cpu().numpy(), fo.bwt)


def test_index_build_dev_then_search_and_locate(oracle, gpu):
    """End to end on device: generate, build, search, locate; every read that
    came from the text and is unique locates to its own offset."""
    n, R, m = 2000000, 50000, 50
    t = gpu.gen_genome(n, synth.GENOME_SEED)
    ix = gpu.build_index_dev(t, n)
    reads, offs = gpu.gen_reads(t, n, R, m, synth.READS_SEED)
    sp, ep, lr = ix.search(reads, m, R, want_locrow=True)
    pos = gpu.to_numpy_u32(ix.locate_rows(lr))
    sp, ep, offs = (gpu.to_numpy_u32(x) for x in (sp, ep, offs))
    uniq = (ep - sp == 1) & (offs != 0xFFFFFFFF)
    assert uniq.mean() > 0.95
    np.testing.assert_array_equal(pos[uniq], offs[uniq])
    host = oracle.gen_genome(n, synth.GENOME_SEED)
    fo = oracle.make_fmi(host, n)
    assert (ix.endloc, ix.C) == (fo.endloc, fo.C)
    np.testing.assert_array_equal(ix.export()[0], fo.bwt)
