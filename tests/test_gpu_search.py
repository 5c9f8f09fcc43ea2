"""HP2 parity on the GPU: rank blocks, backward search and locate through the
C-ABI against the CPU oracle on the same seeded inputs (bit-exact)."""
import numpy as np
import pytest
import torch

from bwt_amd import synth

pytestmark = pytest.mark.gpu


def _index_pair(oracle, gpu, n, seed=synth.GENOME_SEED, text=None):
    if text is None:
        text = oracle.gen_genome(n, seed)
    fo = oracle.make_fmi(text, n)
    ix = gpu.index_from_parts(fo.bwt, n, fo.endloc, fo.idxs, lean=True)
    return text, fo, ix


@pytest.mark.parametrize("n", [1, 5, 13, 63, 64, 65, 777, 4096, 100003, 1000000])
def test_rank_blocks_match_seq_index(oracle, gpu, n):
    """Device rank blocks (from stored BWT, as read_index rebuilds them) give
    the reference's rank rows (seqindex.c:23-87) and C[] (seqindex.c:183-187)."""
    _, fo, ix = _index_pair(oracle, gpu, n)
    assert ix.n == n and ix.endloc == fo.endloc
    assert ix.C == fo.C
    np.testing.assert_array_equal(ix.rank_rows16(), fo.rank_rows)
    bwt, idxs = ix.export()
    np.testing.assert_array_equal(bwt, fo.bwt)
    np.testing.assert_array_equal(idxs, fo.idxs)


@pytest.mark.parametrize("n,R,m", [(777, 500, 12), (100003, 5000, 20), (1000000, 10000, 20),
                                   (1000000, 20000, 50), (1000000, 3000, 100), (1000000, 1000, 17),
                                   (1000000, 3000, 150), (1000000, 2001, 251), (1000000, 700, 600)])
def test_search_matches_loc_search(oracle, gpu, n, R, m):
    text, fo, ix = _index_pair(oracle, gpu, n)
    reads, offs = oracle.gen_reads(text, n, R, m, synth.READS_SEED)
    sp_o, ep_o = oracle.search_batch(fo, reads, m, m, R)
    d = torch.from_numpy(reads.reshape(-1)).cuda()
    sp, ep, lr = ix.search(d, m, R, want_locrow=True)
    sp, ep, lr = (gpu.to_numpy_u32(x) for x in (sp, ep, lr))
    np.testing.assert_array_equal(sp, sp_o)
    np.testing.assert_array_equal(ep, ep_o)
    # locate(): unc_sa(sp) unless the interval emptied before the last step
    want = np.array([oracle.locate(fo, reads[r]) for r in range(min(R, 2000))], np.int64)
    pos = gpu.to_numpy_u32(ix.locate_rows(torch.from_numpy(lr.view(np.int32)).cuda()))
    got = pos[: len(want)].astype(np.int64)
    got[got == 0xFFFFFFFF] = -1
    np.testing.assert_array_equal(got, want)
    # reads sampled from the text locate to their own offset when unique
    uniq = (ep - sp == 1) & (offs != 0xFFFFFFFF)
    assert uniq.sum() > 0.9 * R * 0.5
    np.testing.assert_array_equal(pos[uniq], offs[uniq])


def test_search_overlapping_stride1(oracle, gpu):
    """fmitest's workload shape: overlapping 12-mers pats+i (fmitest.c:60-63)."""
    n, R, m = 100003, 20000, 12
    text, fo, ix = _index_pair(oracle, gpu, n)
    pats = (synth.rnd(7, np.arange(R + m, dtype=np.uint64)) & np.uint64(3)).astype(np.uint8)
    sp_o, ep_o = oracle.search_batch(fo, pats, m, 1, R)
    sp, ep = ix.search(torch.from_numpy(pats).cuda(), m, R, stride=1)
    np.testing.assert_array_equal(gpu.to_numpy_u32(sp), sp_o)
    np.testing.assert_array_equal(gpu.to_numpy_u32(ep), ep_o)
    # reverse_search's return value (seqindex.c:228-242) follows from the interval
    for r in range(0, R, 997):
        rs = oracle.reverse_search(fo, pats[r:r + m])
        sp_r, ep_r = int(sp_o[r]), int(ep_o[r])
        assert rs in (0, ep_r - sp_r + 1)


def test_search_wide_stride_and_unaligned_base(oracle, gpu):
    n, R, m, stride = 100003, 3000, 33, 301
    text, fo, ix = _index_pair(oracle, gpu, n)
    reads, _ = oracle.gen_reads(text, n, R, m, 99)
    buf = np.full(3 + R * stride + m, 7, np.uint8)  # junk between reads
    for r in range(R):
        buf[3 + r * stride: 3 + r * stride + m] = reads[r]
    sp_o, ep_o = oracle.search_batch(fo, reads, m, m, R)
    d = torch.from_numpy(buf).cuda()
    sp, ep = ix.search(d[3:], m, R, stride=stride)
    np.testing.assert_array_equal(gpu.to_numpy_u32(sp), sp_o)
    np.testing.assert_array_equal(gpu.to_numpy_u32(ep), ep_o)
    # narrow stride, unaligned base: LDS staging path
    buf2 = np.zeros(5 + R * m, np.uint8)
    buf2[5:] = reads.reshape(-1)
    sp, ep = ix.search(torch.from_numpy(buf2).cuda()[5:], m, R)
    np.testing.assert_array_equal(gpu.to_numpy_u32(sp), sp_o)
    np.testing.assert_array_equal(gpu.to_numpy_u32(ep), ep_o)


@pytest.mark.parametrize("n", [13, 777, 100003, 1000000])
def test_locate_every_row_is_the_suffix_array(oracle, gpu, n):
    """unc_sa(i) == SA[i] for every row (seqindex.c:244-255)."""
    text = oracle.gen_genome(n, synth.GENOME_SEED)
    sa = oracle.histsort(text, n)
    fo = oracle.make_fmi(text, n, sa)
    ix = gpu.index_from_parts(fo.bwt, n, fo.endloc, fo.idxs, lean=True)
    rows = torch.arange(0, n + 1, dtype=torch.int64, device="cuda").to(torch.int32)
    pos, steps = ix.locate_rows(rows, count_steps=True)
    np.testing.assert_array_equal(gpu.to_numpy_u32(pos), sa)
    # HBM keeps every 8th row, the file every 32nd: same answers, shorter walks
    _, steps_o = oracle.locate_rows(fo, np.arange(n + 1, dtype=np.uint32))
    assert steps <= steps_o
    if n > 1000:
        assert 5.0 < steps / (n + 1) < 9.0


def test_locate_skips_invalid_rows(oracle, gpu):
    n = 4096
    _, fo, ix = _index_pair(oracle, gpu, n)
    rows = np.array([0, 0xFFFFFFFF, 5, n, n + 1, 0xFFFFFFFE, 77], np.uint32)
    pos = gpu.to_numpy_u32(ix.locate_rows(torch.from_numpy(rows.view(np.int32)).cuda()))
    for r, p in zip(rows, pos):
        if r > n:
            assert p == 0xFFFFFFFF
        else:
            assert p == oracle.unc_sa(fo, int(r))


def test_tandem_repeat_index(oracle, gpu):
    """Highly repetitive text: wide SA intervals, long equal runs in the BWT."""
    n = 200000
    text = oracle.gen_tandem(n, 7, 11)
    sa = oracle.sa_doubling(text, n)
    fo = oracle.make_fmi(text, n, sa)
    ix = gpu.index_from_parts(fo.bwt, n, fo.endloc, fo.idxs, lean=True)
    np.testing.assert_array_equal(ix.rank_rows16(), fo.rank_rows)
    reads, _ = oracle.gen_reads(text, n, 4000, 30, 5)
    sp_o, ep_o = oracle.search_batch(fo, reads, 30, 30, 4000)
    sp, ep = ix.search(torch.from_numpy(reads.reshape(-1)).cuda(), 30, 4000)
    np.testing.assert_array_equal(gpu.to_numpy_u32(sp), sp_o)
    np.testing.assert_array_equal(gpu.to_numpy_u32(ep), ep_o)
    rows = torch.arange(0, n + 1, 3, dtype=torch.int64, device="cuda").to(torch.int32)
    pos = gpu.to_numpy_u32(ix.locate_rows(rows))
    np.testing.assert_array_equal(pos, sa[::3])


def test_device_generators_match_host(oracle, gpu):
    n, R, m = 100003, 5000, 50
    t = gpu.gen_genome(n, synth.GENOME_SEED)
    host = oracle.gen_genome(n, synth.GENOME_SEED)
    np.testing.assert_array_equal(t[: n // 4 + 1].cpu().numpy(), host)
    np.testing.assert_array_equal(synth.gen_genome(n), host)
    reads, offs = gpu.gen_reads(t, n, R, m, synth.READS_SEED)
    r_o, o_o = oracle.gen_reads(host, n, R, m, synth.READS_SEED)
    np.testing.assert_array_equal(reads[: R * m].cpu().numpy().reshape(R, m), r_o)
    np.testing.assert_array_equal(gpu.to_numpy_u32(offs), o_o)


def _mutated_reads(text_bases, n, R, m, seed):
    """Reads cut from the text, then altered so that every branch of a search
    is taken: exact copies, one substitution at a random base, reads hanging
    over the start of the text, reads hanging over its end, random reads."""
    rng = np.random.default_rng(seed)
    reads = np.empty((R, m), np.uint8)
    for r in range(R):
        kind = r % 5
        if kind == 3 and n >= 2:      # last bases are text[0:k], the bases in front are not text
            k = int(rng.integers(1, min(m, n) + 1))
            reads[r, : m - k] = rng.integers(0, 4, m - k)
            reads[r, m - k:] = text_bases[:k]
        elif kind == 4 or n < m:      # random, mostly absent
            reads[r] = rng.integers(0, 4, m)
        else:
            j = int(rng.integers(0, n - m + 1))
            reads[r] = text_bases[j: j + m]
            if kind == 1:
                q = int(rng.integers(0, m))
                reads[r, q] = (reads[r, q] + 1 + int(rng.integers(0, 3))) & 3
            elif kind == 2 and m >= 2:  # copy that runs off the end of the text
                k = int(rng.integers(1, m))
                reads[r, :k] = text_bases[n - k:]
                reads[r, k:] = rng.integers(0, 4, m - k)
    return reads


def _check_search_locate(gpu, ix, d, m, R, sp, ep, pos, msg):
    """bwtgpu_search_locate and its packed form give what the search followed by
    locate_rows gave (seqindex.c:280-309, then seqindex.c:258-277), array by array."""
    a, b, c = (gpu.to_numpy_u32(x) for x in ix.search_locate(d, m, R))
    np.testing.assert_array_equal(a, sp, err_msg=msg)
    np.testing.assert_array_equal(b, ep, err_msg=msg)
    np.testing.assert_array_equal(c, pos, err_msg=msg)
    words = gpu.pack_reads_dev(d, m, R)
    a, b, c = (gpu.to_numpy_u32(x) for x in ix.search_locate_packed(words, m, R))
    np.testing.assert_array_equal(a, sp, err_msg=msg + " packed")
    np.testing.assert_array_equal(b, ep, err_msg=msg + " packed")
    np.testing.assert_array_equal(c, pos, err_msg=msg + " packed")


@pytest.mark.parametrize("n,R,m,kind", [(13, 300, 20, "random"), (40, 600, 9, "random"), (777, 2000, 25, "random"),
                                        (100003, 20000, 50, "random"), (1000000, 20000, 100, "random"),
                                        (200000, 10000, 64, "mutated-repeat"), (50000, 5000, 33, "tandem"),
                                        (300000, 6000, 130, "random"), (300000, 4000, 260, "random"),
                                        (5000, 3000, 600, "random")],
                         ids=lambda v: str(v))
@pytest.mark.parametrize("records", ["twelve", "wide"])
def test_search_with_kept_suffix_array_equals_rank_only_search(oracle, gpu, n, R, m, kind, records, monkeypatch):
    """An index built on the device keeps SA, ISA and the text and finishes a
    search from the text once one row is left; an index loaded from its stored
    parts has only the rank blocks.  Both must give loc_search's interval
    (seqindex.c:280-309) and locate's row (seqindex.c:258-277) for every read.
    records: the position records of twelve positions and 64 bases (what a text of
    this size gets) or the wide ones of eight positions and 128 bases (texts of
    2^27 bases and more)."""
    monkeypatch.setenv("BWT_MI355X_WIDE_RECORDS", "1" if records == "wide" else "0")
    if kind == "random":
        text = oracle.gen_genome(n, synth.GENOME_SEED + n)
    elif kind == "tandem":
        text = oracle.gen_tandem(n, 7, 11)
    else:
        text = oracle.gen_tandem(n, 1000, 11, 50, 3)
    bases = synth.unpack(text, n)
    fo = oracle.make_fmi(text, n, oracle.sa_doubling(text, n) if kind != "random" else None)
    full = gpu.build_index(text, n)
    lean = gpu.index_from_parts(fo.bwt, n, fo.endloc, fo.idxs, lean=True)
    full.prepare_search()  # the position records and the k-mer table are made on first search otherwise
    lean.prepare_search()
    assert full.device_bytes > lean.device_bytes + 8 * n
    reads = _mutated_reads(bases, n, R, m, 1234 + n)
    sp_o, ep_o = oracle.search_batch(fo, reads, m, m, R)
    d = torch.from_numpy(reads.reshape(-1)).cuda()
    out = {}
    for name, ix in (("full", full), ("lean", lean)):
        sp, ep, lr = (gpu.to_numpy_u32(x) for x in ix.search(d, m, R, want_locrow=True))
        np.testing.assert_array_equal(sp, sp_o, err_msg=name)
        np.testing.assert_array_equal(ep, ep_o, err_msg=name)
        pos = gpu.to_numpy_u32(ix.locate_rows(torch.from_numpy(lr.view(np.int32)).cuda()))
        out[name] = (lr, pos)
        _check_search_locate(gpu, ix, d, m, R, sp, ep, pos, name)
    np.testing.assert_array_equal(out["full"][0], out["lean"][0])
    np.testing.assert_array_equal(out["full"][1], out["lean"][1])
    want = np.array([oracle.locate(fo, reads[r]) for r in range(min(R, 3000))], np.int64)
    got = out["full"][1][: len(want)].astype(np.int64)
    got[got == 0xFFFFFFFF] = -1
    np.testing.assert_array_equal(got, want)
    if n >= m:
        assert (ep_o > sp_o).sum() > R // 8 and (ep_o <= sp_o).sum() > R // 8


def test_compact_drops_the_kept_arrays_and_counting_search_agrees(oracle, gpu):
    n, R, m = 300000, 8000, 40
    text = oracle.gen_genome(n, 77)
    fo = oracle.make_fmi(text, n)
    ix = gpu.build_index(text, n)
    reads = _mutated_reads(synth.unpack(text, n), n, R, m, 5)
    sp_o, ep_o = oracle.search_batch(fo, reads, m, m, R)
    d = torch.from_numpy(reads.reshape(-1)).cuda()
    ix.prepare_search()
    big = ix.device_bytes
    loads_full, sp, ep = ix.search_loads(d, m, R)
    np.testing.assert_array_equal(gpu.to_numpy_u32(sp), sp_o)
    np.testing.assert_array_equal(gpu.to_numpy_u32(ep), ep_o)
    ix.compact()
    assert ix.device_bytes < big - 8 * n
    loads_lean, sp, ep = ix.search_loads(d, m, R)
    np.testing.assert_array_equal(gpu.to_numpy_u32(sp), sp_o)
    np.testing.assert_array_equal(gpu.to_numpy_u32(ep), ep_o)
    assert loads_lean["text_windows_16B"] == 0 and loads_lean["sa_isa_entries_4B"] == 0
    assert loads_full["text_windows_16B"] > 0
    assert loads_full["rank_blocks_16B"] < loads_lean["rank_blocks_16B"] <= 2 * (m - 1) * R
    rows = torch.arange(0, n + 1, 7, dtype=torch.int64, device="cuda").to(torch.int32)
    np.testing.assert_array_equal(gpu.to_numpy_u32(ix.locate_rows(rows)), oracle.histsort(text, n)[::7])


@pytest.mark.parametrize("k", [0, 2, 5, 9, 11])
def test_kmer_table_start_gives_the_same_intervals(oracle, gpu, k, monkeypatch):
    """Searches start from the table entry of the read's last k bases; reads
    shorter than k, reads of exactly k bases and absent strings included."""
    n = 300000
    text = oracle.gen_genome(n, 4242)
    fo = oracle.make_fmi(text, n)
    monkeypatch.setenv("BWT_MI355X_KMER", str(k))
    # entries of two-word tables hold the interval's size in up to six spare bits; none are spare
    # for texts of 2^30 bases and more, which is forced here for two of the table sizes
    if k in (5, 11):
        monkeypatch.setenv("BWT_MI355X_KMER_SIZEBITS", "0" if k == 5 else "2")
    full = gpu.build_index(text, n)
    lean = gpu.index_from_parts(fo.bwt, n, fo.endloc, fo.idxs, lean=True)
    full.prepare_search()  # the table is made on first search: make it while the size is forced
    lean.prepare_search()
    monkeypatch.delenv("BWT_MI355X_KMER")
    monkeypatch.delenv("BWT_MI355X_KMER_SIZEBITS", raising=False)
    assert full.kmer_table == ((k, 2) if k >= 2 else (0, 0)) and lean.kmer_table == ((k, 1) if k >= 2 else (0, 0))
    bases = synth.unpack(text, n)
    for m in (1, 4, 8, 9, 10, 11, 12, 16, 17, 23, 32, 33, 70):
        R = 3000
        reads = _mutated_reads(bases, n, R, m, 100 + m)
        sp_o, ep_o = oracle.search_batch(fo, reads, m, m, R)
        d = torch.from_numpy(reads.reshape(-1)).cuda()
        for name, ix in (("full", full), ("lean", lean)):
            sp, ep, lr = (gpu.to_numpy_u32(x) for x in ix.search(d, m, R, want_locrow=True))
            np.testing.assert_array_equal(sp, sp_o, err_msg=f"{name} m={m}")
            np.testing.assert_array_equal(ep, ep_o, err_msg=f"{name} m={m}")
            want = np.array([oracle.locate(fo, reads[r]) for r in range(300)], np.int64)
            pos = gpu.to_numpy_u32(ix.locate_rows(torch.from_numpy(lr.view(np.int32)).cuda()))
            _check_search_locate(gpu, ix, d, m, R, sp, ep, pos, f"{name} m={m}")
            got = pos[:300].astype(np.int64)
            got[got == 0xFFFFFFFF] = -1
            np.testing.assert_array_equal(got, want, err_msg=f"{name} m={m}")
    loads, _, _ = full.search_loads(torch.from_numpy(reads.reshape(-1)).cuda(), 70, 3000)
    assert loads["kmer_entries"] == (3000 if k >= 2 else 0)


@pytest.mark.parametrize("k,sizebits", [(7, None), (9, None), (9, "0"), (11, "2")])
def test_kmer_table_entry_whose_rows_include_the_dollar_row(oracle, gpu, k, sizebits, monkeypatch):
    """A table entry of a few rows keeps their BWT symbols, except when the row of suffix 0 is among
    them (it has no symbol): reads that end in the text's first k bases, where those occur more than
    once, and reads that end in the other strings of a few rows, against loc_search and locate."""
    n = 120000
    rng = np.random.default_rng(5 * k)
    bases = rng.integers(0, 4, n).astype(np.uint8)
    # the first k bases occur twice more, once behind each possible base
    for at, before in ((30000, 1), (70000, 2)):
        bases[at:at + k] = bases[:k]
        bases[at - 1] = before
    text = synth.pack(bases)
    fo = oracle.make_fmi(text, n)
    monkeypatch.setenv("BWT_MI355X_KMER", str(k))
    if sizebits is not None:
        monkeypatch.setenv("BWT_MI355X_KMER_SIZEBITS", sizebits)
    full = gpu.build_index(text, n)
    full.prepare_search()
    monkeypatch.delenv("BWT_MI355X_KMER")
    monkeypatch.delenv("BWT_MI355X_KMER_SIZEBITS", raising=False)
    assert full.kmer_table == (k, 2)
    m = k + 25
    reads = []
    for at in (30000, 70000):
        reads.append(bases[at + k - m: at + k].copy())          # occurs: ends in the string, one row goes on
        for c in range(4):                                       # every base in front of the string
            rd = np.concatenate([rng.integers(0, 4, m - k - 1), [c], bases[:k]]).astype(np.uint8)
            reads.append(rd)
    reads.append(np.concatenate([rng.integers(0, 4, m - k), bases[:k]]).astype(np.uint8))
    reads.append(bases[:k].copy()[-k:].repeat(1))                # the string itself is too short for m: padded below
    reads[-1] = np.concatenate([rng.integers(0, 4, m - k), reads[-1]]).astype(np.uint8)
    # and reads cut all over the text (strings of one, a few and many rows)
    for j in rng.integers(0, n - m, 3000):
        reads.append(bases[j:j + m].copy())
    reads = np.stack(reads)
    R = len(reads)
    sp_o, ep_o = oracle.search_batch(fo, reads, m, m, R)
    want = np.array([oracle.locate(fo, reads[r]) for r in range(R)], np.int64)
    d = torch.from_numpy(reads.reshape(-1)).cuda()
    sp, ep, pos = (gpu.to_numpy_u32(x) for x in full.search_locate(d, m, R))
    np.testing.assert_array_equal(sp, sp_o)
    np.testing.assert_array_equal(ep, ep_o)
    got = pos.astype(np.int64)
    got[got == 0xFFFFFFFF] = -1
    np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("tail", ["CAAA", "TTTT", "GATTACA", "AAAAAAAA", "ACGT"])
@pytest.mark.parametrize("k", [2, 3, 5, 6])
def test_kmer_table_around_the_short_suffixes(oracle, gpu, k, tail, monkeypatch):
    """The table keeps interval starts only; an interval ends where the next one
    starts except where rows of suffixes shorter than k lie between.  Every
    string of k and of k + 1 bases is searched on texts whose last bases make
    those rows fall before, between and after occupied intervals."""
    if k in (3, 6):  # entries without the interval's size (see test_kmer_table_start_gives_the_same_intervals)
        monkeypatch.setenv("BWT_MI355X_KMER_SIZEBITS", "0")
    rng = np.random.default_rng(len(tail) * 16 + k)
    for n in (len(tail), len(tail) + 1, 40, 3001):
        bases = rng.integers(0, 4, n).astype(np.uint8)
        bases[n - len(tail):] = ["ACGT".index(ch) for ch in tail]
        text = synth.pack(bases)
        fo = oracle.make_fmi(text, n)
        monkeypatch.setenv("BWT_MI355X_KMER", str(k))
        full = gpu.build_index(text, n)
        lean = gpu.index_from_parts(fo.bwt, n, fo.endloc, fo.idxs, lean=True)
        full.prepare_search()  # the table is made on first search: make it while the size is forced
        lean.prepare_search()
        monkeypatch.delenv("BWT_MI355X_KMER")
        assert full.kmer_table == (k, 2) and lean.kmer_table == (k, 1)
        for m in (k, k + 1):
            R = 4 ** m
            x = np.arange(R, dtype=np.uint32)
            reads = np.stack([(x >> (2 * (m - 1 - j))) & 3 for j in range(m)], axis=1).astype(np.uint8)
            sp_o, ep_o = oracle.search_batch(fo, reads, m, m, R)
            d = torch.from_numpy(reads.reshape(-1)).cuda()
            for name, ix in (("full", full), ("lean", lean)):
                sp, ep = (gpu.to_numpy_u32(t) for t in ix.search(d, m, R))
                np.testing.assert_array_equal(sp, sp_o, err_msg=f"{name} n={n} m={m}")
                np.testing.assert_array_equal(ep, ep_o, err_msg=f"{name} n={n} m={m}")


@pytest.mark.parametrize("levels", ["1", "2"])
@pytest.mark.parametrize("n", [5000, 100003, 3000000])
def test_two_step_inverse_suffix_array(oracle, gpu, n, levels, monkeypatch):
    """Large texts build the inverse suffix array by partition + fill; forced
    here on small ones.  Every search that ends in the text comparison reads it."""
    text = oracle.gen_genome(n, 555 + n)
    fo = oracle.make_fmi(text, n)
    monkeypatch.setenv("BWT_MI355X_ISA_TWO_STEP", levels)
    ix = gpu.build_index(text, n)
    ix.prepare_search()  # the position records are made on first search: make them while the path is forced
    monkeypatch.delenv("BWT_MI355X_ISA_TWO_STEP")
    R, m = 20000, 60
    reads = _mutated_reads(synth.unpack(text, n), n, R, m, 77)
    sp_o, ep_o = oracle.search_batch(fo, reads, m, m, R)
    sp, ep = ix.search(torch.from_numpy(reads.reshape(-1)).cuda(), m, R)
    np.testing.assert_array_equal(gpu.to_numpy_u32(sp), sp_o)
    np.testing.assert_array_equal(gpu.to_numpy_u32(ep), ep_o)


def test_attach_text_to_a_loaded_index(oracle, gpu):
    """An index loaded from its stored parts takes the text it was built from
    (checked against its BWT) and then searches like one built from the text;
    a different text is refused and changes nothing."""
    n, R, m = 500000, 20000, 70
    text = oracle.gen_genome(n, 808)
    fo = oracle.make_fmi(text, n)
    ix = gpu.index_from_parts(fo.bwt, n, fo.endloc, fo.idxs, lean=True)
    ix.prepare_search()
    lean_bytes = ix.device_bytes
    other = oracle.gen_genome(n, 809)
    with pytest.raises(gpu.BwtGpuError, match="not the one"):
        ix.attach_text(other)
    assert ix.device_bytes == lean_bytes
    ix.attach_text(text)
    ix.prepare_search()
    assert ix.device_bytes > lean_bytes + 8 * n
    ix.attach_text(text)  # a second call is a no-op
    reads = _mutated_reads(synth.unpack(text, n), n, R, m, 31)
    sp_o, ep_o = oracle.search_batch(fo, reads, m, m, R)
    d = torch.from_numpy(reads.reshape(-1)).cuda()
    loads, sp, ep = ix.search_loads(d, m, R)
    np.testing.assert_array_equal(gpu.to_numpy_u32(sp), sp_o)
    np.testing.assert_array_equal(gpu.to_numpy_u32(ep), ep_o)
    assert loads["text_windows_16B"] > 0
    rows = torch.arange(0, n + 1, 5, dtype=torch.int64, device="cuda").to(torch.int32)
    np.testing.assert_array_equal(gpu.to_numpy_u32(ix.locate_rows(rows)), oracle.histsort(text, n)[::5])
    # device-resident text
    ix2 = gpu.index_from_parts(fo.bwt, n, fo.endloc, fo.idxs, lean=True)
    ix2.attach_text(torch.from_numpy(np.concatenate([text, np.zeros(64, np.uint8)])).cuda())
    sp2, ep2 = ix2.search(d, m, R)
    np.testing.assert_array_equal(gpu.to_numpy_u32(sp2), sp_o)
