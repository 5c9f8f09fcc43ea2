"""The drop-in boundary itself: the reference's C symbols exported by
libbwt_mi355x.so, called the way the reference's callers call them, and the
command-line drivers."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from bwt_amd import _lib
from helpers import GOLDEN, GENOME_SEED, READS_SEED, ROOT, file_sha, sha

pytestmark = pytest.mark.gpu

u8p = C.POINTER(C.c_uint8)
i32p = C.POINTER(C.c_int32)


class FM(C.Structure):  # seqindex.h:14-22
    _fields_ = [("bwt", u8p), ("idxs", i32p), ("rank_index", C.POINTER(i32p)), ("lookup", u8p),
                ("endloc", C.c_int), ("C", C.c_int * 5), ("len", C.c_int)]


@pytest.fixture(scope="module")
def capi(gpu):
    L = C.CDLL(_lib.LIB_PATH)
    fp = C.POINTER(FM)
    L.histsort.restype = C.c_void_p
    L.histsort.argtypes = [u8p, C.c_int]
    L.csuff_arr.restype = C.c_void_p
    L.csuff_arr.argtypes = [u8p, C.c_uint]
    L.sprintcbwt.restype = C.c_uint
    L.sprintcbwt.argtypes = [u8p, C.c_void_p, C.c_uint, u8p]
    L.makecbwt.argtypes = [u8p, C.c_int, u8p]
    L.makebwt.restype = C.c_void_p
    L.makebwt.argtypes = [u8p, C.c_int]
    L.make_fmi.restype = fp
    L.make_fmi.argtypes = [u8p, C.c_int]
    L.make_fmi_sacak.restype = fp
    L.make_fmi_sacak.argtypes = [u8p, C.c_int]
    L.destroy_fmi.argtypes = [fp]
    L.rank.argtypes = [fp, C.c_char, C.c_int]
    L.lf.argtypes = [fp, C.c_int]
    L.unc_sa.argtypes = [fp, C.c_int]
    L.reverse_search.argtypes = [fp, u8p, C.c_int]
    L.locate.argtypes = [fp, u8p, C.c_int]
    L.loc_search.argtypes = [fp, u8p, C.c_int, i32p, i32p]
    L.seq_index.restype = C.POINTER(i32p)
    L.seq_index.argtypes = [u8p, C.c_int, C.c_int, u8p]
    L.seq_rank.argtypes = [u8p, C.POINTER(i32p), C.c_int, C.c_int, C.c_char, u8p]
    L.lookup_table.restype = C.c_void_p
    L.fmi_search_batch.argtypes = [fp, u8p, C.c_int, C.c_long, C.c_long, i32p, i32p]
    L.fmi_reverse_search_batch.argtypes = [fp, u8p, C.c_int, C.c_long, C.c_long, i32p]
    L.fmi_locate_batch.argtypes = [fp, u8p, C.c_int, C.c_long, C.c_long, i32p]
    L.fmi_search_locate_batch.argtypes = [fp, u8p, C.c_int, C.c_long, C.c_long, i32p, i32p, i32p]
    L.fmi_unc_sa_batch.argtypes = [fp, i32p, C.c_long, i32p]
    libc = C.CDLL(None)
    libc.fopen.restype = C.c_void_p
    libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
    libc.fclose.argtypes = [C.c_void_p]
    libc.free.argtypes = [C.c_void_p]
    L.write_index.argtypes = [fp, C.c_void_p]
    L.read_index.restype = fp
    L.read_index.argtypes = [C.c_char_p, C.c_void_p]
    L.libc = libc
    return L


def _p(a, t=u8p):
    return a.ctypes.data_as(t)


def _take(L, ptr, count):
    a = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint32)), shape=(count,)).copy()
    L.libc.free(ptr)
    return a


def test_histsort_csuff_arr_sprintcbwt(capi, oracle):
    n = 100003
    text = oracle.gen_genome(n, GENOME_SEED)
    want = oracle.histsort(text, n)
    sa = _take(capi, capi.histsort(_p(text), n), n + 1)
    np.testing.assert_array_equal(sa, want)
    np.testing.assert_array_equal(_take(capi, capi.csuff_arr(_p(text), n), n + 1), want)
    bwt_o, e_o = oracle.sprintcbwt(text, want, n)
    out = np.zeros((n + 3) // 4, np.uint8)
    assert capi.sprintcbwt(_p(text), _p(sa, C.c_void_p), n, _p(out)) == e_o
    np.testing.assert_array_equal(out, bwt_o)
    out2 = np.zeros_like(out)
    assert capi.makecbwt(_p(text), n, _p(out2)) == e_o
    np.testing.assert_array_equal(out2, bwt_o)
    # makebwt: the text form, $ included, last row dropped (histsortcomp.c:272)
    t40 = GOLDEN["tiny40"]
    text40 = np.frombuffer(bytes.fromhex(t40["text_hex"]), np.uint8).copy()
    s = C.string_at(capi.makebwt(_p(text40), 40)).decode()
    sa40 = t40["sa"]
    from bwt_amd import synth
    b = synth.unpack(text40, 40)
    assert s == "".join("$" if x == 0 else str(b[x - 1]) for x in sa40[:40])


def test_make_fmi_struct_and_scalar_queries(capi, oracle):
    rec = GOLDEN["random"][3]  # n = 100003
    n = rec["n"]
    text = oracle.gen_genome(n, GENOME_SEED)
    fo = oracle.make_fmi(text, n)
    fmi = capi.make_fmi(_p(text), n)
    f = fmi.contents
    assert (f.len, f.endloc, list(f.C)) == (n, rec["endloc"], rec["C"])
    assert sha(np.ctypeslib.as_array(f.bwt, shape=((n + 3) // 4,))) == rec["bwt"]
    assert sha(np.ctypeslib.as_array(f.idxs, shape=(1 + n // 32,))) == rec["idxs"]
    rows = 1 + (n + 15) // 16
    rr = np.array([[f.rank_index[j][c] for c in range(4)] for j in range(0, rows, 97)], np.uint32)
    np.testing.assert_array_equal(rr, fo.rank_rows[::97])
    np.testing.assert_array_equal(np.ctypeslib.as_array(f.lookup, shape=(1024,)), oracle.lookup_table())
    for i in [0, 1, 31, 32, fo.endloc - 1, fo.endloc, fo.endloc + 1, n // 2, n - 1, n]:
        assert capi.lf(fmi, i) == oracle.lf(fo, i)
        assert capi.unc_sa(fmi, i) == oracle.unc_sa(fo, i)
        for c in range(4):
            assert capi.rank(fmi, bytes([c]), i) == oracle.rank(fo, c, i)
    reads, _ = oracle.gen_reads(text, n, 40, 20, READS_SEED)
    for r in range(40):
        sp, ep = C.c_int(), C.c_int()
        capi.loc_search(fmi, _p(reads[r]), 20, C.byref(sp), C.byref(ep))
        assert (sp.value, ep.value) == oracle.loc_search(fo, reads[r])
        assert capi.reverse_search(fmi, _p(reads[r]), 20) == oracle.reverse_search(fo, reads[r])
        if ep.value - sp.value == 1:
            assert capi.locate(fmi, _p(reads[r]), 20) == oracle.locate(fo, reads[r])
    # longer reads take the other small-call path (the search kernel proper), absent and mutated ones included
    long_reads, _ = oracle.gen_reads(text, n, 30, 150, READS_SEED + 1)
    long_reads[::3, 75] = (long_reads[::3, 75] + 1) & 3
    for r in range(30):
        sp, ep = C.c_int(), C.c_int()
        capi.loc_search(fmi, _p(long_reads[r]), 150, C.byref(sp), C.byref(ep))
        assert (sp.value, ep.value) == oracle.loc_search(fo, long_reads[r])
        assert capi.reverse_search(fmi, _p(long_reads[r]), 150) == oracle.reverse_search(fo, long_reads[r])
        assert capi.locate(fmi, _p(long_reads[r]), 150) == oracle.locate(fo, long_reads[r])
    # seq_index / seq_rank on a caller-owned BWT
    bwt = fo.bwt.copy()
    tbl = oracle.lookup_table()
    idx = capi.seq_index(_p(bwt), n, 16, _p(tbl))
    for i in [0, 5, 16, 17, 4095, n - 3, n]:
        for c in range(4):
            want = int(fo.rank_rows[i // 16][c]) + sum(
                1 for p in range(i // 16 * 16, i) if (bwt[p >> 2] >> (2 * (3 - (p & 3)))) & 3 == c)
            assert capi.seq_rank(_p(bwt), idx, 16, i, bytes([c]), _p(tbl)) == want
    capi.destroy_fmi(fmi)


def test_scalar_queries_from_four_threads(capi, oracle):
    """fmitest.c:67-91 calls the query functions from four threads without locks:
    scalar calls from four threads at once (ctypes drops the interpreter lock
    around each) give what the oracle gives."""
    import threading

    n = 60000
    text = oracle.gen_genome(n, GENOME_SEED + 3)
    fo = oracle.make_fmi(text, n)
    fmi = capi.make_fmi(_p(text), n)
    reads, _ = oracle.gen_reads(text, n, 4 * 1500, 14, READS_SEED + 5)
    want = [(oracle.loc_search(fo, r), oracle.reverse_search(fo, r)) for r in reads]
    got = [None] * len(reads)

    def work(t):
        for r in range(t, len(reads), 4):
            sp, ep = C.c_int(), C.c_int()
            capi.loc_search(fmi, _p(reads[r]), 14, C.byref(sp), C.byref(ep))
            got[r] = ((sp.value, ep.value), capi.reverse_search(fmi, _p(reads[r]), 14))

    threads = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert got == want
    capi.destroy_fmi(fmi)


def test_batch_forms_and_index_file(capi, oracle, tmp_path):
    rec = GOLDEN["random"][4]  # n = 10^6, 10k x 20
    n, R, m = rec["n"], rec["R"], rec["m"]
    text = oracle.gen_genome(n, GENOME_SEED)
    reads, _ = oracle.gen_reads(text, n, R, m, READS_SEED)
    fmi = capi.make_fmi_sacak(_p(text), n)
    sp, ep = np.zeros(R, np.int32), np.zeros(R, np.int32)
    assert capi.fmi_search_batch(fmi, _p(reads), m, m, R, _p(sp, i32p), _p(ep, i32p)) == 0
    assert sha(sp.view(np.uint32)) == rec["sp"] and sha(ep.view(np.uint32)) == rec["ep"]
    rs = np.zeros(R, np.int32)
    assert capi.fmi_reverse_search_batch(fmi, _p(reads), m, m, R, _p(rs, i32p)) == 0
    assert sha(rs) == rec["reverse_search"]
    pos = np.zeros(R, np.int32)
    assert capi.fmi_locate_batch(fmi, _p(reads), m, m, R, _p(pos, i32p)) == 0
    assert sha(np.where(ep - sp == 1, pos, -2).astype(np.int32)) == rec["locate_unique"]
    sp2, ep2, pos2 = (np.zeros(R, np.int32) for _ in range(3))
    assert capi.fmi_search_locate_batch(fmi, _p(reads), m, m, R, _p(sp2, i32p), _p(ep2, i32p), _p(pos2, i32p)) == 0
    assert (sp2 == sp).all() and (ep2 == ep).all() and (pos2 == pos).all()
    rows = (np.arange(rec["probe_rows"], dtype=np.int32) * rec["probe_rows_step"])
    out = np.zeros(len(rows), np.int32)
    assert capi.fmi_unc_sa_batch(fmi, _p(rows, i32p), len(rows), _p(out, i32p)) == 0
    assert sha(out.view(np.uint32)) == rec["unc_sa"]
    # write_index: the reference's file, byte for byte; read_index: back again
    path = str(tmp_path / "c.idx")
    fpw = capi.libc.fopen(path.encode(), b"wb")
    capi.write_index(fmi, fpw)
    capi.libc.fclose(fpw)
    assert file_sha(path) == (rec["index_file"], rec["index_file_bytes"])
    capi.destroy_fmi(fmi)
    fpr = capi.libc.fopen(path.encode(), b"rb")
    f2 = capi.read_index(None, fpr)
    capi.libc.fclose(fpr)
    assert (f2.contents.len, f2.contents.endloc, list(f2.contents.C)) == (n, rec["endloc"], rec["C"])
    sp2, ep2 = np.zeros(R, np.int32), np.zeros(R, np.int32)
    assert capi.fmi_search_batch(f2, _p(reads), m, m, R, _p(sp2, i32p), _p(ep2, i32p)) == 0
    np.testing.assert_array_equal(sp2, sp)
    np.testing.assert_array_equal(ep2, ep)
    capi.destroy_fmi(f2)
    # a truncated file: NULL and the reference's message (fileio.c:57-60)
    data = open(path, "rb").read()
    open(path, "wb").write(data[: len(data) // 2])
    fpr = capi.libc.fopen(path.encode(), b"rb")
    assert not capi.read_index(None, fpr)
    capi.libc.fclose(fpr)


def test_build_index_driver_writes_the_reference_file(oracle, tmp_path):
    g = GOLDEN["packer_1003"]
    seq = tmp_path / "seq.txt"
    seq.write_bytes(g["ascii"].encode("latin-1"))
    idx = tmp_path / "seq.idx"
    exe = os.path.join(ROOT, "bwt_amd", "drivers", "build_index")
    out = subprocess.run([exe, str(seq), str(idx)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert "Finished reading sequence from file" in out.stdout
    assert file_sha(str(idx)) == (g["index_file"], g["index_file_bytes"])
    # the 7-byte case recorded from the reference binary: TTT at the end reads as AAA
    seq.write_bytes(b"ACGTTTT")
    out = subprocess.run([exe, str(seq), str(idx)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    q = GOLDEN["packer_ACGTTTT"]
    want = np.array([7] + q["C"] + [q["endloc"]] + q["idxs"], np.uint32).tobytes() + bytes.fromhex(q["bwt_hex"])
    assert idx.read_bytes() == want


def test_fmitest_driver_counts_what_the_text_holds(tmp_path):
    """drivers/fmitest draws its text and its 10^7 overlapping 12-mers from
    rand(), as fmitest.c:127-153 does; the number of 12-mers it reports as found
    must be the number that occur in that text (counted here with numpy from the
    inputs the driver writes out when given a second argument)."""
    import re
    import numpy as np

    n, npat, m = 1000000, 10000000, 12
    exe = os.path.join(ROOT, "bwt_amd", "drivers", "fmitest")
    dump = tmp_path / "inputs.bin"
    out = subprocess.run([exe, str(n), str(dump)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert "Built index with 1000000 base pairs" in out.stdout
    assert "Searched 10000000 12bp sequences" in out.stdout
    got = int(re.search(r"\((\d+) of them occur in the text\)", out.stdout).group(1))

    raw = np.fromfile(dump, dtype=np.uint8)
    packed, pats = raw[: n // 4 + 1], raw[n // 4 + 1:]
    assert pats.size == npat + 11
    bases = ((packed[:, None] >> np.array([6, 4, 2, 0], dtype=np.uint8)) & 3).reshape(-1)[:n].astype(np.uint32)

    def codes(b, count):  # the m-mer starting at each of the first `count` positions, as a number
        acc = np.zeros(count, dtype=np.uint32)
        for k in range(m):
            acc = (acc << 2) | b[k:k + count]
        return acc

    in_text = np.zeros(1 << (2 * m), dtype=bool)
    in_text[codes(bases, n - m + 1)] = True
    want = int(in_text[codes(pats.astype(np.uint32), npat)].sum())
    assert 500000 < want < 650000  # about npat * n / 4^12
    assert got == want


def test_single_queries_on_the_host_and_on_the_device_agree(capi, oracle, monkeypatch):
    """rank / lf / unc_sa / loc_search / reverse_search / locate / mms answer one
    query from the host arrays of the fm_index; BWT_MI355X_SCALAR=device sends the
    same call to the GPU as a batch of one.  Both must be the oracle's answer."""
    capi.mms.argtypes = [C.POINTER(FM), u8p, C.c_int, i32p, i32p]
    n = 200003
    text = oracle.gen_genome(n, GENOME_SEED + 5)
    fo = oracle.make_fmi(text, n)
    fmi = capi.make_fmi(_p(text), n)
    from bwt_amd import synth
    bases = synth.unpack(text, n)
    rng = np.random.default_rng(8)
    rows = [0, 1, fo.endloc, n] + [int(x) for x in rng.integers(0, n + 1, 60)]
    pats = []
    for t in range(60):
        m = int(rng.integers(1, 70))
        j = int(rng.integers(0, n - m + 1))
        p = bases[j:j + m].copy()
        if t % 3 == 0:
            p[int(rng.integers(0, m))] ^= 2
        if t % 4 == 0 and m > 4:
            p[int(rng.integers(0, m - 1))] = 5
        pats.append(p)
    answers = {}
    for where in ("host", "device"):
        if where == "device":
            monkeypatch.setenv("BWT_MI355X_SCALAR", "device")
        else:
            monkeypatch.delenv("BWT_MI355X_SCALAR", raising=False)
        got = []
        for r in rows:
            got.append((capi.lf(fmi, r), capi.unc_sa(fmi, r), tuple(capi.rank(fmi, bytes([c]), r) for c in range(4))))
        sp, ep = C.c_int(), C.c_int()
        for p in pats:
            plain = p.copy()
            plain[plain == 5] = 0
            capi.loc_search(fmi, _p(plain), len(p), C.byref(sp), C.byref(ep))
            got.append((sp.value, ep.value, capi.reverse_search(fmi, _p(plain), len(p)),
                        capi.locate(fmi, _p(plain), len(p))))
            got.append((capi.mms(fmi, _p(p), len(p), C.byref(sp), C.byref(ep)), sp.value, ep.value))
        answers[where] = got
    assert answers["host"] == answers["device"]
    k = 0
    for r in rows:
        assert answers["host"][k] == (oracle.lf(fo, r), oracle.unc_sa(fo, r), tuple(oracle.rank(fo, c, r) for c in range(4)))
        k += 1
    for p in pats:
        plain = p.copy()
        plain[plain == 5] = 0
        s, e = (int(x) for x in oracle.loc_search(fo, plain))
        assert answers["host"][k][:3] == (s, e, oracle.reverse_search(fo, plain))
        assert answers["host"][k][3] == oracle.locate(fo, plain)
        assert answers["host"][k + 1] == tuple(oracle.mms(fo, p))
        k += 2
    capi.destroy_fmi(fmi)


def test_histhelper_sorts_a_run_of_suffixes(capi, oracle):
    """histhelper (histsortcomp.h:4): a run of suffix starts that share their first
    bases, handed over in any order, comes back in suffix-array order, in arr and
    in aux."""
    capi.histhelper.restype = None
    capi.histhelper.argtypes = [u8p, C.c_int, i32p, i32p, C.c_int, C.c_int, C.c_int]
    n = 60000
    text = oracle.gen_genome(n, GENOME_SEED + 2)
    sa = oracle.histsort(text, n).astype(np.int32)
    rng = np.random.default_rng(2)
    # the whole array, depth 0, as histsort calls it (histsortcomp.c:171)
    arr = sa.copy()
    rng.shuffle(arr)
    aux = np.zeros_like(arr)
    capi.histhelper(_p(text), n, _p(arr, i32p), _p(aux, i32p), 0, n + 1, 0)
    np.testing.assert_array_equal(arr, sa)
    np.testing.assert_array_equal(aux, sa)
    # one first-base bucket at depth 1, as the reference's threads call it (histsortcomp.c:107); the rest untouched
    from bwt_amd import synth
    bases = synth.unpack(text, n)
    lo = 1 + int((bases == 0).sum())
    hi = lo + int((bases == 1).sum())
    arr = sa.copy()
    part = arr[lo:hi].copy()
    rng.shuffle(part)
    arr[lo:hi] = part
    before = arr.copy()
    aux = np.full_like(arr, -5)
    capi.histhelper(_p(text), n, _p(arr, i32p), _p(aux, i32p), lo, hi, 1)
    np.testing.assert_array_equal(arr, sa)
    np.testing.assert_array_equal(aux[lo:hi], sa[lo:hi])
    assert (aux[:lo] == -5).all() and (aux[hi:] == -5).all() and (before[:lo] == arr[:lo]).all()
    # runs of zero and one element
    capi.histhelper(_p(text), n, _p(arr, i32p), _p(aux, i32p), 7, 7, 3)
    capi.histhelper(_p(text), n, _p(arr, i32p), _p(aux, i32p), 7, 8, 3)
    np.testing.assert_array_equal(arr, sa)
