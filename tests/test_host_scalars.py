"""The single-query functions of the reference API (rank, lf, unc_sa, loc_search,
reverse_search, locate, mms: seqindex.c:214-351) as libbwt_mi355x.so answers them
on the host, from the arrays of an fm_index -- against the oracle's restatement
of the reference, on an fm_index filled from the oracle's own arrays.  No GPU is
involved: one query per call is host work by design (the batch forms are the GPU
path, tests/test_gpu_capi.py)."""
import ctypes as C

import numpy as np
import pytest

from bwt_amd import _lib, synth


class FmIndex(C.Structure):  # seqindex.h:14-22
    _fields_ = [("bwt", C.c_void_p), ("idxs", C.c_void_p), ("rank_index", C.POINTER(C.c_void_p)),
                ("lookup", C.c_void_p), ("endloc", C.c_int), ("C", C.c_int * 5), ("len", C.c_int)]


def _lib_with_reference_api():
    lib = C.CDLL(_lib.LIB_PATH)
    P = C.POINTER(FmIndex)
    lib.rank.argtypes = [P, C.c_char, C.c_int]
    lib.lf.argtypes = [P, C.c_int]
    lib.unc_sa.argtypes = [P, C.c_int]
    lib.loc_search.argtypes = [P, C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.loc_search.restype = None
    lib.reverse_search.argtypes = [P, C.c_char_p, C.c_int]
    lib.locate.argtypes = [P, C.c_char_p, C.c_int]
    lib.mms.argtypes = [P, C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    return lib


def host_fm_index(fo, n):
    """An fm_index over the oracle's arrays, as a caller that fills the struct itself would build it."""
    keep = {"bwt": np.ascontiguousarray(fo.bwt), "idxs": np.ascontiguousarray(fo.idxs),
            "rows": np.ascontiguousarray(fo.rank_rows.astype(np.int32))}
    nrows = keep["rows"].shape[0]
    ptrs = (C.c_void_p * nrows)(*[keep["rows"].ctypes.data + 16 * j for j in range(nrows)])
    keep["ptrs"] = ptrs
    f = FmIndex()
    f.bwt = keep["bwt"].ctypes.data
    f.idxs = keep["idxs"].ctypes.data
    f.rank_index = C.cast(ptrs, C.POINTER(C.c_void_p))
    f.lookup = None
    f.endloc = fo.endloc
    for i in range(5):
        f.C[i] = fo.C[i]
    f.len = n
    return f, keep


@pytest.mark.parametrize("n,kind", [(13, "random"), (64, "random"), (777, "random"), (100003, "random"),
                                    (20000, "tandem")])
def test_host_single_queries_match_the_oracle(oracle, n, kind, monkeypatch):
    monkeypatch.delenv("BWT_MI355X_SCALAR", raising=False)
    lib = _lib_with_reference_api()
    text = oracle.gen_genome(n, synth.GENOME_SEED + n) if kind == "random" else oracle.gen_tandem(n, 7, 11)
    fo = oracle.make_fmi(text, n, oracle.sa_doubling(text, n) if kind != "random" else None)
    f, keep = host_fm_index(fo, n)
    rng = np.random.default_rng(n)
    rows = np.unique(np.concatenate([np.arange(min(n + 1, 200)), rng.integers(0, n + 1, 500), [fo.endloc, n]]))
    for r in rows:
        r = int(r)
        assert lib.lf(C.byref(f), r) == oracle.lf(fo, r)
        assert lib.unc_sa(C.byref(f), r) == oracle.unc_sa(fo, r)
        for c in range(4):
            assert lib.rank(C.byref(f), bytes([c]), r) == oracle.rank(fo, c, r)
    # rank takes n + 1 as well (the end of the last interval)
    for c in range(4):
        assert lib.rank(C.byref(f), bytes([c]), n + 1) == oracle.rank(fo, c, n + 1)
    bases = synth.unpack(text, n)
    sp, ep = C.c_int(), C.c_int()
    for t in range(400):
        m = int(rng.integers(1, min(n, 60) + 1))
        if t % 3 == 0 or n < m:
            pat = rng.integers(0, 4, m).astype(np.uint8)
        else:
            j = int(rng.integers(0, n - m + 1))
            pat = bases[j:j + m].copy()
            if t % 3 == 1 and m > 2:
                pat[int(rng.integers(0, m))] ^= 1
        b = pat.tobytes()
        lib.loc_search(C.byref(f), b, m, C.byref(sp), C.byref(ep))
        assert (sp.value, ep.value) == tuple(int(x) for x in oracle.loc_search(fo, pat))
        assert lib.reverse_search(C.byref(f), b, m) == oracle.reverse_search(fo, pat)
        want = oracle.locate(fo, pat)
        if not (want != -1 and int(oracle.loc_search(fo, pat)[0]) > n):  # sp = n + 1: out of bounds in the reference
            assert lib.locate(C.byref(f), b, m) == want
        # mms with Ns (5) sprinkled in, not all of them N
        q = pat.copy()
        if m > 3 and t % 2:
            q[rng.integers(0, m, 2)] = 5
            if (q == 5).all():
                q[0] = 1
        got = lib.mms(C.byref(f), q.tobytes(), m, C.byref(sp), C.byref(ep))
        w_len, w_sp, w_ep = oracle.mms(fo, q)
        assert (got, sp.value, ep.value) == (w_len, w_sp, w_ep)


def test_a_damaged_index_does_not_spin(oracle, monkeypatch):
    """unc_sa walks LF until a stored row; on an index whose LF map is not one cycle
    (here: a wrong $ row) the walk is cut off and reports -1."""
    monkeypatch.delenv("BWT_MI355X_SCALAR", raising=False)
    lib = _lib_with_reference_api()
    n = 5000
    text = oracle.gen_genome(n, 9)
    fo = oracle.make_fmi(text, n)
    f, keep = host_fm_index(fo, n)
    f.endloc = (fo.endloc + 1237) % n + 1
    out = [lib.unc_sa(C.byref(f), r) for r in range(1, 400)]
    assert all(-1 <= v <= n for v in out)
