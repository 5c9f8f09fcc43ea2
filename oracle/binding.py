"""ctypes bindings for the CPU oracle (oracle/libbwt_oracle.so) and, when it has
been built, the unmodified reference (oracle/_ref/libbwtref.so).

Test infrastructure only: the checker, never the thing measured or shipped.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(_HERE, "libbwt_oracle.so")
REF_SO = os.path.join(_HERE, "_ref", "libbwtref.so")

u8p = C.POINTER(C.c_uint8)
u32p = C.POINTER(C.c_uint32)
i32p = C.POINTER(C.c_int32)


def _p(a, t):
    return a.ctypes.data_as(t)


def build(force=False):
    """Compile the oracle (and the reference when its tree is present)."""
    if force or not os.path.exists(ORACLE_SO) or (
        os.path.isdir("/root/reference") and not os.path.exists(REF_SO)
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s", "all"])


class _OrcFmi(C.Structure):
    _fields_ = [
        ("bwt", u8p),
        ("idxs", u32p),
        ("rank_rows", u32p),
        ("lookup", C.c_uint8 * 1024),
        ("endloc", C.c_uint32),
        ("C", C.c_uint32 * 5),
        ("len", C.c_uint32),
    ]


class OrcIndex:
    """Owning wrapper around an orc_fmi*."""

    def __init__(self, lib, ptr):
        self._lib = lib
        self.ptr = ptr
        f = ptr.contents
        self.len = int(f.len)
        self.endloc = int(f.endloc)
        self.C = [int(x) for x in f.C]

    def _view(self, p, count, dtype):
        if count == 0:
            return np.zeros(0, dtype)
        return np.ctypeslib.as_array(p, shape=(count,)).view(dtype).copy()

    @property
    def bwt(self):
        return self._view(self.ptr.contents.bwt, (self.len + 3) // 4, np.uint8)

    @property
    def idxs(self):
        return self._view(self.ptr.contents.idxs, 1 + self.len // 32, np.uint32)

    @property
    def rank_rows(self):
        rows = 1 + (self.len + 15) // 16
        return self._view(self.ptr.contents.rank_rows, rows * 4, np.uint32).reshape(rows, 4)

    def __del__(self):
        if getattr(self, "ptr", None):
            self._lib.orc_destroy_fmi(self.ptr)
            self.ptr = None


class Oracle:
    def __init__(self, path=ORACLE_SO):
        L = C.CDLL(path)
        self.L = L
        fp = C.POINTER(_OrcFmi)
        L.orc_rnd.restype = C.c_uint64
        L.orc_rnd.argtypes = [C.c_uint64, C.c_uint64]
        L.orc_gen_genome.argtypes = [u8p, C.c_uint64, C.c_uint64]
        L.orc_gen_tandem.argtypes = [u8p, C.c_uint64] + [C.c_uint64] * 4
        L.orc_gen_reads.argtypes = [u8p, C.c_uint64, u8p, C.c_uint64, C.c_uint32, C.c_uint64, u32p]
        L.orc_histsort.restype = C.c_void_p
        L.orc_histsort.argtypes = [u8p, C.c_uint32, C.c_int, C.c_uint32]
        L.orc_sa_doubling.restype = C.c_void_p
        L.orc_sa_doubling.argtypes = [u8p, C.c_uint32]
        L.orc_check_sa.argtypes = [u8p, C.c_uint32, u32p]
        L.orc_sprintcbwt.restype = C.c_uint32
        L.orc_sprintcbwt.argtypes = [u8p, u32p, C.c_uint32, u8p]
        L.orc_lookup_table.argtypes = [u8p]
        L.orc_seq_index.restype = C.c_void_p
        L.orc_seq_index.argtypes = [u8p, C.c_uint32]
        L.orc_make_fmi.restype = fp
        L.orc_make_fmi.argtypes = [u8p, C.c_uint32, u32p]
        L.orc_destroy_fmi.argtypes = [fp]
        L.orc_rank.restype = C.c_uint32
        L.orc_rank.argtypes = [fp, C.c_uint, C.c_uint32]
        L.orc_lf.restype = C.c_uint32
        L.orc_lf.argtypes = [fp, C.c_uint32]
        L.orc_unc_sa.restype = C.c_uint32
        L.orc_unc_sa.argtypes = [fp, C.c_uint32, u32p]
        L.orc_reverse_search.argtypes = [fp, u8p, C.c_int]
        L.orc_loc_search.argtypes = [fp, u8p, C.c_int, u32p, u32p]
        L.orc_locate.restype = C.c_int64
        L.orc_locate.argtypes = [fp, u8p, C.c_int]
        L.orc_mms.argtypes = [fp, u8p, C.c_int, u32p, u32p]
        L.orc_search_reads_one.argtypes = [fp, u8p, C.c_int, C.c_int, u32p]
        L.orc_dp_align.argtypes = [u8p, C.c_int, u8p, C.c_int, C.c_int, u8p, u32p]
        L.orc_align_read_anchored.argtypes = [fp, u8p, u8p, C.c_int, C.c_int, u8p, u32p]
        L.orc_search_batch.argtypes = [fp, u8p, C.c_uint32, C.c_uint64, C.c_uint64, u32p, u32p, C.c_int]
        L.orc_locate_rows.argtypes = [fp, u32p, C.c_uint64, u32p, C.POINTER(C.c_uint64), C.c_int]
        L.orc_write_index_path.argtypes = [fp, C.c_char_p]
        L.orc_read_index_path.restype = fp
        L.orc_read_index_path.argtypes = [C.c_char_p]
        L.orc_pack_ascii.argtypes = [u8p, C.c_uint64, u8p]
        self._libc = C.CDLL(None)
        self._libc.free.argtypes = [C.c_void_p]

    # ---- inputs ----
    def gen_genome(self, n, seed):
        t = np.zeros(n // 4 + 1, np.uint8)
        self.L.orc_gen_genome(_p(t, u8p), n, seed)
        return t

    def gen_tandem(self, n, period, unit_seed, mut_every=0, mut_seed=0):
        t = np.zeros(n // 4 + 1, np.uint8)
        self.L.orc_gen_tandem(_p(t, u8p), n, period, unit_seed, mut_every, mut_seed)
        return t

    def gen_reads(self, text, n, R, m, seed):
        reads = np.zeros((R, m), np.uint8)
        offs = np.zeros(R, np.uint32)
        self.L.orc_gen_reads(_p(text, u8p), n, _p(reads, u8p), R, m, seed, _p(offs, u32p))
        return reads, offs

    # ---- SA ----
    def _take_u32(self, ptr, count):
        if not ptr:
            return None
        a = np.ctypeslib.as_array(C.cast(ptr, u32p), shape=(count,)).copy()
        self._libc.free(ptr)
        return a

    def histsort(self, text, n, threads=1, max_depth=0):
        return self._take_u32(self.L.orc_histsort(_p(text, u8p), n, threads, max_depth), n + 1)

    def sa_doubling(self, text, n):
        return self._take_u32(self.L.orc_sa_doubling(_p(text, u8p), n), n + 1)

    def check_sa(self, text, n, sa):
        sa = np.ascontiguousarray(sa, np.uint32)
        return self.L.orc_check_sa(_p(text, u8p), n, _p(sa, u32p))

    def sprintcbwt(self, text, sa, n):
        sa = np.ascontiguousarray(sa, np.uint32)
        out = np.zeros((n + 3) // 4, np.uint8)
        e = self.L.orc_sprintcbwt(_p(text, u8p), _p(sa, u32p), n, _p(out, u8p))
        return out, int(e)

    def lookup_table(self):
        t = np.zeros(1024, np.uint8)
        self.L.orc_lookup_table(_p(t, u8p))
        return t

    def seq_index(self, bwt, n):
        rows = 1 + (n + 15) // 16
        return self._take_u32(self.L.orc_seq_index(_p(bwt, u8p), n), rows * 4).reshape(rows, 4)

    # ---- index ----
    def make_fmi(self, text, n, sa=None):
        if sa is not None:
            sa = np.ascontiguousarray(sa, np.uint32)
            ptr = self.L.orc_make_fmi(_p(text, u8p), n, _p(sa, u32p))
        else:
            ptr = self.L.orc_make_fmi(_p(text, u8p), n, None)
        return OrcIndex(self.L, ptr)

    def rank(self, f, c, idx):
        return int(self.L.orc_rank(f.ptr, c, idx))

    def lf(self, f, idx):
        return int(self.L.orc_lf(f.ptr, idx))

    def unc_sa(self, f, idx):
        return int(self.L.orc_unc_sa(f.ptr, idx, None))

    def reverse_search(self, f, pat):
        pat = np.ascontiguousarray(pat, np.uint8)
        return int(self.L.orc_reverse_search(f.ptr, _p(pat, u8p), len(pat)))

    def loc_search(self, f, pat):
        pat = np.ascontiguousarray(pat, np.uint8)
        sp, ep = C.c_uint32(), C.c_uint32()
        self.L.orc_loc_search(f.ptr, _p(pat, u8p), len(pat), C.byref(sp), C.byref(ep))
        return sp.value, ep.value

    def locate(self, f, pat):
        pat = np.ascontiguousarray(pat, np.uint8)
        return int(self.L.orc_locate(f.ptr, _p(pat, u8p), len(pat)))

    def mms(self, f, pat):
        pat = np.ascontiguousarray(pat, np.uint8)
        sp, ep = C.c_uint32(), C.c_uint32()
        k = self.L.orc_mms(f.ptr, _p(pat, u8p), len(pat), C.byref(sp), C.byref(ep))
        return int(k), sp.value, ep.value

    def search_reads_one(self, f, read, anchor=20):
        read = np.ascontiguousarray(read, np.uint8)
        out = np.zeros(4, np.uint32)
        self.L.orc_search_reads_one(f.ptr, _p(read, u8p), len(read), anchor, _p(out, u32p))
        return out

    ALIGN_UNDEFINED = -2 ** 31

    def align_read_anchored(self, f, text, pattern, anchor_len=12):
        """align_read_anchored (single_align.c:163-298): (position or 0, unit CIGAR operations in push order)"""
        pattern = np.ascontiguousarray(pattern, np.uint8)
        ops = np.zeros(5 * len(pattern) + 64, np.uint8)
        nops = C.c_uint32()
        pos = self.L.orc_align_read_anchored(f.ptr, _p(text, u8p), _p(pattern, u8p), len(pattern), anchor_len,
                                             _p(ops, u8p), C.byref(nops))
        return int(pos), ops[: nops.value].tobytes()

    def dp_align(self, s1, s2, mode):
        """nw_fast (mode 0) / sw_fast (mode 1): (return value, unit CIGAR operations as bytes)"""
        s1 = np.ascontiguousarray(s1, np.uint8)
        s2 = np.ascontiguousarray(s2, np.uint8)
        ops = np.zeros(len(s1) + len(s2) + 1, np.uint8)
        nops = C.c_uint32()
        ret = self.L.orc_dp_align(_p(s1, u8p), len(s1), _p(s2, u8p), len(s2), mode, _p(ops, u8p), C.byref(nops))
        return int(ret), ops[: nops.value].tobytes()

    def search_batch(self, f, pats, m, stride, R, threads=4):
        pats = np.ascontiguousarray(pats, np.uint8)
        sp = np.zeros(R, np.uint32)
        ep = np.zeros(R, np.uint32)
        self.L.orc_search_batch(f.ptr, _p(pats, u8p), m, stride, R, _p(sp, u32p), _p(ep, u32p), threads)
        return sp, ep

    def locate_rows(self, f, rows, threads=4):
        rows = np.ascontiguousarray(rows, np.uint32)
        pos = np.zeros(len(rows), np.uint32)
        steps = C.c_uint64()
        self.L.orc_locate_rows(f.ptr, _p(rows, u32p), len(rows), _p(pos, u32p), C.byref(steps), threads)
        return pos, int(steps.value)

    def write_index(self, f, path):
        return self.L.orc_write_index_path(f.ptr, os.fsencode(path))

    def read_index(self, path):
        ptr = self.L.orc_read_index_path(os.fsencode(path))
        return OrcIndex(self.L, ptr) if ptr else None

    def pack_ascii(self, ascii_bytes):
        a = np.frombuffer(bytes(ascii_bytes), np.uint8)
        out = np.zeros(len(a) // 4 + 1, np.uint8)
        self.L.orc_pack_ascii(_p(a, u8p) if len(a) else None, len(a), _p(out, u8p))
        return out


# --------------------------------------------------------------------------
# the unmodified reference (present only where oracle/_ref has been built)
# --------------------------------------------------------------------------


class _RefFmi(C.Structure):
    # seqindex.h:14-22
    _fields_ = [
        ("bwt", u8p),
        ("idxs", i32p),
        ("rank_index", C.POINTER(i32p)),
        ("lookup", u8p),
        ("endloc", C.c_int),
        ("C", C.c_int * 5),
        ("len", C.c_int),
    ]


class RefIndex:
    def __init__(self, lib, ptr):
        self._lib = lib
        self.ptr = ptr
        f = ptr.contents
        self.len = int(f.len)
        self.endloc = int(f.endloc)
        self.C = [int(x) for x in f.C]

    @property
    def bwt(self):
        return np.ctypeslib.as_array(self.ptr.contents.bwt, shape=((self.len + 3) // 4,)).copy()

    @property
    def idxs(self):
        return np.ctypeslib.as_array(self.ptr.contents.idxs, shape=(1 + self.len // 32,)).astype(np.uint32)

    @property
    def rank_rows(self):
        rows = 1 + (self.len + 15) // 16
        out = np.zeros(rows * 4, np.int32)
        self._lib.ref_copy_rank_rows(self.ptr, _p(out, i32p))
        return out.astype(np.uint32).reshape(rows, 4)

    def __del__(self):
        if getattr(self, "ptr", None):
            self._lib.destroy_fmi(self.ptr)
            self.ptr = None


class Reference:
    """The compiled reference; inputs must satisfy its limits (len < 2^31,
    base len reads 0, patterns 0..3)."""

    def __init__(self, path=REF_SO):
        L = C.CDLL(path)
        self.L = L
        fp = C.POINTER(_RefFmi)
        L.histsort.restype = C.c_void_p
        L.histsort.argtypes = [u8p, C.c_int]
        L.csuff_arr.restype = C.c_void_p
        L.csuff_arr.argtypes = [u8p, C.c_uint]
        L.sprintcbwt.restype = C.c_uint
        L.sprintcbwt.argtypes = [u8p, u32p, C.c_uint, u8p]
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
        L.mms.argtypes = [fp, u8p, C.c_int, i32p, i32p]
        L.lookup_table.restype = C.c_void_p
        L.ref_loc_search_mt.argtypes = [fp, u8p, C.c_int, C.c_long, C.c_long, i32p, i32p, C.c_int]
        L.ref_reverse_search_mt.restype = C.c_long
        L.ref_reverse_search_mt.argtypes = [fp, u8p, C.c_int, C.c_long, C.c_long, C.c_int]
        L.ref_unc_sa_mt.argtypes = [fp, i32p, C.c_long, i32p, C.c_int]
        L.ref_dp_align.argtypes = [u8p, C.c_int, u8p, C.c_int, C.c_int, u8p, u32p]
        L.ref_align_read_anchored.argtypes = [fp, u8p, u8p, C.c_int, C.c_int, u8p, u32p]
        L.ref_write_index_path.argtypes = [fp, C.c_char_p]
        L.ref_read_index_path.restype = fp
        L.ref_read_index_path.argtypes = [C.c_char_p]
        L.ref_copy_rank_rows.argtypes = [fp, i32p]
        self._libc = C.CDLL(None)
        self._libc.free.argtypes = [C.c_void_p]

    def _take(self, ptr, count):
        a = np.ctypeslib.as_array(C.cast(ptr, u32p), shape=(count,)).copy()
        self._libc.free(ptr)
        return a

    def histsort(self, text, n):
        return self._take(self.L.histsort(_p(text, u8p), n), n + 1)

    def align_read_anchored(self, f, text, pattern, anchor_len=12):
        """the reference's align_read_anchored on a fresh stack, as single_align.c:479-483 calls it"""
        pattern = np.ascontiguousarray(pattern, np.uint8)
        ops = np.zeros(5 * len(pattern) + 64, np.uint8)
        nops = C.c_uint32()
        pos = self.L.ref_align_read_anchored(f.ptr, _p(text, u8p), _p(pattern, u8p), len(pattern), anchor_len,
                                             _p(ops, u8p), C.byref(nops))
        return int(pos), ops[: nops.value].tobytes()

    def dp_align(self, s1, s2, mode):
        """the reference's nw_fast (mode 0) / sw_fast (mode 1) on a fresh stack"""
        s1 = np.ascontiguousarray(s1, np.uint8)
        s2 = np.ascontiguousarray(s2, np.uint8)
        ops = np.zeros(len(s1) + len(s2) + 1, np.uint8)
        nops = C.c_uint32()
        ret = self.L.ref_dp_align(_p(s1, u8p), len(s1), _p(s2, u8p), len(s2), mode, _p(ops, u8p), C.byref(nops))
        return int(ret), ops[: nops.value].tobytes()

    def csuff_arr(self, text, n):
        return self._take(self.L.csuff_arr(_p(text, u8p), n), n + 1)

    def sprintcbwt(self, text, sa, n):
        sa = np.ascontiguousarray(sa, np.uint32)
        out = np.zeros((n + 3) // 4, np.uint8)
        e = self.L.sprintcbwt(_p(text, u8p), _p(sa, u32p), n, _p(out, u8p))
        return out, int(e)

    def lookup_table(self):
        p = self.L.lookup_table()
        a = np.ctypeslib.as_array(C.cast(p, u8p), shape=(1024,)).copy()
        self._libc.free(p)
        return a

    def make_fmi(self, text, n, sacak=False):
        fn = self.L.make_fmi_sacak if sacak else self.L.make_fmi
        return RefIndex(self.L, fn(_p(text, u8p), n))

    def rank(self, f, c, idx):
        return int(self.L.rank(f.ptr, bytes([c]), idx))

    def lf(self, f, idx):
        return int(self.L.lf(f.ptr, idx))

    def unc_sa(self, f, idx):
        return int(self.L.unc_sa(f.ptr, idx))

    def reverse_search(self, f, pat):
        pat = np.ascontiguousarray(pat, np.uint8)
        return int(self.L.reverse_search(f.ptr, _p(pat, u8p), len(pat)))

    def loc_search(self, f, pat):
        pat = np.ascontiguousarray(pat, np.uint8)
        sp, ep = C.c_int(), C.c_int()
        self.L.loc_search(f.ptr, _p(pat, u8p), len(pat), C.byref(sp), C.byref(ep))
        return sp.value, ep.value

    def locate(self, f, pat):
        """prints a warning on stdout when the read has several hits"""
        pat = np.ascontiguousarray(pat, np.uint8)
        return int(self.L.locate(f.ptr, _p(pat, u8p), len(pat)))

    def mms(self, f, pat):
        pat = np.ascontiguousarray(pat, np.uint8)
        sp, ep = C.c_int(), C.c_int()
        k = self.L.mms(f.ptr, _p(pat, u8p), len(pat), C.byref(sp), C.byref(ep))
        return int(k), sp.value, ep.value

    def search_batch(self, f, pats, m, stride, R, threads=4):
        pats = np.ascontiguousarray(pats, np.uint8)
        sp = np.zeros(R, np.int32)
        ep = np.zeros(R, np.int32)
        self.L.ref_loc_search_mt(f.ptr, _p(pats, u8p), m, stride, R, _p(sp, i32p), _p(ep, i32p), threads)
        return sp.astype(np.uint32), ep.astype(np.uint32)

    def reverse_search_batch(self, f, pats, m, stride, R, threads=4):
        pats = np.ascontiguousarray(pats, np.uint8)
        return int(self.L.ref_reverse_search_mt(f.ptr, _p(pats, u8p), m, stride, R, threads))

    def locate_rows(self, f, rows, threads=4):
        rows = np.ascontiguousarray(rows, np.int32)
        pos = np.zeros(len(rows), np.int32)
        self.L.ref_unc_sa_mt(f.ptr, _p(rows, i32p), len(rows), _p(pos, i32p), threads)
        return pos.astype(np.uint32)

    def write_index(self, f, path):
        return self.L.ref_write_index_path(f.ptr, os.fsencode(path))

    def read_index(self, path):
        ptr = self.L.ref_read_index_path(os.fsencode(path))
        return RefIndex(self.L, ptr) if ptr else None


_oracle = None
_reference = None


def load_oracle():
    global _oracle
    if _oracle is None:
        build()
        _oracle = Oracle()
    return _oracle


def load_reference():
    """Returns the compiled reference, or None where oracle/_ref is absent."""
    global _reference
    if _reference is None and os.path.exists(REF_SO):
        _reference = Reference()
    return _reference
