"""Thin Python plumbing over the C-ABI: torch owns device memory and streams,
libbwt_mi355x.so does the work.  Used by the tests and bench.py; the drop-in
boundary for C callers is the shared library itself (include/*.h).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import BwtGpuError, check

NONE_ROW = 0xFFFFFFFF


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


def init(device=0):
    lib = _lib.load()
    check(lib.bwtgpu_init(int(device)))
    torch.cuda.set_device(int(device))
    return lib


def _as_u32(t):
    """torch has no uint32 arithmetic; device rows are carried as int32 bits."""
    return t.view(torch.int32) if t.dtype != torch.int32 else t


def to_numpy_u32(t):
    return t.detach().cpu().numpy().view(np.uint32)


def timing_enable(on=True):
    check(_lib.load().bwtgpu_timing_enable(1 if on else 0))


def timing_reset():
    check(_lib.load().bwtgpu_timing_reset())


def timing_get(name):
    n = C.c_uint64()
    ms = C.c_double()
    check(_lib.load().bwtgpu_timing_get(name.encode(), C.byref(n), C.byref(ms)))
    return int(n.value), float(ms.value)


def timing_report():
    buf = C.create_string_buffer(1 << 16)
    k = _lib.load().bwtgpu_timing_report(buf, len(buf))
    out = {}
    for line in buf.raw[:k].decode().splitlines():
        name, launches, ms = line.split()
        out[name] = (int(launches), float(ms))
    return out


class GpuIndex:
    """Owning handle of an HBM-resident FM-index (bwtgpu_index*)."""

    def __init__(self, handle):
        if not handle:
            raise BwtGpuError("index construction failed: " + _lib.last_error())
        self.h = C.c_void_p(handle)
        n, e = C.c_uint32(), C.c_uint32()
        Cs = (C.c_uint32 * 5)()
        check(_lib.load().bwtgpu_index_info(self.h, C.byref(n), C.byref(e), Cs))
        self.n, self.endloc, self.C = int(n.value), int(e.value), [int(x) for x in Cs]
        self._ws = None
        self._lws = None

    def close(self):
        if self.h:
            _lib.load().bwtgpu_index_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def keeps_suffix_array_by_default(self):
        """False when the library is told to build compact indexes, or the text is
        too long for the position records (see bwtgpu_index_attach_sa)."""
        import os
        return not os.environ.get("BWT_MI355X_COMPACT") and self.n <= 0xFFFFFF00

    @property
    def device_bytes(self):
        return int(_lib.load().bwtgpu_index_device_bytes(self.h))

    def export(self):
        """(bwt bytes, idxs) in the reference's stored layout (fileio.c:14-15)."""
        bwt = np.zeros((self.n + 3) // 4, np.uint8)
        idxs = np.zeros(1 + self.n // 32, np.uint32)
        check(_lib.load().bwtgpu_index_export(self.h, bwt.ctypes.data_as(C.c_void_p),
                                              idxs.ctypes.data_as(C.c_void_p)))
        return bwt, idxs

    def rank_rows16(self):
        rows = 1 + (self.n + 15) // 16
        out = np.zeros((rows, 4), np.uint32)
        check(_lib.load().bwtgpu_index_rank_rows16(self.h, out.ctypes.data_as(C.c_void_p)))
        return out

    def _workspace(self, nbytes):
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        return self._ws

    def search(self, d_pats, m, R, stride=None, want_locrow=False, out=None):
        """d_pats: uint8 CUDA tensor, read r at offset r*stride (default m).
        Returns (sp, ep[, locrow]) as int32 CUDA tensors holding uint32 bits."""
        lib = _lib.load()
        stride = m if stride is None else stride
        need = (R - 1) * stride + m if R else 0
        if d_pats.numel() < need:
            raise ValueError(f"pattern buffer holds {d_pats.numel()} bytes, needs {need}")
        if out is None:
            sp = torch.empty(R, dtype=torch.int32, device="cuda")
            ep = torch.empty(R, dtype=torch.int32, device="cuda")
            lr = torch.empty(R, dtype=torch.int32, device="cuda") if want_locrow else None
        else:
            sp, ep, lr = out
        wsb = int(lib.bwtgpu_search_workspace_bytes(R, m))
        ws = self._workspace(wsb)
        check(lib.bwtgpu_search(self.h, _ptr(d_pats), m, stride, R, _ptr(sp), _ptr(ep), _ptr(lr),
                                _ptr(ws), ws.numel(), _stream()))
        return (sp, ep, lr) if want_locrow else (sp, ep)

    def search_locate(self, d_pats, m, R, stride=None, out=None):
        """bwtgpu_search_locate: loc_search and locate() of every read in one call.
        Returns (sp, ep, pos) as int32 CUDA tensors holding uint32 bits."""
        lib = _lib.load()
        stride = m if stride is None else stride
        need = (R - 1) * stride + m if R else 0
        if d_pats.numel() < need:
            raise ValueError(f"pattern buffer holds {d_pats.numel()} bytes, needs {need}")
        sp, ep, pos = out if out is not None else (torch.empty(R, dtype=torch.int32, device="cuda") for _ in range(3))
        ws = self._workspace(int(lib.bwtgpu_search_locate_workspace_bytes(R, m)))
        check(lib.bwtgpu_search_locate(self.h, _ptr(d_pats), m, stride, R, _ptr(sp), _ptr(ep), _ptr(pos),
                                       _ptr(ws), ws.numel(), _stream()))
        return sp, ep, pos

    def search_locate_packed(self, d_words, m, R, out=None):
        """bwtgpu_search_locate_packed: the same for reads packed to 2 bits per base."""
        lib = _lib.load()
        wpr = (m + 15) // 16
        if d_words.numel() < R * wpr:
            raise ValueError(f"packed buffer holds {d_words.numel()} words, needs {R * wpr}")
        sp, ep, pos = out if out is not None else (torch.empty(R, dtype=torch.int32, device="cuda") for _ in range(3))
        ws = self._workspace(int(lib.bwtgpu_search_locate_packed_workspace_bytes(R)))
        check(lib.bwtgpu_search_locate_packed(self.h, _ptr(d_words), m, R, _ptr(sp), _ptr(ep), _ptr(pos),
                                              _ptr(ws), ws.numel(), _stream()))
        return sp, ep, pos

    def search_static(self, d_pats, m, R, stride=None, want_locrow=False):
        """bwtgpu_search_static: the same search with the library's own static
        scratch in place of a workspace argument."""
        stride = m if stride is None else stride
        need = (R - 1) * stride + m if R else 0
        if d_pats.numel() < need:
            raise ValueError(f"pattern buffer holds {d_pats.numel()} bytes, needs {need}")
        sp = torch.empty(R, dtype=torch.int32, device="cuda")
        ep = torch.empty(R, dtype=torch.int32, device="cuda")
        lr = torch.empty(R, dtype=torch.int32, device="cuda") if want_locrow else None
        check(_lib.load().bwtgpu_search_static(self.h, _ptr(d_pats), m, stride, R, _ptr(sp), _ptr(ep), _ptr(lr),
                                               _stream()))
        return (sp, ep, lr) if want_locrow else (sp, ep)

    def search_packed(self, d_words, m, R, want_locrow=False, out=None):
        """bwtgpu_search_packed: reads already packed to 2 bits per base, read r the
        (m+15)//16 int32 words at d_words[r * ((m+15)//16):] (pack_reads_dev's format)."""
        wpr = (m + 15) // 16
        if d_words.numel() < R * wpr:
            raise ValueError(f"packed buffer holds {d_words.numel()} words, needs {R * wpr}")
        if out is None:
            sp = torch.empty(R, dtype=torch.int32, device="cuda")
            ep = torch.empty(R, dtype=torch.int32, device="cuda")
            lr = torch.empty(R, dtype=torch.int32, device="cuda") if want_locrow else None
        else:
            sp, ep, lr = out
        check(_lib.load().bwtgpu_search_packed(self.h, _ptr(d_words), m, R, _ptr(sp), _ptr(ep), _ptr(lr), _stream()))
        return (sp, ep, lr) if want_locrow else (sp, ep)

    def search_loads(self, d_pats, m, R, stride=None):
        """Loads one search of the batch makes: dict of counts (see
        bwtgpu_search_profile) and the bytes they add up to."""
        lib = _lib.load()
        stride = m if stride is None else stride
        need = (R - 1) * stride + m if R else 0
        if d_pats.numel() < need:
            raise ValueError(f"pattern buffer holds {d_pats.numel()} bytes, needs {need}")
        sp = torch.empty(R, dtype=torch.int32, device="cuda")
        ep = torch.empty(R, dtype=torch.int32, device="cuda")
        ws = torch.empty(int(lib.bwtgpu_search_workspace_bytes(R, m)) + 256, dtype=torch.uint8, device="cuda")
        loads = (C.c_uint64 * 5)()
        check(lib.bwtgpu_search_profile(self.h, _ptr(d_pats), m, stride, R, _ptr(sp), _ptr(ep), _ptr(ws),
                                        ws.numel(), loads, _stream()))
        blk, txt, ent, pat, kmr = (int(x) for x in loads)
        k, w = self.kmer_table
        per_entry = 8  # two neighbouring starts, or a start and its position word; a further start read is among `ent`
        return {"rank_blocks_16B": blk, "text_windows_16B": txt, "sa_isa_entries_4B": ent,
                "pattern_words_4B": pat, "kmer_entries": kmr, "kmer_entry_bytes": per_entry, "kmer_k": k,
                "bytes": 16 * blk + 16 * txt + 4 * ent + 4 * pat + per_entry * kmr}, sp, ep

    @property
    def kmer_table(self):
        """(bases per string, words per entry) of the k-mer table, (0, 0) when there is none (yet)."""
        k, w = C.c_uint32(), C.c_uint32()
        check(_lib.load().bwtgpu_index_kmer_table(self.h, C.byref(k), C.byref(w)))
        return int(k.value), int(w.value)

    def attach_text(self, text):
        """Give an index loaded from its stored parts the text it was built from
        (numpy on the host or a CUDA tensor); see bwtgpu_index_attach_text."""
        lib = _lib.load()
        if isinstance(text, torch.Tensor):
            check(lib.bwtgpu_index_attach_text_dev(self.h, _ptr(text), _stream()))
        else:
            text = np.ascontiguousarray(text, np.uint8)
            assert len(text) >= self.n // 4 + 1
            check(lib.bwtgpu_index_attach_text(self.h, text.ctypes.data_as(C.c_void_p), _stream()))

    def compact(self):
        """Drop the kept suffix array, its inverse and the text (see bwtgpu_index_compact)."""
        check(_lib.load().bwtgpu_index_compact(self.h))

    def certify(self, d_text):
        """bwtgpu_index_certify: dict of violation counts, all zero for a correct index."""
        lib = _lib.load()
        ws = torch.empty(int(lib.bwtgpu_sa_certify_workspace_bytes(self.n)), dtype=torch.uint8, device="cuda")
        rep = (C.c_uint64 * 8)()
        check(lib.bwtgpu_index_certify(self.h, _ptr(d_text), _ptr(ws), ws.numel(), rep, _stream()))
        return dict(zip(("out_of_range", "not_a_permutation", "order", "bwt_blocks", "samples", "endloc"),
                        (int(x) for x in rep)))

    def replicate(self, device):
        """A copy of the index in the memory of GPU `device` (bwtgpu_index_replicate)."""
        return GpuIndex(_lib.load().bwtgpu_index_replicate(self.h, int(device)))

    def prepare_search(self):
        """Make the search-only arrays now (see bwtgpu_index_prepare_search)."""
        check(_lib.load().bwtgpu_index_prepare_search(self.h, _stream()))

    def locate_rows(self, d_rows, count_steps=False, out=None):
        lib = _lib.load()
        R = d_rows.numel()
        pos = torch.empty(R, dtype=torch.int32, device="cuda") if out is None else out
        steps = torch.zeros(1, dtype=torch.int64, device="cuda") if count_steps else None
        if self._lws is None:
            self._lws = torch.empty(int(lib.bwtgpu_locate_workspace_bytes()), dtype=torch.uint8,
                                    device="cuda")
        check(lib.bwtgpu_locate_rows(self.h, _ptr(d_rows), R, _ptr(pos), _ptr(steps),
                                     _ptr(self._lws), self._lws.numel(), _stream()))
        if count_steps:
            return pos, int(steps.item())
        return pos


def index_from_parts(bwt, n, endloc, idxs, lean=False):
    """Stored parts of an index (numpy, host) -> HBM index; the device-side
    counterpart of read_index's rebuild (fileio.c:62-63).  lean: the index stays
    on rank blocks, sampled rows and k-mer table (bwtgpu_index_keep_lean); else
    its first search recomputes suffix array and text and it searches like a
    built one."""
    bwt = np.ascontiguousarray(bwt, np.uint8)
    idxs = np.ascontiguousarray(idxs, np.uint32)
    assert len(bwt) >= (n + 3) // 4 and len(idxs) >= 1 + n // 32
    h = _lib.load().bwtgpu_index_from_parts(bwt.ctypes.data_as(C.c_void_p), n, endloc,
                                            idxs.ctypes.data_as(C.c_void_p), _stream())
    ix = GpuIndex(h)
    if lean:
        check(_lib.load().bwtgpu_index_keep_lean(ix.h))
    return ix


def build_index(text, n):
    """make_fmi (seqindex.c:161-189) from a packed text on the host."""
    text = np.ascontiguousarray(text, np.uint8)
    assert len(text) >= n // 4 + 1
    return GpuIndex(_lib.load().bwtgpu_index_build(text.ctypes.data_as(C.c_void_p), n, _stream()))


def build_index_dev(d_text, n):
    assert d_text.numel() >= n // 4 + 1
    return GpuIndex(_lib.load().bwtgpu_index_build_dev(_ptr(d_text), n, _stream()))


def sa_build(d_text, n, ws=None):
    """histsort / csuff_arr on device: returns the n+1 SA rows (int32 bits)."""
    lib = _lib.load()
    sa = torch.empty(n + 1, dtype=torch.int32, device="cuda")
    wsb = int(lib.bwtgpu_sa_workspace_bytes(n))
    if ws is None or ws.numel() < wsb:
        ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    check(lib.bwtgpu_sa_build(_ptr(d_text), n, _ptr(sa), _ptr(ws), ws.numel(), _stream()))
    return sa


def sa_certify(d_text, n, d_sa):
    """bwtgpu_sa_certify: (out_of_range, not_a_permutation, order) violation counts."""
    lib = _lib.load()
    ws = torch.empty(int(lib.bwtgpu_sa_certify_workspace_bytes(n)), dtype=torch.uint8, device="cuda")
    rep = (C.c_uint64 * 8)()
    check(lib.bwtgpu_sa_certify(_ptr(d_text), n, _ptr(d_sa), _ptr(ws), ws.numel(), rep, _stream()))
    return int(rep[0]), int(rep[1]), int(rep[2])


def sa_bucket_counts(d_text, n):
    """Suffixes per leading four-base string (256 counts, numpy uint64)."""
    out = np.zeros(256, np.uint64)
    check(_lib.load().bwtgpu_sa_bucket_counts(_ptr(d_text), n, out.ctypes.data_as(C.c_void_p), _stream()))
    return out


def sa_build_range(d_text, n, lo, hi, d_rows, d_bw=None, ws=None):
    """One shard of the suffix array (see bwtgpu_sa_build_range) written to the
    front of d_rows / d_bw.  Returns (row_begin, rows, endloc or 0)."""
    lib = _lib.load()
    wsb = int(lib.bwtgpu_sa_workspace_bytes(n))
    if ws is None or ws.numel() < wsb:
        ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    begin, rows, endloc = C.c_uint64(), C.c_uint64(), C.c_uint32()
    cap = d_rows.numel() if d_bw is None else min(d_rows.numel(), d_bw.numel())
    check(lib.bwtgpu_sa_build_range(_ptr(d_text), n, lo, hi, _ptr(d_rows), _ptr(d_bw), cap, C.byref(begin),
                                    C.byref(rows), C.byref(endloc), _ptr(ws), ws.numel(), _stream()))
    return int(begin.value), int(rows.value), int(endloc.value)


def build_index_from_sa(d_text, n, d_sa, d_bw=None, endloc=0):
    """make_fmi's tail from a suffix array in HBM (n+1 rows)."""
    assert d_sa.numel() >= n + 1 and (d_bw is None or d_bw.numel() >= n)
    return GpuIndex(_lib.load().bwtgpu_index_build_from_sa(_ptr(d_text), n, _ptr(d_sa), _ptr(d_bw), endloc,
                                                           _stream()))


def pack_reads_dev(d_pats, m, R, chunk=1 << 20):
    """R reads of m bases, one base per byte at d_pats[r * m:], packed on the device
    with torch ops into the format of bwtgpu_search_packed: (m+15)//16 words per
    read, base k at bits 2 (k % 16) of word k // 16.  Set-up code for benches and
    tests (the product packs inside its search kernel, or on the host)."""
    wpr = (m + 15) // 16
    out = torch.empty(R * wpr, dtype=torch.int32, device="cuda")
    shifts = (2 * torch.arange(16, device="cuda", dtype=torch.int64)).view(1, 1, 16)
    for r0 in range(0, R, chunk):
        r1 = min(R, r0 + chunk)
        b = d_pats[r0 * m:r1 * m].view(r1 - r0, m).to(torch.int64) & 3
        if wpr * 16 != m:
            b = torch.nn.functional.pad(b, (0, wpr * 16 - m))
        w = (b.view(r1 - r0, wpr, 16) << shifts).sum(dim=2)
        w = torch.where(w >= (1 << 31), w - (1 << 32), w).to(torch.int32)
        out[r0 * wpr:r1 * wpr] = w.reshape(-1)
    return out


def pack_bw2(d_bw, rows):
    """BWT symbols one per byte -> four per byte (bwtgpu_pack_bw2)."""
    out = torch.empty((rows + 3) // 4, dtype=torch.uint8, device="cuda")
    check(_lib.load().bwtgpu_pack_bw2(_ptr(d_bw), rows, _ptr(out), _stream()))
    return out


def build_index_from_shards(d_text, n, parts):
    """make_fmi's index from what every shard contributed (bwt_amd.dist.ShardedParts)."""
    k = len(parts.rows_of)
    begin = (C.c_uint64 * k)(*[b for b, _ in parts.rows_of])
    rows = (C.c_uint64 * k)(*[r for _, r in parts.rows_of])
    offs = (C.c_uint64 * k)(*parts.bw2_offsets)
    return GpuIndex(_lib.load().bwtgpu_index_build_from_shards(_ptr(d_text), n, parts.endloc, k, begin, rows, offs,
                                                               _ptr(parts.bw2), _ptr(parts.samples), _stream()))


def build_index_sharded(d_text, n, group=None, want_sa=True):
    """make_fmi with the suffix array sorted in shards, one per rank of `group`:
    each rank sorts its bucket range, the ranks all-gather the index products of
    their rows (2-bit BWT symbols + every 8th row, 0.75 bytes per base:
    bwt_amd.dist.build_index_parts_sharded) and every rank assembles its own
    copy of the index.  With want_sa the rows themselves (4 bytes per base) are
    all-gathered afterwards and attached, so that searches get the position
    records as on an index built whole.  Texts whose shards hold long repeats are
    built whole on every rank instead.  Returns (index, "sharded" | "whole")."""
    import torch.distributed as tdist

    from . import dist as D

    world = tdist.get_world_size(group)
    rank = tdist.get_rank(group)
    counts = sa_bucket_counts(d_text, n)
    ranges = D.plan_bucket_ranges(counts, world)
    mine = D.shard_rows(counts, ranges)[rank][1]
    ws = torch.empty(int(_lib.load().bwtgpu_sa_workspace_bytes(n)), dtype=torch.uint8, device="cuda")
    rows_buf = torch.empty(max(mine, 1), dtype=torch.int32, device="cuda")
    bw_buf = torch.empty(max(mine, 1), dtype=torch.uint8, device="cuda")

    def build_range(lo, hi, rows_out, bw_out):
        try:
            return sa_build_range(d_text, n, lo, hi, rows_out, bw_out, ws)
        except BwtGpuError as e:
            if getattr(e, "code", 0) == -6:
                raise D.NeedsWholeText(str(e))
            raise

    try:
        parts = D.build_index_parts_sharded(n, counts, build_range, pack_bw2, rows_buf, bw_buf, group)
    except D.NeedsWholeText:
        del rows_buf, bw_buf, ws
        return build_index_dev(d_text, n), "whole"
    del ws, bw_buf
    ix = build_index_from_shards(d_text, n, parts)
    if want_sa and ix.keeps_suffix_array_by_default:
        sa = torch.empty(n + 1, dtype=torch.int32, device="cuda")
        D.gather_suffix_array(n, parts.rows_of, rows_buf, sa, group)
        check(_lib.load().bwtgpu_index_attach_sa(ix.h, _ptr(d_text), _ptr(sa), _stream()))
    return ix, "sharded"


def dp_align_batch(d_s1, d_off1, d_s2, d_off2, R, mode, max_len1, max_len2):
    """nw_fast (mode 0) / sw_fast (mode 1) for R pairs (see bwtgpu_dp_align_batch).
    Returns (ret, ops, nops): ops[off1[r] + off2[r] ...] holds pair r's unit operations."""
    ret = torch.empty(R, dtype=torch.int32, device="cuda")
    nops = torch.empty(R, dtype=torch.int32, device="cuda")
    ops = torch.zeros(d_s1.numel() + d_s2.numel() + 1, dtype=torch.uint8, device="cuda")
    check(_lib.load().bwtgpu_dp_align_batch(_ptr(d_s1), _ptr(d_off1), _ptr(d_s2), _ptr(d_off2), R, mode, max_len1,
                                            max_len2, _ptr(ret), _ptr(ops), _ptr(nops), _stream()))
    return ret, ops, nops


def align_reads(ix, d_text, d_reads, d_offsets, R, anchor_len, max_len):
    """align_read_anchored for R reads (see bwtgpu_align_reads).  Returns (pos, ops, nops):
    ops[5 * offsets[r] + 64 * r ...] holds read r's unit operations, first pushed first."""
    pos = torch.empty(R, dtype=torch.int32, device="cuda")
    nops = torch.empty(R, dtype=torch.int32, device="cuda")
    ops = torch.zeros(5 * d_reads.numel() + 64 * R + 64, dtype=torch.uint8, device="cuda")
    check(_lib.load().bwtgpu_align_reads(ix.h, _ptr(d_text), _ptr(d_reads), _ptr(d_offsets), R, anchor_len,
                                         max_len, _ptr(pos), _ptr(ops), _ptr(nops), _stream()))
    return pos, ops, nops


def sa_last_stats():
    st = (C.c_uint64 * 8)()
    check(_lib.load().bwtgpu_sa_last_stats(st))
    return [int(x) for x in st]


def bwt_from_sa(d_text, d_sa, n):
    lib = _lib.load()
    nb = (n + 3) // 4
    bwt = torch.zeros((nb + 15) // 16 * 16 + 16, dtype=torch.uint8, device="cuda")
    endloc = torch.zeros(1, dtype=torch.int32, device="cuda")
    check(lib.bwtgpu_bwt_from_sa(_ptr(d_text), _ptr(d_sa), n, _ptr(bwt), _ptr(endloc), _stream()))
    return bwt[:nb], int(endloc.item()) & 0xFFFFFFFF


def gen_genome(n, seed):
    lib = _lib.load()
    # padded so kernels may read whole 16-byte vectors past the last base
    t = torch.zeros(n // 4 + 1 + 64, dtype=torch.uint8, device="cuda")
    check(lib.bwtgpu_gen_genome(_ptr(t), n, seed, _stream()))
    return t


def gen_reads(d_text, n, R, m, seed, want_offsets=True, first=0):
    """Reads [first, first + R) of the read set `seed` names (1 % random, the rest
    cut from the text at the offsets returned)."""
    lib = _lib.load()
    reads = torch.empty(R * m + 64, dtype=torch.uint8, device="cuda")
    offs = torch.empty(R, dtype=torch.int32, device="cuda") if want_offsets else None
    check(lib.bwtgpu_gen_reads_slice(_ptr(d_text), n, _ptr(reads), first, R, m, seed, _ptr(offs), _stream()))
    return reads, offs


def gather_rate(gib=4, loads=1 << 28):
    """Random 64-byte lines per second over a buffer of `gib` GiB (bwtgpu_diag_gather_rate)."""
    buf = torch.ones(int(gib * (1 << 30)) // 4, dtype=torch.int32, device="cuda")
    out = C.c_double()
    check(_lib.load().bwtgpu_diag_gather_rate(_ptr(buf), buf.numel() * 4, loads, C.byref(out), _stream()))
    del buf
    return float(out.value)
