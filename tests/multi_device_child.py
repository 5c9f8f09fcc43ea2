"""Child process of test_gpu_multi_device.py: the reference-API layer with several device slots
(BWT_MI355X_DEVICES set by the parent before this process loads the library).  Every batch form
that is shared over the slots is compared with the oracle; prints one line "OK slots=<n> ..."."""
import ctypes as C
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from bwt_amd import _lib, synth  # noqa: E402
from oracle import load_oracle  # noqa: E402

vp = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731


def main():
    n, R, m = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    orc = load_oracle()
    L = C.CDLL(_lib.LIB_PATH)
    L.make_fmi.restype = C.c_void_p
    L.make_fmi.argtypes = [C.c_void_p, C.c_int]
    L.read_index.restype = C.c_void_p
    L.read_index.argtypes = [C.c_char_p, C.c_void_p]
    L.write_index.argtypes = [C.c_void_p, C.c_void_p]
    L.destroy_fmi.argtypes = [C.c_void_p]
    for name in ("fmi_search_batch", "fmi_reverse_search_batch", "fmi_locate_batch", "fmi_search_locate_batch",
                 "fmi_search_batch_packed", "fmi_unc_sa_batch", "fmi_mms_batch", "fmi_pack_reads"):
        getattr(L, name).argtypes = None
    libc = C.CDLL(None)
    libc.fopen.restype = C.c_void_p
    libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
    libc.fclose.argtypes = [C.c_void_p]
    slots = L.fmi_device_slots()

    text = orc.gen_genome(n, synth.GENOME_SEED + 21)
    fo = orc.make_fmi(text, n)
    reads, offs = orc.gen_reads(text, n, R, m, synth.READS_SEED + 21)
    reads = np.ascontiguousarray(reads)
    sp_o, ep_o = orc.search_batch(fo, reads, m, m, R)
    found = ep_o > sp_o
    pos_o, _ = orc.locate_rows(fo, np.where(found, sp_o, 0).astype(np.uint32))

    def check(fmi, tag):
        fmi = C.c_void_p(fmi)
        cl = lambda x: C.c_long(x)  # noqa: E731
        sp, ep, pos, res = (np.full(R, -7, np.int32) for _ in range(4))
        assert L.fmi_search_batch(fmi, vp(reads), m, cl(m), cl(R), vp(sp), vp(ep)) == 0, tag
        assert (sp.view(np.uint32) == sp_o).all() and (ep.view(np.uint32) == ep_o).all(), tag + ": search"
        sp[:] = ep[:] = -7
        assert L.fmi_search_locate_batch(fmi, vp(reads), m, cl(m), cl(R), vp(sp), vp(ep), vp(pos)) == 0, tag
        assert (sp.view(np.uint32) == sp_o).all() and (ep.view(np.uint32) == ep_o).all(), tag + ": search_locate"
        assert (pos.view(np.uint32)[found] == pos_o[found]).all(), tag + ": positions"
        pos2 = np.full(R, -7, np.int32)
        assert L.fmi_locate_batch(fmi, vp(reads), m, cl(m), cl(R), vp(pos2)) == 0, tag
        assert (pos2 == pos).all(), tag + ": locate"
        assert L.fmi_reverse_search_batch(fmi, vp(reads), m, cl(m), cl(R), vp(res)) == 0, tag
        want = np.array([orc.reverse_search(fo, reads[r]) for r in range(0, R, 211)], np.int32)
        assert (res[::211] == want).all(), tag + ": reverse_search"
        words = np.zeros((R, (m + 15) // 16), np.uint32)
        L.fmi_pack_reads(vp(reads), m, cl(m), cl(R), vp(words))
        sp[:] = ep[:] = pos2[:] = -7
        assert L.fmi_search_batch_packed(fmi, vp(words), m, cl(R), vp(sp), vp(ep), vp(pos2)) == 0, tag
        assert (sp.view(np.uint32) == sp_o).all() and (ep.view(np.uint32) == ep_o).all(), tag + ": packed"
        assert (pos2[found] == pos[found]).all(), tag + ": packed positions"
        rows = np.where(found, sp_o, 0).astype(np.int32)
        pos3 = np.full(R, -7, np.int32)
        assert L.fmi_unc_sa_batch(fmi, vp(rows), cl(R), vp(pos3)) == 0, tag
        assert (pos3.view(np.uint32) == pos_o).all(), tag + ": unc_sa"
        edited = reads.copy()
        edited[::3, m // 2] = 5  # an N in every third read
        matched, msp, mep = (np.full(R, -7, np.int32) for _ in range(3))
        assert L.fmi_mms_batch(fmi, vp(edited), m, cl(m), cl(R), vp(matched), vp(msp), vp(mep)) == 0, tag
        for r in range(0, R, 173):
            got = (int(matched[r]), int(msp[r]), int(mep[r]))
            assert got == orc.mms(fo, edited[r]), tag + f": mms of read {r}"
        # overlapping reads at stride 1 (fmitest.c:54-65)
        R1 = min(R, n - m)
        flat = synth.unpack(text, n).astype(np.uint8)
        sp1, ep1 = (np.full(R1, -7, np.int32) for _ in range(2))
        assert L.fmi_search_batch(fmi, vp(flat), m, cl(1), cl(R1), vp(sp1), vp(ep1)) == 0, tag
        pick = np.arange(0, R1, 389)
        win = np.stack([flat[j:j + m] for j in pick])
        a, b = orc.search_batch(fo, win, m, m, len(pick))
        assert (sp1.view(np.uint32)[pick] == a).all() and (ep1.view(np.uint32)[pick] == b).all(), tag + ": stride 1"

    fmi = L.make_fmi(vp(text), n)
    assert fmi, "make_fmi failed"
    check(fmi, "built")
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "x.idx").encode()
        f = libc.fopen(path, b"wb")
        L.write_index(C.c_void_p(fmi), C.c_void_p(f))
        libc.fclose(C.c_void_p(f))
        f = libc.fopen(path, b"rb")
        loaded = L.read_index(None, C.c_void_p(f))
        libc.fclose(C.c_void_p(f))
    assert loaded, "read_index failed"
    check(loaded, "loaded")
    check(fmi, "built, again")  # replicas now exist
    L.destroy_fmi(C.c_void_p(loaded))
    L.destroy_fmi(C.c_void_p(fmi))
    print(f"OK slots={slots} n={n} reads={R}x{m}")


if __name__ == "__main__":
    main()
