"""mms (seqindex.c:314-351) and the search_reads anchor loop
(search_reads.c:102-160) on the GPU against the golden vectors and the oracle."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
import torch

from bwt_amd import _lib
from helpers import GOLDEN, GENOME_SEED, ROOT, mms_patterns

pytestmark = pytest.mark.gpu


def _mms(gpu, ix, pats):
    R, m = pats.shape
    d = torch.from_numpy(np.ascontiguousarray(pats).reshape(-1)).cuda()
    got = torch.empty(R, dtype=torch.int32, device="cuda")
    sp = torch.empty(R, dtype=torch.int32, device="cuda")
    ep = torch.empty(R, dtype=torch.int32, device="cuda")
    _lib.check(_lib.load().bwtgpu_mms(ix.h, d.data_ptr(), m, m, R, got.data_ptr(), sp.data_ptr(),
                                      ep.data_ptr(), None))
    return got.cpu().numpy(), gpu.to_numpy_u32(sp), gpu.to_numpy_u32(ep)


@pytest.mark.parametrize("rec", [r for r in GOLDEN["random"] if "mms" in r], ids=lambda r: f"n{r['n']}-m{r['mms_m']}")
def test_mms_golden(oracle, gpu, rec):
    n = rec["n"]
    text = oracle.gen_genome(n, GENOME_SEED)
    ix = gpu.build_index(text, n)
    pats = mms_patterns(oracle, text, n, 300, rec["mms_m"], 17)
    got, sp, ep = _mms(gpu, ix, pats)
    assert [[int(a), int(b), int(c)] for a, b, c in zip(got, sp, ep)] == rec["mms"]


def test_mms_mismatching_reads_and_scalar_call(oracle, gpu):
    """reads with substitutions stop early; the reference-API scalar mms agrees"""
    n, R, m = 300000, 3000, 60
    text = oracle.gen_genome(n, 21)
    fo = oracle.make_fmi(text, n)
    ix = gpu.build_index(text, n)
    reads, _ = oracle.gen_reads(text, n, R, m, 22)
    from bwt_amd import synth
    bases = synth.unpack(text, n)
    rng = np.random.default_rng(3)
    for r in range(R):
        for _ in range(rng.integers(0, 3)):
            p = rng.integers(0, m)
            reads[r, p] = (reads[r, p] + 1 + rng.integers(0, 3)) & 3
        if r % 5 == 0:
            reads[r, rng.integers(0, m)] = 5
        if r % 7 == 0:   # an N among the last bases: no table start for this read
            reads[r, m - 1 - rng.integers(0, 12)] = 5
        if r % 11 == 0:  # the read ends with the first bases of the text: the match runs into its start
            k = int(rng.integers(1, m))
            reads[r, m - k:] = bases[:k]
    want = [oracle.mms(fo, reads[r]) for r in range(R)]
    # index with SA/ISA/text and table, index from stored parts (table only), bare rank blocks
    lean = gpu.index_from_parts(fo.bwt, n, fo.endloc, fo.idxs, lean=True)
    for name, index in (("full", ix), ("from parts", lean)):
        got, sp, ep = _mms(gpu, index, reads)
        assert [(int(a), int(b), int(c)) for a, b, c in zip(got, sp, ep)] == want, name
    lean.compact()
    got, sp, ep = _mms(gpu, lean, reads)
    assert [(int(a), int(b), int(c)) for a, b, c in zip(got, sp, ep)] == want
    hl = C.CDLL(_lib.LIB_PATH)
    hl.make_fmi.restype = C.c_void_p
    hl.make_fmi.argtypes = [C.c_void_p, C.c_int]
    hl.mms.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    hl.destroy_fmi.argtypes = [C.c_void_p]
    fmi = hl.make_fmi(text.ctypes.data_as(C.c_void_p), n)
    for r in range(0, R, 211):
        a, b = C.c_int(), C.c_int()
        k = hl.mms(fmi, reads[r].ctypes.data_as(C.c_void_p), m, C.byref(a), C.byref(b))
        assert (k, a.value, b.value) == oracle.mms(fo, reads[r])
    hl.destroy_fmi(fmi)


def _make_reads_file(oracle, text, n, count, seed):
    """reads of varying length with substitutions, some Ns, some junk lines"""
    rng = np.random.default_rng(seed)
    lines, enc = [], []
    from bwt_amd import synth
    bases = synth.unpack(text, n)
    for r in range(count):
        L = int(rng.integers(15, 140))
        j = int(rng.integers(0, n - L))
        b = bases[j:j + L].copy()
        for _ in range(rng.integers(0, 4)):
            p = rng.integers(0, L)
            b[p] = (b[p] + 1) & 3
        s = "".join("ACGT"[x] for x in b)
        if r % 9 == 0:
            p = int(rng.integers(0, L))
            s = s[:p] + "N" + s[p + 1:]
        if r % 50 == 49:
            s = "".join("ACGT"[x] for x in rng.integers(0, 4, L))
        lines.append(s)
        enc.append(np.array([{"A": 0, "C": 1, "G": 2, "T": 3}.get(ch, 5) for ch in s], np.uint8))
    return lines, enc


def test_search_reads_kernel_and_driver(oracle, gpu, tmp_path):
    n = 400000
    text = oracle.gen_genome(n, 31)
    fo = oracle.make_fmi(text, n)
    lines, enc = _make_reads_file(oracle, text, n, 400, 7)
    want = np.array([oracle.search_reads_one(fo, e, 20) for e in enc], np.uint32)
    # device entry point
    ix = gpu.build_index(text, n)
    offs = np.zeros(len(enc) + 1, np.uint64)
    offs[1:] = np.cumsum([len(e) for e in enc])
    flat = np.concatenate(enc)
    rev = np.concatenate([e[::-1] for e in enc])
    d_f, d_r = torch.from_numpy(flat).cuda(), torch.from_numpy(rev).cuda()
    d_o = torch.from_numpy(offs.view(np.int64)).cuda()
    out = torch.empty(len(enc) * 4, dtype=torch.int32, device="cuda")
    _lib.check(_lib.load().bwtgpu_search_reads(ix.h, d_f.data_ptr(), d_r.data_ptr(), d_o.data_ptr(), len(enc),
                                               20, out.data_ptr(), None))
    np.testing.assert_array_equal(gpu.to_numpy_u32(out).reshape(-1, 4), want)
    assert (want[:, 0] > 0).mean() > 0.5  # most reads do anchor
    # the driver: ASCII genome -> build_index -> search_reads, same report lines
    from bwt_amd import synth
    seq = tmp_path / "g.txt"
    n4 = n - n % 4  # the packer zeroes the last len%4 bases; keep them out of the way
    ascii_genome = "".join("ACGT"[x] for x in synth.unpack(text, n)[:n4])
    seq.write_text(ascii_genome)
    idx, rd = tmp_path / "g.idx", tmp_path / "r.txt"
    rd.write_text("\n".join(lines) + "\n")
    drv = os.path.join(ROOT, "bwt_amd", "drivers")
    assert subprocess.run([os.path.join(drv, "build_index"), str(seq), str(idx)], capture_output=True,
                          timeout=120).returncode == 0
    res = subprocess.run([os.path.join(drv, "search_reads"), str(seq), str(idx), str(rd)], capture_output=True,
                         text=True, timeout=120)
    assert res.returncode == 0, res.stderr
    assert res.stdout.startswith("Beginning alignment\n")
    # expected report from the oracle on the same (truncated) genome
    t4 = synth.pack(synth.unpack(text, n)[:n4])
    f4 = oracle.make_fmi(t4, n4)
    expect = []
    for r, e in enumerate(enc):
        o = oracle.search_reads_one(f4, e, 20).astype(np.int64)
        fpos, bpos = (int(np.int32(np.uint32(x))) for x in o[2:])
        if o[0] and o[1] and abs(fpos - bpos) < 10000:
            expect.append(f"\nRead {r}: Aligned both forward ({o[0]}) and backward ({o[1]})\n"
                          f"At locations {fpos} and {bpos} respectively\n{lines[r]}\n")
    assert res.stdout == "Beginning alignment\n" + "".join(expect)
