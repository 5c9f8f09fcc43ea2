"""An index loaded from its stored parts (BWT + sampled rows, fileio.c:24-65)
recomputes the suffix array and the text on its first search and then holds the
same search-only arrays as one built from the text: same arrays, same answers."""
import ctypes as C

import numpy as np
import pytest
import torch

from bwt_amd import _lib, synth
from helpers import GENOME_SEED, align_cases

pytestmark = pytest.mark.gpu


def _texts(oracle):
    yield "random 13", 13, oracle.gen_genome(13, GENOME_SEED)
    yield "random 1000", 1000, oracle.gen_genome(1000, GENOME_SEED + 1)
    yield "random 70001", 70001, oracle.gen_genome(70001, GENOME_SEED + 2)
    yield "random 3 M", 3_000_000, oracle.gen_genome(3_000_000, GENOME_SEED + 3)
    yield "tandem 7", 50_000, oracle.gen_tandem(50_000, 7, 11, 0, 0)
    yield "tandem 1000 mutated", 200_000, oracle.gen_tandem(200_000, 1000, 11, 5000, 3)
    yield "one symbol", 5000, np.zeros(5000 // 4 + 1, np.uint8)


def _d(text):
    return torch.from_numpy(np.concatenate([text, np.zeros(64, np.uint8)])).cuda()


def test_loaded_index_recovers_suffix_array_and_text(oracle, gpu):
    for name, n, text in _texts(oracle):
        fo = oracle.make_fmi(text, n)
        sa = oracle.sa_doubling(text, n) if n <= 200_000 else None
        ix = gpu.index_from_parts(fo.bwt, n, fo.endloc, fo.idxs)
        assert ix.kmer_table == (0, 0), name  # nothing is made before the first search
        ix.prepare_search()
        k, w = ix.kmer_table
        built = gpu.build_index(text, n)
        built.prepare_search()
        assert (k, w) == built.kmer_table, name  # two-word table of the same size: the suffix array is there
        assert ix.device_bytes == built.device_bytes, name
        # the recovered rows and text against the text itself: certificate of the whole index
        rep = ix.certify(_d(text))
        assert all(v == 0 for v in rep.values()), (name, rep)
        rows = torch.arange(0, n + 1, dtype=torch.int64, device="cuda").to(torch.int32)
        got = gpu.to_numpy_u32(ix.locate_rows(rows))
        want = gpu.to_numpy_u32(built.locate_rows(rows))
        assert np.array_equal(got, want), name
        if sa is not None:
            assert np.array_equal(got, np.asarray(sa, np.uint32)), name
        ix.close()
        built.close()


@pytest.mark.parametrize("m", [1, 12, 50, 100, 250])
def test_loaded_index_searches_like_a_built_one_and_the_oracle(oracle, gpu, m):
    for name, n, text in _texts(oracle):
        if n < m:
            continue
        fo = oracle.make_fmi(text, n)
        R = 3000
        reads, _ = oracle.gen_reads(text, n, R, m, 77 + m)
        rng = np.random.default_rng(m)
        for r in range(0, R, 3):  # a substitution: the search ends early or somewhere else
            p = int(rng.integers(0, m))
            reads[r, p] = (reads[r, p] + 1 + int(rng.integers(0, 3))) & 3
        bases = synth.unpack(text, n)
        for r in range(1, R, 50):  # reads that hang over the start of the text
            k = int(rng.integers(1, m + 1))
            reads[r, m - k:] = bases[:k]
        sp_o, ep_o = oracle.search_batch(fo, reads, m, m, R)
        d = torch.from_numpy(reads.reshape(-1)).cuda()
        loaded = gpu.index_from_parts(fo.bwt, n, fo.endloc, fo.idxs)
        lean = gpu.index_from_parts(fo.bwt, n, fo.endloc, fo.idxs, lean=True)
        outs = []
        for ix in (loaded, lean):
            sp, ep, lr = ix.search(d, m, R, want_locrow=True)
            assert np.array_equal(gpu.to_numpy_u32(sp), sp_o), (name, m)
            assert np.array_equal(gpu.to_numpy_u32(ep), ep_o), (name, m)
            outs.append((lr, ix.locate_rows(lr)))
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), (name, m)
        assert loaded.kmer_table[1] == (2 if loaded.kmer_table[0] else 0)
        assert lean.kmer_table[1] in (0, 1)
        loaded.close()
        lean.close()


def test_keep_lean_and_compact_cancel_the_recovery(oracle, gpu):
    n = 100_000
    text = oracle.gen_genome(n, GENOME_SEED + 9)
    fo = oracle.make_fmi(text, n)
    lean = gpu.index_from_parts(fo.bwt, n, fo.endloc, fo.idxs, lean=True)
    lean.prepare_search()
    assert lean.kmer_table[1] == 1
    squeezed = gpu.index_from_parts(fo.bwt, n, fo.endloc, fo.idxs)
    squeezed.compact()
    squeezed.prepare_search()
    assert squeezed.kmer_table == (0, 0)
    assert squeezed.device_bytes < lean.device_bytes
    # keep_lean after the arrays are there changes nothing
    full = gpu.index_from_parts(fo.bwt, n, fo.endloc, fo.idxs)
    full.prepare_search()
    before = full.device_bytes
    _lib.check(_lib.load().bwtgpu_index_keep_lean(full.h))
    assert full.device_bytes == before and full.kmer_table[1] == 2


def test_loaded_index_mms_and_aligner(oracle, gpu):
    from test_align import _gpu_align
    from test_gpu_mms import _mms
    from helpers import mms_patterns

    n = 300_000
    text = oracle.gen_genome(n, 31)
    fo = oracle.make_fmi(text, n)
    loaded = gpu.index_from_parts(fo.bwt, n, fo.endloc, fo.idxs)
    pats = mms_patterns(oracle, text, n, 600, 70, 5)
    got, sp, ep = _mms(gpu, loaded, pats)
    want = [oracle.mms(fo, pats[r]) for r in range(len(pats))]
    assert [(int(a), int(b), int(c)) for a, b, c in zip(got, sp, ep)] == [tuple(int(x) for x in w) for w in want]
    reads = align_cases(synth.unpack(text, n), n, 600, 12, lo=40, hi=160)
    wanta = [oracle.align_read_anchored(fo, text, rd, 12) for rd in reads]
    gota = _gpu_align(gpu, loaded, text, reads, 12, 170)
    for rd, g, w in zip(reads, gota, wanta):
        assert g == w, len(rd)
