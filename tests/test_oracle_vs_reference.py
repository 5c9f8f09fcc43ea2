"""Where oracle/_ref has been built: the oracle against the compiled reference
directly, on inputs beyond the committed fixtures."""
import numpy as np
import pytest


@pytest.mark.parametrize("n,seed", [(1, 1), (2, 2), (5, 3), (31, 4), (32, 5), (33, 6), (4097, 7), (250001, 8)])
def test_fmi_fields(oracle, reference, n, seed):
    text = oracle.gen_genome(n, seed)
    sa = reference.histsort(text, n)
    np.testing.assert_array_equal(oracle.histsort(text, n), sa)
    np.testing.assert_array_equal(oracle.sa_doubling(text, n), sa)
    np.testing.assert_array_equal(reference.csuff_arr(text, n), sa)
    fo, fr = oracle.make_fmi(text, n), reference.make_fmi(text, n)
    assert (fo.endloc, fo.C) == (fr.endloc, fr.C)
    np.testing.assert_array_equal(fo.bwt, fr.bwt)
    np.testing.assert_array_equal(fo.idxs, fr.idxs)
    np.testing.assert_array_equal(fo.rank_rows, fr.rank_rows)
    for i in range(0, n + 1, max(1, n // 200)):
        assert oracle.lf(fo, i) == reference.lf(fr, i)
        assert oracle.unc_sa(fo, i) == reference.unc_sa(fr, i) == sa[i]
        for c in range(4):
            assert oracle.rank(fo, c, i) == reference.rank(fr, c, i)


def test_index_files_are_byte_identical(oracle, reference, tmp_path):
    n = 100003
    text = oracle.gen_genome(n, 9)
    a, b = str(tmp_path / "a.idx"), str(tmp_path / "b.idx")
    oracle.write_index(oracle.make_fmi(text, n), a)
    reference.write_index(reference.make_fmi(text, n), b)
    assert open(a, "rb").read() == open(b, "rb").read()
    # each side reads the other's file
    fo, fr = oracle.read_index(b), reference.read_index(a)
    np.testing.assert_array_equal(fo.rank_rows, fr.rank_rows)


def test_search_and_mms(oracle, reference):
    n, R, m = 300000, 3000, 36
    text = oracle.gen_genome(n, 10)
    fo, fr = oracle.make_fmi(text, n), reference.make_fmi(text, n)
    reads, _ = oracle.gen_reads(text, n, R, m, 11)
    so, eo = oracle.search_batch(fo, reads, m, m, R)
    sr, er = reference.search_batch(fr, reads, m, m, R)
    np.testing.assert_array_equal(so, sr)
    np.testing.assert_array_equal(eo, er)
    rng = np.random.default_rng(1)
    for r in range(400):
        p = reads[r].copy()
        p[rng.integers(0, m, 2)] = 5
        assert oracle.mms(fo, p) == reference.mms(fr, p)
        assert oracle.reverse_search(fo, reads[r]) == reference.reverse_search(fr, reads[r])
