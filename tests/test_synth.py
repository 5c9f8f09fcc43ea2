"""The numpy input generators against the oracle's C generators."""
import numpy as np
import pytest

from bwt_amd import synth


@pytest.mark.parametrize("n", [1, 3, 4, 5, 31, 32, 33, 1000, 100003])
def test_genome(oracle, n):
    a = synth.gen_genome(n, 77)
    np.testing.assert_array_equal(a, oracle.gen_genome(n, 77))
    assert len(a) == n // 4 + 1
    bases = synth.unpack(a, n)
    np.testing.assert_array_equal(synth.pack(bases), a)  # tail bits are zero


def test_reads(oracle):
    n, R, m = 50000, 3000, 37
    t = synth.gen_genome(n)
    r1, o1 = synth.gen_reads(t, n, R, m, 5)
    r2, o2 = oracle.gen_reads(t, n, R, m, 5)
    np.testing.assert_array_equal(r1, r2)
    np.testing.assert_array_equal(o1, o2)
    frac = (o1 == 0xFFFFFFFF).mean()
    assert 0.002 < frac < 0.03


@pytest.mark.parametrize("period,mut", [(7, 0), (1000, 0), (1000, 97)])
def test_tandem(oracle, period, mut):
    n = 20011
    np.testing.assert_array_equal(synth.gen_tandem(n, period, 11, mut, 3), oracle.gen_tandem(n, period, 11, mut, 3))
