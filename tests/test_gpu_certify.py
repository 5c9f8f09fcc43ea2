"""The on-device certificate of a suffix array / index (bwtgpu_sa_certify,
bwtgpu_index_certify): zero violations on what the build produces -- checked
here against the oracle's own checker and arrays at sizes the oracle handles --
and violations reported for every kind of damage."""
import numpy as np
import pytest
import torch

from bwt_amd import synth

pytestmark = pytest.mark.gpu


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int32 if a.dtype == np.uint32 else a.dtype)).cuda()


@pytest.mark.parametrize("n", [1, 2, 13, 64, 777, 100003, 1000000])
def test_correct_array_and_index_certify_clean(oracle, gpu, n):
    text = oracle.gen_genome(n, synth.GENOME_SEED + n)
    sa = oracle.histsort(text, n)
    assert oracle.check_sa(text, n, sa) == 0
    d_text = _dev(np.concatenate([text, np.zeros(64, np.uint8)]))
    assert gpu.sa_certify(d_text, n, _dev(sa)) == (0, 0, 0)
    ix = gpu.build_index(text, n)
    rep = ix.certify(d_text)
    assert all(v == 0 for v in rep.values()), rep
    ix.close()


@pytest.mark.parametrize("kind", ["tandem7", "tandem1000mut", "homopolymer"])
def test_repeats_certify_clean(oracle, gpu, kind):
    n = 200000
    if kind == "tandem7":
        text = oracle.gen_tandem(n, 7, 11)
    elif kind == "tandem1000mut":
        text = oracle.gen_tandem(n, 1000, 11, 50, 3)
    else:
        text = np.zeros(n // 4 + 1, np.uint8)
    d_text = _dev(np.concatenate([text, np.zeros(64, np.uint8)]))
    ix = gpu.build_index(text, n)
    rep = ix.certify(d_text)
    assert all(v == 0 for v in rep.values()), rep
    ix.close()


def test_damage_is_reported(oracle, gpu):
    n = 50000
    text = oracle.gen_genome(n, 5)
    sa = oracle.histsort(text, n)
    d_text = _dev(np.concatenate([text, np.zeros(64, np.uint8)]))
    # two neighbouring rows swapped: still a permutation, order broken
    bad = sa.copy()
    bad[[1000, 1001]] = bad[[1001, 1000]]
    r = gpu.sa_certify(d_text, n, _dev(bad))
    assert r[0] == 0 and r[1] == 0 and r[2] >= 1 and oracle.check_sa(text, n, bad) != 0
    # two distant rows swapped
    bad = sa.copy()
    bad[[7, 40000]] = bad[[40000, 7]]
    r = gpu.sa_certify(d_text, n, _dev(bad))
    assert r[1] == 0 and r[2] >= 1
    # a value twice (another one missing): not a permutation
    bad = sa.copy()
    bad[123] = bad[124]
    r = gpu.sa_certify(d_text, n, _dev(bad))
    assert r[1] >= 1 and oracle.check_sa(text, n, bad) != 0
    # a value out of range, and row 0 not the empty suffix
    bad = sa.copy()
    bad[500] = n + 7
    assert gpu.sa_certify(d_text, n, _dev(bad))[0] == 1
    bad = sa.copy()
    bad[[0, 1]] = bad[[1, 0]]
    assert gpu.sa_certify(d_text, n, _dev(bad))[0] >= 1
    # the right array for another text
    other = oracle.gen_genome(n, 6)
    d_other = _dev(np.concatenate([other, np.zeros(64, np.uint8)]))
    r = gpu.sa_certify(d_other, n, _dev(sa))
    assert r[0] == 0 and r[1] == 0 and r[2] > 100
    # an index certified against another text: order, BWT blocks and (almost surely) the $ row differ
    ix = gpu.build_index(text, n)
    rep = ix.certify(d_other)
    assert rep["order"] > 100 and rep["bwt_blocks"] > 100 and rep["samples"] == 0
    # an index without its suffix array cannot be certified
    ix.compact()
    with pytest.raises(gpu.BwtGpuError, match="does not keep"):
        ix.certify(d_text)
    ix.close()
