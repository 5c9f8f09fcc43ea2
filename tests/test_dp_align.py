"""nw_fast / sw_fast (smw.c:26-200): the oracle's restatement against the
compiled reference, and the GPU batch kernel against the oracle."""
import hashlib

import numpy as np
import pytest

from helpers import GOLDEN, dp_cases, dp_digest


@pytest.mark.parametrize("mode", [0, 1], ids=["nw_fast", "sw_fast"])
def test_oracle_dp_equals_the_reference(oracle, reference, mode):
    for s1, g in dp_cases(11 + mode, 1500, 70, 80):
        assert oracle.dp_align(s1, g, mode) == reference.dp_align(s1, g, mode)


@pytest.mark.parametrize("mode", [0, 1], ids=["nw_fast", "sw_fast"])
def test_oracle_dp_equals_the_golden_vectors(oracle, mode):
    """results the reference gave when the fixtures were made (tests/golden/make_golden.py)"""
    rec = GOLDEN["dp"][mode]
    cases = dp_cases(rec["seed"], rec["count"], rec["max1"], rec["max2"])
    got = [oracle.dp_align(s1, g, mode) for s1, g in cases]
    assert [[r, o.decode()] for r, o in got[:len(rec["first"])]] == rec["first"]
    assert dp_digest(got) == rec["sha256"]


def test_oracle_dp_known_answers(oracle):
    """small cases worked by hand from smw.c's scores (0 / -6, gap -3 per base, -5 to open)"""
    acgt = np.array([0, 1, 2, 3], np.uint8)
    assert oracle.dp_align(acgt, acgt, 0) == (3, b"MMMM")
    assert oracle.dp_align(acgt, acgt, 1) == (0, b"MMMM")
    # genome has two more bases after the read: nw_fast stops at the best cell of the last row
    assert oracle.dp_align(acgt, np.array([0, 1, 2, 3, 0, 0], np.uint8), 0) == (3, b"MMMM")
    # sw_fast always ends in the corner: the two extra bases are deletions, pushed first
    assert oracle.dp_align(acgt, np.array([0, 1, 2, 3, 0, 0], np.uint8), 1) == (0, b"DDMMMM")
    # N agrees with anything
    assert oracle.dp_align(np.array([0, 5, 2], np.uint8), np.array([0, 3, 2], np.uint8), 0) == (2, b"MMM")
    assert oracle.dp_align(np.zeros(0, np.uint8), acgt, 0) == (0, b"")
    assert oracle.dp_align(np.zeros(0, np.uint8), acgt, 1) == (0, b"DDDD")
    assert oracle.dp_align(acgt, np.zeros(0, np.uint8), 0) == (-1, b"IIII")


def _run_gpu(gpu, cases, mode, max1, max2):
    import torch

    off1 = np.zeros(len(cases) + 1, np.int64)
    off2 = np.zeros(len(cases) + 1, np.int64)
    off1[1:] = np.cumsum([len(a) for a, _ in cases])
    off2[1:] = np.cumsum([len(b) for _, b in cases])
    s1 = np.concatenate([a for a, _ in cases] + [np.zeros(1, np.uint8)])
    s2 = np.concatenate([b for _, b in cases] + [np.zeros(1, np.uint8)])
    ret, ops, nops = gpu.dp_align_batch(torch.from_numpy(s1).cuda(), torch.from_numpy(off1).cuda(),
                                        torch.from_numpy(s2).cuda(), torch.from_numpy(off2).cuda(), len(cases), mode,
                                        max1, max2)
    ret, ops, nops = ret.cpu().numpy(), ops.cpu().numpy(), gpu.to_numpy_u32(nops)
    got = []
    for r in range(len(cases)):
        o = int(off1[r] + off2[r])
        got.append((int(ret[r]), None if nops[r] == 0xFFFFFFFF else ops[o: o + int(nops[r])].tobytes()))
    return got


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [0, 1], ids=["nw_fast", "sw_fast"])
@pytest.mark.parametrize("max1,max2,count", [(20, 30, 3000), (63, 64, 3000), (64, 65, 2000), (100, 110, 3000),
                                             (150, 200, 1500), (250, 260, 600)])
def test_gpu_dp_equals_the_oracle(oracle, gpu, mode, max1, max2, count):
    cases = dp_cases(1000 * mode + max1, count, max1, max2)
    got = _run_gpu(gpu, cases, mode, max1, max2)
    for (s1, g), have in zip(cases, got):
        assert have == oracle.dp_align(s1, g, mode), (len(s1), len(g))


@pytest.mark.gpu
def test_gpu_dp_refuses_what_it_was_not_sized_for(gpu):
    cases = [(np.zeros(30, np.uint8), np.zeros(10, np.uint8)), (np.zeros(5, np.uint8), np.zeros(5, np.uint8))]
    got = _run_gpu(gpu, cases, 0, 20, 20)
    assert got[0] == (0, None) and got[1] == (4, b"MMMMM")
    from bwt_amd import _lib
    with pytest.raises(_lib.BwtGpuError):
        _run_gpu(gpu, cases, 0, 4000, 4000)  # 64 KiB of LDS do not hold that
