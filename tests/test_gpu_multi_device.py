"""Several GPUs behind the reference's C API: the index replicated, the reads of every batch form
partitioned into contiguous shares, one host thread and one GPU per share (fmitest.c:67-91 is the
reference's form of it: four worker threads, a quarter of the reads each).  A one-GPU machine tests
it by listing its GPU several times: every mention is a slot with its own replica of the index.
The device set is read once per process, so each case runs in a child process."""
import os
import subprocess
import sys

import pytest

from helpers import ROOT

CHILD = os.path.join(ROOT, "tests", "multi_device_child.py")


def _run(devices, n, R, m, extra_env=None):
    env = dict(os.environ)
    env.pop("BWT_MI355X_DEVICE", None)
    env["BWT_MI355X_DEVICES"] = devices
    env["BWT_MI355X_MIN_READS_PER_DEVICE"] = "2000"
    env.update(extra_env or {})
    out = subprocess.run([sys.executable, CHILD, str(n), str(R), str(m)], env=env, capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    return out.stdout.strip().splitlines()[-1]


@pytest.mark.gpu
@pytest.mark.parametrize("devices,slots,n,R,m", [("0,0,0", 3, 400_003, 50_001, 40), ("0,0,0,0,0,0,0,0", 8, 150_000, 40_000, 24),
                                                 ("0", 1, 100_000, 10_000, 30)])
def test_batches_shared_over_device_slots_equal_the_oracle(gpu, devices, slots, n, R, m):
    line = _run(devices, n, R, m)
    assert line.startswith(f"OK slots={slots} "), line


@pytest.mark.gpu
def test_shared_batches_on_lean_replicas(gpu):
    """BWT_MI355X_COMPACT: no suffix array travels, every slot searches on rank blocks."""
    line = _run("0,0", 120_000, 30_000, 33, {"BWT_MI355X_COMPACT": "1"})
    assert line.startswith("OK slots=2 "), line


def test_share_bounds_and_slot_list_need_no_gpu():
    """The arithmetic of the split, and the parsing of the device list (CPU)."""
    code = r'''
import ctypes as C, sys
sys.path.insert(0, %r)
from bwt_amd import _lib, dist
L = C.CDLL(_lib.LIB_PATH)
L.fmi_share_bounds.argtypes = [C.c_long, C.c_int, C.c_int, C.POINTER(C.c_long), C.POINTER(C.c_long)]
for count in (0, 1, 7, 8, 9, 1000, 10**7 + 11, 3 * 10**9 + 5):
    for shares in (1, 2, 3, 8, 16):
        prev = 0
        for g in range(shares):
            lo, hi = C.c_long(), C.c_long()
            L.fmi_share_bounds(count, shares, g, C.byref(lo), C.byref(hi))
            assert (lo.value, hi.value) == dist.slice_bounds(count, shares, g), (count, shares, g)
            assert lo.value == prev and hi.value - lo.value in (count // shares, count // shares + 1)
            prev = hi.value
        assert prev == count
print("slots", L.fmi_device_slots())
''' % ROOT
    for devices, want in (("0,0,1", 3), ("5", 1), ("0,1,2,3,4,5,6,7", 8)):
        env = dict(os.environ, BWT_MI355X_DEVICES=devices)
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stderr[-2000:]
        assert out.stdout.strip().endswith(f"slots {want}"), out.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("form", ["built", "built, searched", "loaded", "lean"])
def test_a_replica_answers_like_its_source(oracle, gpu, form):
    """bwtgpu_index_replicate (here onto the same GPU): the copy makes its own search-only arrays and gives
    the source's intervals, rows and positions; it outlives the source."""
    import numpy as np
    import torch

    from bwt_amd import synth

    n, R, m = 250_003, 20_000, 36
    text = oracle.gen_genome(n, synth.GENOME_SEED + 31)
    fo = oracle.make_fmi(text, n)
    if form.startswith("built"):
        src = gpu.build_index(text, n)
    else:
        src = gpu.index_from_parts(fo.bwt, n, fo.endloc, fo.idxs, lean=form == "lean")
    reads, _ = oracle.gen_reads(text, n, R, m, synth.READS_SEED + 31)
    d = torch.from_numpy(np.concatenate([reads.reshape(-1), np.zeros(64, np.uint8)])).cuda()
    if form == "built, searched":
        src.search(d, m, R)
    copy = src.replicate(0)
    assert (copy.n, copy.endloc, copy.C) == (src.n, src.endloc, src.C)
    want = [gpu.to_numpy_u32(x) for x in src.search_locate(d, m, R)]
    src.close()
    got = [gpu.to_numpy_u32(x) for x in copy.search_locate(d, m, R)]
    for g, w in zip(got, want):
        np.testing.assert_array_equal(g, w)
    sp_o, ep_o = oracle.search_batch(fo, reads, m, m, R)
    np.testing.assert_array_equal(got[0], sp_o)
    np.testing.assert_array_equal(got[1], ep_o)
    if form != "lean":
        assert copy.device_bytes > 5 * n  # the copy has position records and a k-mer table of its own
    copy.close()
