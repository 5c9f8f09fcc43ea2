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
