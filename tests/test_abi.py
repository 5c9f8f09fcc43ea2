"""The C-ABI library loads and exports every symbol include/*.h declares.
No compute calls: this runs where there is no GPU."""
import ctypes
import os
import re

import pytest

from bwt_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    names = re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{]*\)\s*;", src)
    return sorted(set(names) - {"defined"})


@pytest.mark.parametrize("header", ["bwt_gpu.h", "bwt_mi355x.h"])
def test_every_declared_symbol_is_exported(header):
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = declared(header)
    assert len(names) > 15
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_python_binding_covers_the_device_header():
    assert sorted(_lib.SIGNATURES) == declared("bwt_gpu.h")
    _lib.load()


def test_reference_struct_layout():
    """fm_index as seqindex.h:14-22 lays it out on x86-64: 64 bytes."""
    class FM(ctypes.Structure):
        _fields_ = [("bwt", ctypes.c_void_p), ("idxs", ctypes.c_void_p), ("rank_index", ctypes.c_void_p),
                    ("lookup", ctypes.c_void_p), ("endloc", ctypes.c_int), ("C", ctypes.c_int * 5),
                    ("len", ctypes.c_int)]
    assert ctypes.sizeof(FM) == 64
    assert (FM.endloc.offset, FM.C.offset, FM.len.offset) == (32, 36, 56)


def test_no_cpu_fallback_without_a_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib = _lib.load()
    assert lib.bwtgpu_init(0) != 0
    assert b"device" in lib.bwtgpu_last_error().lower()
    # the reference-API layer reports and returns its failure value
    hl = ctypes.CDLL(_lib.LIB_PATH)
    hl.make_fmi.restype = ctypes.c_void_p
    assert hl.make_fmi(ctypes.create_string_buffer(8), 16) is None


def test_host_packer_matches_build_index(oracle):
    hl = ctypes.CDLL(_lib.LIB_PATH)
    for text in [b"", b"A", b"ACGTTTT", b"ACGTACGT", b"acgtNNCG\nTTGCA", bytes(range(256)) * 3]:
        out = ctypes.create_string_buffer(len(text) // 4 + 1)
        hl.fmi_pack_ascii(text, ctypes.c_long(len(text)), out)
        assert out.raw == oracle.pack_ascii(text).tobytes()
