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
    hl.histsort.restype = ctypes.c_void_p
    assert hl.histsort(ctypes.create_string_buffer(8), 16) is None
    # ... and so do the batch forms, even on an fm_index whose host arrays would answer single queries
    from test_host_scalars import host_fm_index
    from oracle import load_oracle
    orc = load_oracle()
    n = 500
    fo = orc.make_fmi(orc.gen_genome(n, 1), n)
    f, keep = host_fm_index(fo, n)
    sp, ep = (ctypes.c_int * 4)(), (ctypes.c_int * 4)()
    assert hl.fmi_search_batch(ctypes.byref(f), ctypes.create_string_buffer(64), 8, 8, 4, sp, ep) == -1


def test_host_packer_matches_build_index(oracle):
    hl = ctypes.CDLL(_lib.LIB_PATH)
    for text in [b"", b"A", b"ACGTTTT", b"ACGTACGT", b"acgtNNCG\nTTGCA", bytes(range(256)) * 3]:
        out = ctypes.create_string_buffer(len(text) // 4 + 1)
        hl.fmi_pack_ascii(text, ctypes.c_long(len(text)), out)
        assert out.raw == oracle.pack_ascii(text).tobytes()


REFERENCE = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="the reference tree is not on this machine")
@pytest.mark.parametrize("program,extra", [("build_index", []), ("fmitest", []), ("searchtest", []), ("filetest", []),
                                           ("index_test", []), ("search_reads", []), ("gaptest", []),
                                           ("histsortcomptest", []), ("single_align", ["smw.c", "stack.c"]),
                                           ("rnaseqtest", ["smw.c", "stack.c"])])
def test_reference_programs_link_against_the_library(program, extra, tmp_path):
    """The reference's own programs, compiled from where they lie with the
    reference's own headers, link with libbwt_mi355x.so in place of
    histsortcomp.o csacak.o seqindex.o fileio.o (reference Makefile:32-60; the
    aligners keep their own smw.c and stack.c): every symbol they use is
    exported.  Nothing is run (no device here) and nothing is
    kept."""
    import subprocess

    src = os.path.join(REFERENCE, program + ".c")
    if not os.path.exists(src):
        pytest.skip(f"{program}.c is not in the reference tree")
    out = str(tmp_path / program)
    lib_dir = os.path.join(ROOT, "bwt_amd")
    res = subprocess.run(["gcc", "-std=gnu99", "-O1", "-w", "-pthread", "-I" + REFERENCE, "-o", out, src]
                         + [os.path.join(REFERENCE, e) for e in extra] + ["-L" + lib_dir, "-lbwt_mi355x", "-Wl,-rpath," + lib_dir, "-lm"],
                         capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr[-2000:]
