"""The reference's own self-checking program, compiled from its source and
linked with the MI355X library in place of the reference's objects
(oracle/Makefile: refprogs), run on the GPU: the drop-in check.  filetest.c
builds an index with make_fmi, writes it with write_index, reads it back with
read_index and makes 10^6 scalar locate() calls on 16-mers cut from the text at
random places, printing "Ruh roh j jj" when locate did not return the place a
read was cut at (filetest.c:74-105) -- which may only happen for a 16-mer that
occurs more than once, and locate says so first (seqindex.c:274-275)."""
import os
import subprocess
import time

import pytest

from bwt_amd import synth
from helpers import ROOT

pytestmark = pytest.mark.gpu

PROGRAM = os.path.join(ROOT, "oracle", "_ref", "filetest_on_mi355x")


@pytest.mark.skipif(not os.path.exists(PROGRAM), reason="oracle/_ref/filetest_on_mi355x was not built")
def test_reference_filetest_passes_on_the_library(oracle, gpu, tmp_path):
    n = 1_000_000
    text = oracle.gen_genome(n, 77)
    seq = tmp_path / "genome.txt"
    seq.write_text("".join("ACGT"[b] for b in synth.unpack(text, n)))
    t0 = time.perf_counter()
    run = subprocess.run([PROGRAM, str(seq)], capture_output=True, text=True, timeout=280)
    seconds = time.perf_counter() - t0
    assert run.returncode == 0, run.stderr[-1000:]
    assert "over a genome of length 1000000" in run.stderr
    lines = run.stdout.splitlines()
    warnings = sum(1 for ln in lines if ln.startswith("Warning: multiple matches"))
    wrong = [i for i, ln in enumerate(lines) if ln.startswith("Ruh roh")]
    # a 16-mer of a random 1 Mbp text repeats with probability about 2.3e-4: a few hundred of the 10^6 reads
    assert 50 < warnings < 1000
    assert len(wrong) <= warnings
    for i in wrong:  # each complaint follows the warning of the same call
        assert i > 0 and lines[i - 1].startswith("Warning: multiple matches")
    assert set(ln.split()[0] for ln in lines) <= {"Warning:", "Ruh"}
    print(f"filetest: 10^6 scalar locate calls in {seconds:.1f} s, {warnings} repeated 16-mers, {len(wrong)} of them elsewhere")


SINGLE_ALIGN = os.path.join(ROOT, "oracle", "_ref", "single_align_on_mi355x")


@pytest.mark.skipif(not os.path.exists(SINGLE_ALIGN), reason="oracle/_ref/single_align_on_mi355x was not built")
def test_reference_single_align_prints_the_same_on_the_library(oracle, gpu, tmp_path):
    """The reference's single_align.c with its own smw.c and stack.c, every mms,
    unc_sa and read_index a call into the MI355X library: the output of the run
    the fixtures were made from (there on the reference's own objects)."""
    import hashlib

    from helpers import GOLDEN, single_align_inputs

    rec = GOLDEN["single_align"]
    n, text, seq, rds, reads = single_align_inputs(oracle, str(tmp_path))
    idx = str(tmp_path / "genome.idx")
    drv = os.path.join(ROOT, "bwt_amd", "drivers")
    assert subprocess.run([os.path.join(drv, "build_index"), seq, idx], capture_output=True, timeout=120).returncode == 0
    run = subprocess.run([SINGLE_ALIGN, seq, idx, rds], capture_output=True, timeout=280)
    assert run.returncode == 0, run.stderr[-1000:]
    assert hashlib.sha256(run.stdout).hexdigest() == rec["stdout_sha256"]
    assert run.stderr.decode().splitlines()[-1] + "\n" == rec["stderr"]


BUILD_INDEX = os.path.join(ROOT, "oracle", "_ref", "build_index_on_mi355x")


@pytest.mark.skipif(not os.path.exists(BUILD_INDEX), reason="oracle/_ref/build_index_on_mi355x was not built")
def test_reference_build_index_writes_the_same_file_on_the_library(oracle, gpu, tmp_path):
    """The reference's build_index.c (its packer, make_fmi or make_fmi_sacak, write_index) on the library, and
    this repository's drivers/build_index: the same index file, byte for byte, which is also the file the
    oracle writes for that text."""
    from helpers import single_align_inputs

    n, text, seq, _, _ = single_align_inputs(oracle, str(tmp_path))
    theirs, ours, want = (str(tmp_path / x) for x in ("ref.idx", "drv.idx", "orc.idx"))
    assert subprocess.run([BUILD_INDEX, seq, theirs], capture_output=True, timeout=120).returncode == 0
    drv = os.path.join(ROOT, "bwt_amd", "drivers", "build_index")
    assert subprocess.run([drv, seq, ours], capture_output=True, timeout=120).returncode == 0
    assert oracle.write_index(oracle.make_fmi(text, n), want) == 0
    data = open(theirs, "rb").read()
    assert data == open(ours, "rb").read()
    assert data == open(want, "rb").read()
