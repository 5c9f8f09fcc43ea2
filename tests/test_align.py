"""align_read_anchored (single_align.c:163-298): the oracle's restatement
against the compiled reference and the golden vectors, and the GPU kernel (one
wave per read) against the oracle."""
import numpy as np
import pytest

from bwt_amd import synth
from helpers import GOLDEN, align_cases, align_digest


def _setup(oracle):
    rec = GOLDEN["align"]
    n = rec["n"]
    text = oracle.gen_genome(n, rec["genome_seed"])
    reads = align_cases(synth.unpack(text, n), n, rec["count"], rec["reads_seed"])
    return rec, n, text, reads


def test_oracle_align_equals_the_golden_vectors(oracle):
    """positions and CIGAR stacks the reference returned when the fixtures were made"""
    rec, n, text, reads = _setup(oracle)
    fo = oracle.make_fmi(text, n)
    got = [oracle.align_read_anchored(fo, text, rd, rec["anchor"]) for rd in reads]
    got = [g for g in got if g[0] != oracle.ALIGN_UNDEFINED]
    assert len(got) == rec["kept"] and sum(1 for p, _ in got if p) == rec["aligned"]
    assert [[p, o.decode()] for p, o in got[:len(rec["first"])]] == rec["first"]
    assert align_digest(got) == rec["sha256"]


def test_oracle_align_equals_the_reference(oracle, reference):
    n = 150000
    text = oracle.gen_genome(n, 4711)
    fo, fr = oracle.make_fmi(text, n), reference.make_fmi(text, n)
    undefined = 0
    for rd in align_cases(synth.unpack(text, n), n, 1500, 99, lo=25, hi=200):
        a = oracle.align_read_anchored(fo, text, rd, 12)
        if a[0] == oracle.ALIGN_UNDEFINED:  # the reference calls mms on no bases there
            undefined += 1
            continue
        assert a == reference.align_read_anchored(fr, text, rd, 12)
    assert undefined < 150


def _gpu_align(gpu, ix, text, reads, anchor, max_len):
    import torch

    offs = np.zeros(len(reads) + 1, np.int64)
    offs[1:] = np.cumsum([len(r) for r in reads])
    flat = np.concatenate(reads + [np.zeros(1, np.uint8)])
    d_text = torch.from_numpy(np.concatenate([text, np.zeros(64, np.uint8)])).cuda()
    pos, ops, nops = gpu.align_reads(ix, d_text, torch.from_numpy(flat).cuda(), torch.from_numpy(offs).cuda(),
                                     len(reads), anchor, max_len)
    pos, ops, nops = pos.cpu().numpy(), ops.cpu().numpy(), gpu.to_numpy_u32(nops)
    out = []
    for r in range(len(reads)):
        o = 5 * int(offs[r]) + 64 * r
        out.append((int(pos[r]), None if nops[r] >= 0xFFFFFFFE else ops[o:o + int(nops[r])].tobytes()))
    return out


@pytest.mark.gpu
def test_gpu_align_equals_the_oracle_and_the_golden_vectors(oracle, gpu):
    rec, n, text, reads = _setup(oracle)
    fo = oracle.make_fmi(text, n)
    want = [oracle.align_read_anchored(fo, text, rd, rec["anchor"]) for rd in reads]
    full = gpu.build_index(text, n)
    lean = gpu.index_from_parts(fo.bwt, n, fo.endloc, fo.idxs, lean=True)
    for name, ix in (("built from the text", full), ("loaded from parts", lean)):
        got = _gpu_align(gpu, ix, text, reads, rec["anchor"], 160)
        for rd, g, w in zip(reads, got, want):
            assert g == w, (name, len(rd))
        kept = [g for g in got if g[0] != oracle.ALIGN_UNDEFINED]
        assert align_digest(kept) == rec["sha256"]


@pytest.mark.gpu
@pytest.mark.parametrize("n,lo,hi,anchor", [(5000, 25, 60, 12), (300000, 40, 251, 12), (300000, 60, 130, 20),
                                            (60000, 20, 90, 8), (300000, 300, 480, 12)])
def test_gpu_align_more_texts_and_lengths(oracle, gpu, n, lo, hi, anchor):
    text = oracle.gen_genome(n, 31 + n)
    fo = oracle.make_fmi(text, n)
    reads = align_cases(synth.unpack(text, n), n, 1500 if hi < 300 else 300, n + anchor, lo=lo, hi=hi)
    reads += [np.array([0, 1, 2], np.uint8), np.zeros(anchor, np.uint8), np.full(40, 5, np.uint8)[:0]]  # short, empty
    want = [oracle.align_read_anchored(fo, text, rd, anchor) if len(rd) else (0, b"") for rd in reads]
    got = _gpu_align(gpu, gpu.build_index(text, n), text, reads, anchor, min(hi + 10, 479))
    for rd, g, w in zip(reads, got, want):
        assert g == w, len(rd)


@pytest.mark.gpu
@pytest.mark.parametrize("anchor", [12, 20])
def test_gpu_align_anchor_at_the_end_of_the_text_with_trailing_ns(oracle, gpu, anchor):
    """mms skips trailing Ns and counts them (seqindex.c:317-320): an anchor on the last bases of
    the text then reaches past its end, the genome piece of the (empty) tail comes out negative
    (single_align.c:196-198), and nw_fast returns at once for the empty read piece (smw.c:28-30)."""
    n = 460
    text = oracle.gen_genome(n, 77)
    bases = synth.unpack(text, n)
    fo = oracle.make_fmi(text, n)
    reads = []
    for tail_ns in (1, 2, 3, 5):
        for body in (24, 30, 37, 60):
            for cut in (0, 1, 4):  # the copy ends `cut` bases before the end of the text
                rd = np.concatenate([bases[n - cut - body: n - cut], np.full(tail_ns, 5, np.uint8)])
                reads.append(rd.astype(np.uint8))
                edited = rd.copy()
                edited[3] = (edited[3] + 1) & 3
                reads.append(edited.astype(np.uint8))
    want = [oracle.align_read_anchored(fo, text, rd, anchor) for rd in reads]
    for ix in (gpu.build_index(text, n), gpu.index_from_parts(fo.bwt, n, fo.endloc, fo.idxs, lean=True)):
        got = _gpu_align(gpu, ix, text, reads, anchor, 80)
        for rd, g, w in zip(reads, got, want):
            assert g == w, (len(rd), bytes(rd[-8:]).hex())


@pytest.mark.gpu
def test_gpu_align_marks_reads_it_was_not_sized_for(oracle, gpu):
    n = 20000
    text = oracle.gen_genome(n, 5)
    bases = synth.unpack(text, n)
    reads = [bases[100:220].copy(), bases[300:340].copy()]
    got = _gpu_align(gpu, gpu.build_index(text, n), text, reads, 12, 100)
    assert got[0] == (0, None)
    assert got[1][0] == 300 and got[1][1] == b"M" * 40


@pytest.mark.gpu
def test_single_align_program_prints_what_the_reference_program_printed(oracle, gpu, tmp_path):
    """drivers/single_align on the files the reference's own main() was run on
    when the fixtures were made (tests/golden/make_golden.py): same stdout, same
    count on stderr."""
    import hashlib
    import os
    import subprocess

    from helpers import ROOT, single_align_inputs

    rec = GOLDEN["single_align"]
    n, text, seq, rds, reads = single_align_inputs(oracle, str(tmp_path))
    assert len(reads) == rec["reads"]
    idx = str(tmp_path / "genome.idx")
    drv = os.path.join(ROOT, "bwt_amd", "drivers")
    assert subprocess.run([os.path.join(drv, "build_index"), seq, idx], capture_output=True, timeout=120).returncode == 0
    run = subprocess.run([os.path.join(drv, "single_align"), seq, idx, rds], capture_output=True, timeout=300)
    assert run.returncode == 0, run.stderr
    assert run.stdout[:300].decode() == rec["stdout_head"]
    assert run.stdout.count(b"\n") == rec["stdout_lines"]
    assert hashlib.sha256(run.stdout).hexdigest() == rec["stdout_sha256"]
    assert run.stderr.decode().splitlines()[-1] + "\n" == rec["stderr"]


@pytest.mark.gpu
@pytest.mark.parametrize("period,mutations", [(7, 0), (50, 0), (50, 60)])
def test_gpu_align_on_repeats(oracle, gpu, period, mutations):
    """Tandem repeats: anchors are rarely unique and the gap search walks
    intervals of hundreds of rows (single_align.c:217-218), 64 rows a step on
    the GPU, taking the first that fits as the reference does."""
    n = 4000
    text = synth.gen_tandem(n, period, 5, mutations, 9 if mutations else 0)
    fo = oracle.make_fmi(text, n)
    reads = align_cases(synth.unpack(text, n), n, 300, 1234 + period, lo=30, hi=120)
    want = [oracle.align_read_anchored(fo, text, rd, 12) for rd in reads]
    for ix in (gpu.build_index(text, n), gpu.index_from_parts(fo.bwt, n, fo.endloc, fo.idxs, lean=True)):
        got = _gpu_align(gpu, ix, text, reads, 12, 130)
        for rd, g, w in zip(reads, got, want):
            assert g == w, len(rd)
