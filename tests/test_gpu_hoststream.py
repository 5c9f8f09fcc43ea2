"""Host batches: the streamed path of bwtgpu_search_host (reads packed by host
threads chunk by chunk) and the packed-read entry points, against the device
search and the oracle."""
import ctypes as C

import numpy as np
import pytest
import torch

from bwt_amd import _lib, synth

vp = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731


def pack_reference(reads):
    """numpy statement of the packed format: base k at bits 2 (k % 16) of word k / 16"""
    R, m = reads.shape
    words = (m + 15) // 16
    padded = np.zeros((R, words * 16), np.uint64)
    padded[:, :m] = reads & 3
    shifts = (2 * (np.arange(words * 16) % 16)).astype(np.uint64)
    return (padded << shifts).reshape(R, words, 16).sum(axis=2).astype(np.uint32)


@pytest.mark.parametrize("m,stride", [(1, 1), (7, 7), (16, 16), (17, 40), (50, 50), (64, 64), (100, 133), (251, 251)])
def test_host_packer_matches_the_format(m, stride):
    """bwtgpu_pack_reads_host needs no GPU: host threads only."""
    lib = _lib.load()
    rng = np.random.default_rng(m)
    R = 70001
    buf = rng.integers(0, 256, (R - 1) * stride + m, dtype=np.uint8)  # only the low two bits of a byte count
    reads = np.stack([buf[r * stride: r * stride + m] for r in range(0, R, 997)])
    out = np.full((R, (m + 15) // 16), 0xDEADBEEF, np.uint32)
    assert lib.bwtgpu_pack_reads_host(vp(buf), m, stride, R, vp(out)) == 0
    np.testing.assert_array_equal(out[::997], pack_reference(reads))
    all_reads = np.lib.stride_tricks.as_strided(buf, (R, m), (stride, 1))
    np.testing.assert_array_equal(out, pack_reference(all_reads))


@pytest.mark.gpu
@pytest.mark.parametrize("m,R,built", [(50, 300_000, True), (33, 100_003, False), (100, 150_000, True),
                                       (16, 40_000, True), (600, 33_000, True)])
def test_streamed_host_search_equals_device_search(oracle, gpu, m, R, built):
    lib = _lib.load()
    n = 1_000_003
    text = oracle.gen_genome(n, synth.GENOME_SEED + m)
    fo = oracle.make_fmi(text, n)
    ix = gpu.build_index(text, n) if built else gpu.index_from_parts(fo.bwt, n, fo.endloc, fo.idxs, lean=True)
    reads, _ = oracle.gen_reads(text, n, R, m, 77 + m)
    reads = np.ascontiguousarray(reads)
    d = torch.from_numpy(np.concatenate([reads.reshape(-1), np.zeros(64, np.uint8)])).cuda()
    sp_d, ep_d, lr_d = ix.search(d, m, R, want_locrow=True)
    pos_d = gpu.to_numpy_u32(ix.locate_rows(lr_d))
    sp_d, ep_d, lr_d = (gpu.to_numpy_u32(x) for x in (sp_d, ep_d, lr_d))
    # the oracle on a sample
    pick = np.arange(0, R, 97)
    sp_o, ep_o = oracle.search_batch(fo, reads[pick], m, m, len(pick))
    np.testing.assert_array_equal(sp_d[pick], sp_o)
    np.testing.assert_array_equal(ep_d[pick], ep_o)
    # bytes in host memory: streamed (R >= 32768, reads do not overlap)
    sp, ep, lr, pos = (np.full(R, 0x55555555, np.uint32) for _ in range(4))
    _lib.check(lib.bwtgpu_search_host(ix.h, vp(reads), m, m, R, vp(sp), vp(ep), vp(lr), vp(pos)))
    for got, want in ((sp, sp_d), (ep, ep_d), (lr, lr_d), (pos, pos_d)):
        np.testing.assert_array_equal(got, want)
    # only some of the outputs asked for
    sp2 = np.zeros(R, np.uint32)
    _lib.check(lib.bwtgpu_search_host(ix.h, vp(reads), m, m, R, vp(sp2), None, None, None))
    np.testing.assert_array_equal(sp2, sp_d)
    pos2 = np.zeros(R, np.uint32)
    _lib.check(lib.bwtgpu_search_host(ix.h, vp(reads), m, m, R, None, None, None, vp(pos2)))
    np.testing.assert_array_equal(pos2, pos_d)
    # packed on the host first, then the packed entry points (host and device forms)
    words = np.zeros((R, (m + 15) // 16), np.uint32)
    _lib.check(lib.bwtgpu_pack_reads_host(vp(reads), m, m, R, vp(words)))
    sp3, ep3, pos3 = (np.zeros(R, np.uint32) for _ in range(3))
    _lib.check(lib.bwtgpu_search_packed_host(ix.h, vp(words), m, R, vp(sp3), vp(ep3), None, vp(pos3)))
    np.testing.assert_array_equal(sp3, sp_d)
    np.testing.assert_array_equal(ep3, ep_d)
    np.testing.assert_array_equal(pos3, pos_d)
    d_w = torch.from_numpy(words.view(np.int32)).cuda()
    a, b, c = (torch.empty(R, dtype=torch.int32, device="cuda") for _ in range(3))
    _lib.check(lib.bwtgpu_search_packed(ix.h, d_w.data_ptr(), m, R, a.data_ptr(), b.data_ptr(), c.data_ptr(), None))
    np.testing.assert_array_equal(gpu.to_numpy_u32(a), sp_d)
    np.testing.assert_array_equal(gpu.to_numpy_u32(b), ep_d)
    np.testing.assert_array_equal(gpu.to_numpy_u32(c), lr_d)
    ix.close()


@pytest.mark.gpu
def test_reference_api_packed_batch(oracle, gpu):
    """fmi_pack_reads + fmi_search_batch_packed through the reference-API layer."""
    hl = C.CDLL(_lib.LIB_PATH)
    hl.make_fmi.restype = C.c_void_p
    hl.make_fmi.argtypes = [C.c_void_p, C.c_int]
    hl.destroy_fmi.argtypes = [C.c_void_p]
    hl.fmi_pack_reads.restype = None
    hl.fmi_pack_reads.argtypes = [C.c_void_p, C.c_int, C.c_long, C.c_long, C.c_void_p]
    hl.fmi_search_batch_packed.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_long, C.c_void_p, C.c_void_p, C.c_void_p]
    n, R, m = 300_007, 50_000, 40
    text = oracle.gen_genome(n, 5)
    fo = oracle.make_fmi(text, n)
    reads, _ = oracle.gen_reads(text, n, R, m, 6)
    reads = np.ascontiguousarray(reads)
    fmi = hl.make_fmi(vp(text), n)
    words = np.zeros((R, 3), np.uint32)
    hl.fmi_pack_reads(vp(reads), m, m, R, vp(words))
    sp, ep, pos = (np.zeros(R, np.int32) for _ in range(3))
    assert hl.fmi_search_batch_packed(fmi, vp(words), m, R, vp(sp), vp(ep), vp(pos)) == 0
    sp_o, ep_o = oracle.search_batch(fo, reads, m, m, R)
    np.testing.assert_array_equal(sp.view(np.uint32), sp_o)
    np.testing.assert_array_equal(ep.view(np.uint32), ep_o)
    for r in range(0, R, 501):
        assert pos[r] == oracle.locate(fo, reads[r])
    hl.destroy_fmi(fmi)
