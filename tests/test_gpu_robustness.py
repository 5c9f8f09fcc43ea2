"""Behaviour at the edges of the index object: parts that do not belong together,
an index loaded without its sampled rows, and the build caches of threads that
come and go."""
import ctypes as C
import threading

import numpy as np
import pytest
import torch

from bwt_amd import synth

pytestmark = pytest.mark.gpu


def test_inconsistent_parts_end_in_an_error_or_an_index_never_in_a_hang(oracle, gpu):
    """read_index hands the device whatever the file holds.  With a wrong $ row the
    LF map need not be one cycle; walks are cut off, and a load whose walks from the
    stored rows do not pass every row once, each arriving at the next stored row with
    the position stored for it, is refused."""
    for n in (40, 97, 200, 5000):
        text = oracle.gen_genome(n, 11 + n)
        fo = oracle.make_fmi(text, n)
        refused = 0
        tries = range(1, n + 1) if n <= 200 else range(1, n + 1, 97)
        for endloc in tries:
            if endloc == fo.endloc:
                continue
            try:
                ix = gpu.index_from_parts(fo.bwt, n, endloc, fo.idxs)
            except gpu.BwtGpuError as e:
                assert "inconsistent" in str(e)
                refused += 1
                continue
            rows = torch.arange(0, n + 1, dtype=torch.int32, device="cuda")
            pos = gpu.to_numpy_u32(ix.locate_rows(rows))  # ends, whatever it answers
            assert ((pos <= n) | (pos == 0xFFFFFFFF)).all()
            ix.close()
        # the right $ row loads and locates every row
        ix = gpu.index_from_parts(fo.bwt, n, fo.endloc, fo.idxs)
        rows = torch.arange(0, n + 1, dtype=torch.int32, device="cuda")
        np.testing.assert_array_equal(gpu.to_numpy_u32(ix.locate_rows(rows)), oracle.histsort(text, n))
        ix.close()
    # samples of another text: the BWT is sound and every walk ends, but the positions the walks
    # arrive with are not the ones stored for the rows they arrive at
    n = 5000
    a, b = oracle.make_fmi(oracle.gen_genome(n, 1), n), oracle.make_fmi(oracle.gen_genome(n, 2), n)
    with pytest.raises(gpu.BwtGpuError, match="inconsistent"):
        gpu.index_from_parts(a.bwt, n, a.endloc, b.idxs)
    # one stored row off by one: refused as well
    idxs = a.idxs.copy()
    idxs[len(idxs) // 2] ^= 1
    with pytest.raises(gpu.BwtGpuError, match="inconsistent"):
        gpu.index_from_parts(a.bwt, n, a.endloc, idxs)


def test_index_without_sampled_rows_searches_but_refuses_to_locate(oracle, gpu):
    n, R, m = 30000, 500, 25
    text = oracle.gen_genome(n, 3)
    fo = oracle.make_fmi(text, n)
    lib = gpu._lib.load()
    h = lib.bwtgpu_index_from_parts(fo.bwt.ctypes.data_as(C.c_void_p), n, fo.endloc, None, None)
    ix = gpu.GpuIndex(h)
    reads, _ = oracle.gen_reads(text, n, R, m, 5)
    sp_o, ep_o = oracle.search_batch(fo, reads, m, m, R)
    d = torch.from_numpy(reads.reshape(-1)).cuda()
    sp, ep, lr = ix.search(d, m, R, want_locrow=True)
    np.testing.assert_array_equal(gpu.to_numpy_u32(sp), sp_o)
    np.testing.assert_array_equal(gpu.to_numpy_u32(ep), ep_o)
    with pytest.raises(gpu.BwtGpuError, match="without sampled"):
        ix.locate_rows(lr)
    with pytest.raises(gpu.BwtGpuError, match="without sampled"):
        ix.search_locate(d, m, R)
    with pytest.raises(gpu.BwtGpuError, match="without sampled"):
        ix.search_locate_packed(gpu.pack_reads_dev(d, m, R), m, R)
    offs = torch.arange(R + 1, dtype=torch.int64, device="cuda") * m
    out = torch.empty(4 * R, dtype=torch.int32, device="cuda")
    rc = lib.bwtgpu_search_reads(ix.h, d.data_ptr(), None, offs.data_ptr(), R, 20, out.data_ptr(), None)
    assert rc == -2 and b"without sampled" in lib.bwtgpu_last_error()
    ix.close()


def test_release_cache_reaches_the_caches_of_other_threads(oracle, gpu):
    """A worker thread builds an index and ends; its 21 bytes per base of build
    workspace must not stay out of reach: it passes to the idle pool, and
    bwtgpu_release_cache() from any thread returns it to the driver."""
    lib = gpu._lib.load()
    n = 8_000_000
    d_text = gpu.gen_genome(n, synth.GENOME_SEED)
    torch.cuda.synchronize()
    lib.bwtgpu_release_cache()
    free0 = torch.cuda.mem_get_info()[0]
    result = {}

    def worker():
        gpu.init(0)
        ix = gpu.build_index_dev(d_text, n)
        result["C"] = ix.C
        ix.close()

    t = threading.Thread(target=worker)
    t.start()
    t.join()
    assert result["C"][4] == n + 1
    held = free0 - torch.cuda.mem_get_info()[0]
    assert held > 10 * n  # the workspace and the index buffers are still held (pool and the dead thread's cache)
    lib.bwtgpu_release_cache()
    assert free0 - torch.cuda.mem_get_info()[0] < 2 * n  # ... and now they are not
    # a second thread's live cache is released too, while the thread is idle
    ready, go = threading.Event(), threading.Event()

    def worker2():
        gpu.init(0)
        gpu.build_index_dev(d_text, n).close()
        ready.set()
        go.wait(30)

    t = threading.Thread(target=worker2)
    t.start()
    ready.wait(60)
    lib.bwtgpu_release_cache()
    assert free0 - torch.cuda.mem_get_info()[0] < 2 * n
    go.set()
    t.join()
