#!/usr/bin/env python3
"""PCIe-inclusive rates of the host-buffer search (what fmi_search_batch costs a C caller):
bytes in pageable memory (packed by host threads on the way), packed words in pageable
memory, packed words in pinned memory."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bwt_amd import gpu as G, synth, _lib
G.init(0)
n, R, m = 46_000_000, 10_000_000, 50
if len(sys.argv) > 1:
    n, R, m = (int(x) for x in sys.argv[1:4])
t = G.gen_genome(n, synth.GENOME_SEED)
ix = G.build_index_dev(t, n)
ix.prepare_search()
reads, _ = G.gen_reads(t, n, R, m, synth.READS_SEED, want_offsets=False)
h = reads[: R * m].cpu().numpy()
sp = np.empty(R, np.uint32); ep = np.empty(R, np.uint32); pos = np.empty(R, np.uint32)
lib = _lib.load()
vp = lambda a: a.ctypes.data_as(C.c_void_p)
d_sp, d_ep, d_lr = ix.search(reads, m, R, want_locrow=True)
want = [G.to_numpy_u32(x) for x in (d_sp, d_ep)] + [G.to_numpy_u32(ix.locate_rows(d_lr))]
print(f"host threads: {os.environ.get('BWT_MI355X_HOST_THREADS', 'default (8)')}, cpus visible {os.cpu_count()}")


def bench(label, fn, nbytes):
    best = 1e9
    for rep in range(4):
        t0 = time.perf_counter()
        fn()
        best = min(best, time.perf_counter() - t0)
    print(f"{label}: {best*1e3:.1f} ms = {R/best/1e6:.1f} M reads/s ({nbytes/best/1e9:.1f} GB/s over PCIe)", flush=True)


for with_pos in (False, True):
    p = vp(pos) if with_pos else None
    bench(f"bytes in pageable memory, locate={with_pos}",
          lambda: _lib.check(lib.bwtgpu_search_host(ix.h, vp(h), m, m, R, vp(sp), vp(ep), None, p)),
          R * ((m + 15) // 16 * 4) + R * (12 if with_pos else 8))
    assert (sp == want[0]).all() and (ep == want[1]).all() and (not with_pos or (pos == want[2]).all())
W = (m + 15) // 16
words = np.empty((R, W), np.uint32)
t0 = time.perf_counter()
_lib.check(lib.bwtgpu_pack_reads_host(vp(h), m, m, R, vp(words)))
print(f"bwtgpu_pack_reads_host: {(time.perf_counter()-t0)*1e3:.1f} ms for {R*m/1e6:.0f} MB of bases")
for with_pos in (False, True):
    p = vp(pos) if with_pos else None
    bench(f"packed words in pageable memory, locate={with_pos}",
          lambda: _lib.check(lib.bwtgpu_search_packed_host(ix.h, vp(words), m, R, vp(sp), vp(ep), None, p)),
          R * W * 4 + R * (12 if with_pos else 8))
    assert (sp == want[0]).all() and (ep == want[1]).all() and (not with_pos or (pos == want[2]).all())
# pinned input and output arrays
pw = torch.from_numpy(words).pin_memory()
psp, pep, ppos = (torch.empty(R, dtype=torch.int32).pin_memory() for _ in range(3))
for with_pos in (False, True):
    p = C.c_void_p(ppos.data_ptr()) if with_pos else None
    bench(f"packed words in pinned memory, pinned results, locate={with_pos}",
          lambda: _lib.check(lib.bwtgpu_search_packed_host(ix.h, C.c_void_p(pw.data_ptr()), m, R, C.c_void_p(psp.data_ptr()),
                                                           C.c_void_p(pep.data_ptr()), None, p)),
          R * W * 4 + R * (12 if with_pos else 8))
    assert (psp.numpy().view(np.uint32) == want[0]).all() and (pep.numpy().view(np.uint32) == want[1]).all()
