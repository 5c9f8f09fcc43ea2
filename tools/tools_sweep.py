#!/usr/bin/env python3
"""Search/locate kernel time vs index size (diagnostic, GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bwt_amd import gpu as G, synth
G.init(0)
R, m = 10_000_000, int(sys.argv[1]) if len(sys.argv) > 1 else 50
for n in [1_000_000, 4_000_000, 16_000_000, 46_000_000, 128_000_000, 512_000_000]:
    t = G.gen_genome(n, synth.GENOME_SEED)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ix = G.build_index_dev(t, n)
    torch.cuda.synchronize(); tb = time.perf_counter() - t0
    reads, _ = G.gen_reads(t, n, R, m, synth.READS_SEED, want_offsets=False)
    ix.search(reads, m, R, want_locrow=True)
    G.timing_enable(True); G.timing_reset()
    for _ in range(3):
        sp, ep, lr = ix.search(reads, m, R, want_locrow=True)
    pos = ix.locate_rows(lr)
    torch.cuda.synchronize()
    rep = G.timing_report()
    G.timing_enable(False)
    print(f"n={n:>11d} build_ms={tb*1e3:8.2f} index_MB={ix.device_bytes/1e6:8.1f} search_ms={rep['fm_search'][1]/rep['fm_search'][0]:7.3f} "
          f"locate_ms={rep['fm_locate'][1]:7.3f} Greads/s={R/(rep['fm_search'][1]/rep['fm_search'][0])/1e6:6.3f}", flush=True)
    ix.close(); del reads, t
