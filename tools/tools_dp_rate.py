#!/usr/bin/env python3
"""Rate of the nw_fast / sw_fast batch kernel on the GPU box, and of the reference's own functions on one core."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from bwt_amd import gpu as G
from oracle import load_oracle, load_reference
G.init(0)
orc, ref = load_oracle(), load_reference()
rng = np.random.default_rng(3)
for l1, l2, R in ((30, 40, 2_000_000), (100, 110, 1_000_000), (250, 260, 200_000)):
    g = rng.integers(0, 4, (R, l2)).astype(np.uint8)
    s1 = g[:, :l1].copy()
    for _ in range(3):  # three substitutions per read
        p = rng.integers(0, l1, R)
        s1[np.arange(R), p] = (s1[np.arange(R), p] + 1) & 3
    off1 = torch.arange(R + 1, dtype=torch.int64, device="cuda") * l1
    off2 = torch.arange(R + 1, dtype=torch.int64, device="cuda") * l2
    d1 = torch.from_numpy(np.concatenate([s1.reshape(-1), np.zeros(1, np.uint8)])).cuda()
    d2 = torch.from_numpy(np.concatenate([g.reshape(-1), np.zeros(1, np.uint8)])).cuda()
    for mode, name in ((0, "nw_fast"), (1, "sw_fast")):
        for rep in range(2):
            G.timing_enable(True); G.timing_reset()
            ret, ops, nops = G.dp_align_batch(d1, off1, d2, off2, R, mode, l1, l2)
            torch.cuda.synchronize()
            ms = G.timing_report()["dp_align"][1]
        cells = R * l1 * l2
        k = 2000
        impl = ref or orc
        t0 = time.perf_counter()
        want = [impl.dp_align(s1[r], g[r], mode) for r in range(k)]
        dt = time.perf_counter() - t0
        ret_h, ops_h, nops_h = ret.cpu().numpy(), ops.cpu().numpy(), G.to_numpy_u32(nops)
        same = all((int(ret_h[r]), ops_h[r * (l1 + l2): r * (l1 + l2) + int(nops_h[r])].tobytes()) == want[r] for r in range(k))
        print(f"{name} {l1}x{l2}: {R} pairs in {ms:.2f} ms = {R/ms/1e3:.1f} M pairs/s = {cells/ms/1e9:.2f} T cells/s; "
              f"{'reference' if ref else 'oracle'} on one core {k/dt/1e3:.1f} k pairs/s = {k*l1*l2/dt/1e9:.3f} G cells/s; "
              f"first {k} equal: {same}", flush=True)
        assert same
print("OK")
