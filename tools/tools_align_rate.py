#!/usr/bin/env python3
"""Rate of the align_read_anchored kernel on the GPU box, and of the reference's function on one core."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bwt_amd import gpu as G, synth
from oracle import load_oracle, load_reference
G.init(0)
orc, ref = load_oracle(), load_reference()
n, R, m = 46_000_000, 1_000_000, 100
t = G.gen_genome(n, synth.GENOME_SEED)
text = t[: n // 4 + 1].cpu().numpy()
ix = G.build_index_dev(t, n)
reads, _ = G.gen_reads(t, n, R, m, 77, want_offsets=False)
r = reads[: R * m].view(R, m).cpu().numpy().copy()
rng = np.random.default_rng(5)
rows = np.nonzero(rng.random(R) < 0.5)[0]          # half the reads: two substitutions
for _ in range(2):
    p = rng.integers(12, m, len(rows))
    r[rows, p] = (r[rows, p] + 1) & 3
rows = np.nonzero(rng.random(R) < 0.1)[0]          # a tenth: two bases deleted (the tail shifts up, two new bases at the end)
for q in rows:
    p = int(rng.integers(20, m - 20))
    r[q, p:m - 2] = r[q, p + 2:m]
    r[q, m - 2:] = rng.integers(0, 4, 2)
rows = np.nonzero(rng.random(R) < 0.05)[0]         # a twentieth: an N
r[rows, rng.integers(12, m, len(rows))] = 5
flat = torch.from_numpy(np.concatenate([r.reshape(-1), np.zeros(1, np.uint8)])).cuda()
offs = torch.arange(R + 1, dtype=torch.int64, device="cuda") * m
for name in ("full", "compact"):
    if name == "compact":
        ix.compact()
    for rep in range(2):
        G.timing_enable(True); G.timing_reset()
        pos, ops, nops = G.align_reads(ix, t, flat, offs, R, 12, m)
        torch.cuda.synchronize()
        ms = G.timing_report()["align_reads"][1]
    pos_h = pos.cpu().numpy()
    print(f"align_reads {name}: {R} x {m} bp in {ms:.2f} ms = {R/ms/1e3:.2f} M reads/s; aligned {int((pos_h != 0).sum())}, "
          f"undefined in the reference {int((pos_h == -2**31).sum())}", flush=True)
k = 3000
impl = ref or orc
f = impl.make_fmi(text, n)
t0 = time.perf_counter()
want = [impl.align_read_anchored(f, text, r[i], 12) for i in range(k)]
dt = time.perf_counter() - t0
ops_h, nops_h = ops.cpu().numpy(), G.to_numpy_u32(nops)
same = bad = 0
for i in range(k):
    if pos_h[i] == -2**31:
        continue
    o = 5 * i * m + 64 * i
    got = (int(pos_h[i]), ops_h[o:o + int(nops_h[i])].tobytes())
    same += got == want[i]; bad += got != want[i]
print(f"{'reference' if ref else 'oracle'} on one core: {k/dt/1e3:.2f} k reads/s; first {k} reads: {same} equal, {bad} different", flush=True)
assert bad == 0
print("OK")
