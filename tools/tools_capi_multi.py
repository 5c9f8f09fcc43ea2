#!/usr/bin/env python3
"""The batch forms of the reference-API layer over several device slots, on the GPU box:
`tools_capi_multi.py` runs itself once per device list (the list is read once per process) and
prints, per list, the time of make_fmi, of the first shared batch (replicas are made in it) and
of the batches after it.  On a one-GPU box the lists name that GPU several times: what is
measured is the cost of the split itself, not a speed-up."""
import ctypes as C, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np
    from bwt_amd import _lib, synth
    n, R, m = (int(x) for x in sys.argv[2:5])
    hl = C.CDLL(_lib.LIB_PATH)
    hl.make_fmi.restype = C.c_void_p; hl.make_fmi.argtypes = [C.c_void_p, C.c_int]
    hl.destroy_fmi.argtypes = [C.c_void_p]
    hl.fmi_search_locate_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_long, C.c_long] + [C.c_void_p] * 3
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    text = synth.gen_genome(n)
    t0 = time.perf_counter(); fmi = hl.make_fmi(vp(text), n); t1 = time.perf_counter()
    bases = synth.unpack(text, n).astype(np.uint8)
    rng = np.random.default_rng(3)
    offs = rng.integers(0, n - m, R)
    reads = np.ascontiguousarray(bases[offs[:, None] + np.arange(m)[None, :]])
    sp, ep, pos = (np.zeros(R, np.int32) for _ in range(3))
    times = []
    for rep in range(4):
        a = time.perf_counter()
        rc = hl.fmi_search_locate_batch(fmi, vp(reads), m, m, R, vp(sp), vp(ep), vp(pos))
        times.append((time.perf_counter() - a) * 1e3)
        assert rc == 0
    uniq = (ep - sp) == 1
    ok = bool((pos[uniq] == offs[uniq]).all()) and bool((ep > sp).all())
    print(f"devices={os.environ.get('BWT_MI355X_DEVICES')} slots={hl.fmi_device_slots()} make_fmi {1e3*(t1-t0):.1f} ms; "
          f"search+locate of {R} x {m} bp from host bytes: first call {times[0]:.1f} ms (replicas made), then "
          f"{min(times[1:]):.1f} ms = {R / min(times[1:]) / 1e3:.1f} M reads/s; located where cut: {ok}", flush=True)
    hl.destroy_fmi(fmi)
    sys.exit(0)

n, R, m = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (46_000_000, 10_000_000, 50)
for devices in ("0", "0,0", "0,0,0,0", "0,0,0,0,0,0,0,0"):
    env = dict(os.environ, BWT_MI355X_DEVICES=devices)
    env.pop("BWT_MI355X_DEVICE", None)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", str(n), str(R), str(m)], env=env)
    if r.returncode:
        sys.exit(r.returncode)
