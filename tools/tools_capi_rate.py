#!/usr/bin/env python3
"""Cost of the reference-API calls (host buffers in, host structs out) on the GPU box."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bwt_amd import _lib, synth
hl = C.CDLL(_lib.LIB_PATH)
hl.make_fmi.restype = C.c_void_p; hl.make_fmi.argtypes = [C.c_void_p, C.c_int]
hl.destroy_fmi.argtypes = [C.c_void_p]
hl.fmi_reverse_search_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_long, C.c_long, C.c_void_p]
hl.fmi_search_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_long, C.c_long, C.c_void_p, C.c_void_p]
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 46_000_000
text = synth.gen_genome(n)
for rep in range(3):
    t0 = time.perf_counter(); fmi = hl.make_fmi(text.ctypes.data_as(C.c_void_p), n); t1 = time.perf_counter()
    print(f"make_fmi({n}) call {rep}: {(t1-t0)*1e3:.1f} ms", flush=True)
    if rep < 2: hl.destroy_fmi(fmi)
R = 10_000_000
pats = (np.random.default_rng(1).integers(0, 4, R + 11)).astype(np.uint8)
res = np.zeros(R, np.int32)
for rep in range(3):
    t0 = time.perf_counter(); hl.fmi_reverse_search_batch(fmi, pats.ctypes.data_as(C.c_void_p), 12, 1, R, res.ctypes.data_as(C.c_void_p)); t1 = time.perf_counter()
    print(f"fmi_reverse_search_batch 10M x 12 (stride 1) call {rep}: {(t1-t0)*1e3:.1f} ms", flush=True)
hl.destroy_fmi(fmi)
