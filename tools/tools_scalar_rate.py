#!/usr/bin/env python3
"""Scalar calls through the reference's C API on the GPU box: one thread, then four at once (fmitest.c:67-91)."""
import ctypes as C, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bwt_amd import _lib
from oracle import load_oracle
orc = load_oracle()
L = C.CDLL(_lib.LIB_PATH)
L.make_fmi.restype = C.c_void_p; L.make_fmi.argtypes = [C.c_void_p, C.c_int]
L.reverse_search.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
L.locate.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
n = 1_000_000
text = orc.gen_genome(n, 77)
fmi = L.make_fmi(text.ctypes.data_as(C.c_void_p), n)
reads, _ = orc.gen_reads(text, n, 40000, 12, 5)
ptrs = [r.ctypes.data_as(C.c_void_p) for r in reads]
def work(t, nt, fn, acc):
    s = 0
    for i in range(t, len(ptrs), nt):
        s += fn(fmi, ptrs[i], 12)
    acc[t] = s
for name, fn in (("reverse_search", L.reverse_search), ("locate", L.locate)):
    for nt in (1, 4):
        acc = [0] * nt
        th = [threading.Thread(target=work, args=(t, nt, fn, acc)) for t in range(nt)]
        t0 = time.perf_counter()
        for x in th: x.start()
        for x in th: x.join()
        dt = time.perf_counter() - t0
        print(f"{name}: {len(ptrs)} calls from {nt} thread(s) in {dt:.2f} s = {dt/len(ptrs)*1e6:.1f} us a call overall, checksum {sum(acc)}", flush=True)
