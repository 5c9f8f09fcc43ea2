"""Shared helpers for the parity tests."""
import hashlib
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = json.load(open(os.path.join(ROOT, "tests", "golden", "golden.json")))
GENOME_SEED = GOLDEN["genome_seed"]
READS_SEED = GOLDEN["reads_seed"]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def file_sha(path):
    data = open(path, "rb").read()
    return hashlib.sha256(data).hexdigest(), len(data)


def mms_patterns(orc, text, n, count, m, seed):
    """same construction as tests/golden/make_golden.py"""
    reads, _ = orc.gen_reads(text, n, count, m, seed)
    rng = np.random.default_rng(seed)
    for r in range(count):
        k = rng.integers(0, 4)
        for _ in range(k):
            reads[r, rng.integers(0, m)] = 5
        if r % 7 == 0:
            reads[r, m - 1 - rng.integers(0, 3):] = 5
        if (reads[r] == 5).all():
            reads[r, 0] = 1
    return reads


def probe_rows(rec):
    return np.arange(rec["probe_rows"], dtype=np.uint32) * rec["probe_rows_step"]
