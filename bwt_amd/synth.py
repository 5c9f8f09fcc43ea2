"""Seeded synthetic genomes and reads (counter-based splitmix64), in numpy.

The same streams are produced in HBM by bwtgpu_gen_genome / bwtgpu_gen_reads
(csrc/gen.hip); the tests check that the two agree bit for bit.

Packed text layout (the reference's, histsortcomp.c:20-23): base i lives in
byte i>>2 at bits 2*(3-(i&3)); A=0 C=1 G=2 T=3; the buffer has n//4+1 bytes
and every bit from base n on reads 0 (fmitest.c:130-142).
"""
import numpy as np

GENOME_SEED = 0x5EED0001
READS_SEED = 0x5EED0002

_GOLD = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def rnd(seed, k):
    """splitmix64 finaliser of seed + (k+1)*golden; k is a uint64 array."""
    k = np.asarray(k, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + (k + np.uint64(1)) * _GOLD
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def packed_bytes(n):
    return n // 4 + 1


def gen_genome(n, seed=GENOME_SEED):
    nb = packed_bytes(n)
    words = (nb + 7) // 8
    z = rnd(seed, np.arange(words, dtype=np.uint64))
    text = z.astype("<u8").view(np.uint8)[:nb].copy()
    text[n // 4] &= (0x00, 0xC0, 0xF0, 0xFC)[n & 3]
    return text


def unpack(text, n):
    """One base per byte."""
    t = np.asarray(text, np.uint8)[: (n + 3) // 4]
    b = np.empty((len(t), 4), np.uint8)
    for k in range(4):
        b[:, k] = (t >> (2 * (3 - k))) & 3
    return b.reshape(-1)[:n]


def pack(bases):
    """Inverse of unpack; returns n//4+1 bytes with a zero tail."""
    bases = np.asarray(bases, np.uint8)
    n = len(bases)
    padded = np.zeros((n // 4 + 1) * 4, np.uint8)
    padded[:n] = bases & 3
    q = padded.reshape(-1, 4)
    return (q[:, 0] << 6 | q[:, 1] << 4 | q[:, 2] << 2 | q[:, 3]).astype(np.uint8)


def gen_tandem(n, period, unit_seed=GENOME_SEED, mut_every=0, mut_seed=0):
    """unit[i % period], with a point mutation wherever
    rnd(mut_seed, i) % mut_every == 0 (mut_every == 0: none)."""
    unit = unpack(gen_genome(period, unit_seed), period)
    i = np.arange(n, dtype=np.uint64)
    bases = unit[(i % np.uint64(period)).astype(np.int64)].copy()
    if mut_every:
        r = rnd(mut_seed, i)
        hit = (r % np.uint64(mut_every)) == 0
        bump = (np.uint64(1) + (r >> np.uint64(32)) % np.uint64(3)).astype(np.uint8)
        bases[hit] = (bases[hit] + bump[hit]) & 3
    return pack(bases)


def gen_reads(text, n, R, m, seed=READS_SEED):
    """R reads of m bases (one byte per base); about 1 % are random, the rest
    substrings of the text.  Returns (reads[R, m], offsets[R]) with
    offsets == 0xFFFFFFFF for random reads."""
    bases = unpack(text, n)
    r = np.arange(R, dtype=np.uint64)
    off = (rnd(seed, 2 * r) % np.uint64(n - m + 1)).astype(np.int64)
    is_rand = (rnd(seed, 2 * r + 1) % np.uint64(100)) == 0
    reads = bases[off[:, None] + np.arange(m)[None, :]]
    words = (m + 31) // 32
    for rr in np.nonzero(is_rand)[0]:
        z = rnd(seed ^ 0xABCD, np.uint64(rr) * np.uint64(words) + np.arange(words, dtype=np.uint64))
        k = np.arange(m)
        reads[rr] = ((z[k // 32] >> (2 * (k % 32)).astype(np.uint64)) & np.uint64(3)).astype(np.uint8)
    offsets = off.astype(np.uint32)
    offsets[is_rand] = 0xFFFFFFFF
    return np.ascontiguousarray(reads, np.uint8), offsets
