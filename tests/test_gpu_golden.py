"""The HIP path against the committed golden vectors (outputs of the unmodified
reference, tests/golden/make_golden.py): bit-exact SA, BWT, sampled SA, rank
rows, C[], index file bytes, SA intervals, reverse_search values, locate
positions, unc_sa and lf."""
import numpy as np
import pytest
import torch

from helpers import GOLDEN, GENOME_SEED, READS_SEED, probe_rows, sha

pytestmark = pytest.mark.gpu


def _dev(a):
    t = torch.zeros(a.nbytes + 64, dtype=torch.uint8, device="cuda")
    t[: a.nbytes] = torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1))
    return t


def _check_index(gpu, ix, rec):
    assert (ix.endloc, ix.C) == (rec["endloc"], rec["C"])
    bwt, idxs = ix.export()
    assert sha(bwt) == rec["bwt"] and sha(idxs) == rec["idxs"]
    assert sha(ix.rank_rows16()) == rec["rank_rows"]
    # the bytes write_index stores (fileio.c:11-15)
    n = ix.n
    blob = (np.array([n] + ix.C + [ix.endloc], np.uint32).tobytes() + idxs.tobytes() + bwt.tobytes())
    import hashlib
    assert hashlib.sha256(blob).hexdigest() == rec["index_file"] and len(blob) == rec["index_file_bytes"]


def test_tiny40(gpu):
    g = GOLDEN["tiny40"]
    text = np.frombuffer(bytes.fromhex(g["text_hex"]), np.uint8).copy()
    n = g["n"]
    sa = gpu.to_numpy_u32(gpu.sa_build(_dev(text), n))
    assert sa.tolist() == g["sa"]
    ix = gpu.build_index(text, n)
    assert (ix.endloc, ix.C) == (g["endloc"], g["C"])
    bwt, idxs = ix.export()
    assert bwt.tobytes().hex() == g["bwt_hex"] and idxs.tolist() == g["idxs"]
    assert ix.rank_rows16().tolist() == g["rank_rows"]
    sp, ep = ix.search(_dev(np.array([0, 1, 2], np.uint8)), 3, 1)
    assert [int(sp.item()), int(ep.item())] == g["loc_search_ACG"]
    rows = torch.arange(41, dtype=torch.int32, device="cuda")
    assert gpu.to_numpy_u32(ix.locate_rows(rows)).tolist() == g["unc_sa"]


@pytest.mark.parametrize("rec", GOLDEN["random"], ids=lambda r: f"n{r['n']}-m{r.get('m', 0)}")
def test_random_genomes(oracle, gpu, rec):
    from bwt_amd import _lib
    import ctypes as C
    n = rec["n"]
    text = oracle.gen_genome(n, GENOME_SEED)
    assert sha(gpu.to_numpy_u32(gpu.sa_build(_dev(text), n))) == rec["sa"]
    ix = gpu.build_index(text, n)
    _check_index(gpu, ix, rec)
    rows = probe_rows(rec)
    d_rows = torch.from_numpy(rows.view(np.int32)).cuda()
    assert sha(gpu.to_numpy_u32(ix.locate_rows(d_rows))) == rec["unc_sa"]
    lf = torch.empty_like(d_rows)
    _lib.check(_lib.load().bwtgpu_lf_batch(ix.h, d_rows.data_ptr(), len(rows), lf.data_ptr(), None))
    assert sha(gpu.to_numpy_u32(lf)) == rec["lf"]
    if "R" not in rec:
        return
    R, m = rec["R"], rec["m"]
    reads, _ = oracle.gen_reads(text, n, R, m, READS_SEED)
    sp, ep, lr = ix.search(_dev(reads), m, R, want_locrow=True)
    sp, ep, lr = (gpu.to_numpy_u32(x) for x in (sp, ep, lr))
    assert sha(sp) == rec["sp"] and sha(ep) == rec["ep"]
    rs = np.where(lr == 0xFFFFFFFF, 0, (ep - sp + 1).astype(np.int64)).astype(np.int32)
    assert sha(rs) == rec["reverse_search"]
    pos = gpu.to_numpy_u32(ix.locate_rows(torch.from_numpy(lr.view(np.int32)).cuda())).astype(np.int64)
    pos[pos == 0xFFFFFFFF] = -1
    loc = np.where(ep - sp == 1, pos, -2).astype(np.int32)
    assert sha(loc) == rec["locate_unique"]


@pytest.mark.parametrize("rec", GOLDEN["tandem"], ids=lambda r: f"n{r['n']}-p{r['period']}-mut{r['mut_every']}")
def test_tandem_repeats(oracle, gpu, rec):
    n = rec["n"]
    text = oracle.gen_tandem(n, rec["period"], 11, rec["mut_every"], 3)
    assert sha(gpu.to_numpy_u32(gpu.sa_build(_dev(text), n))) == rec["sa"]
    _check_index(gpu, gpu.build_index(text, n), rec)


def test_twelve_million(gpu):
    """the size at which the reference switches to its four-thread path"""
    rec = GOLDEN["threaded"]
    n = rec["n"]
    t = gpu.gen_genome(n, GENOME_SEED)
    assert sha(gpu.to_numpy_u32(gpu.sa_build(t, n))) == rec["sa"]
    _check_index(gpu, gpu.build_index_dev(t, n), rec)
