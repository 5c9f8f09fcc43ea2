"""Pins the CPU oracle to the reference: every vector in tests/golden/golden.json
was produced by running the unmodified reference (tests/golden/make_golden.py);
the oracle must reproduce all of them.  Runs without a GPU and without the
reference tree."""
import numpy as np
import pytest

from helpers import GOLDEN, GENOME_SEED, READS_SEED, file_sha, mms_patterns, probe_rows, sha


def test_tiny40_every_value(oracle):
    g = GOLDEN["tiny40"]
    text = np.frombuffer(bytes.fromhex(g["text_hex"]), np.uint8).copy()
    n = g["n"]
    sa = oracle.histsort(text, n)
    assert sa.tolist() == g["sa"]
    assert oracle.sa_doubling(text, n).tolist() == g["sa"]
    f = oracle.make_fmi(text, n)
    assert (f.endloc, f.C) == (g["endloc"], g["C"])
    assert f.bwt.tobytes().hex() == g["bwt_hex"]
    assert f.idxs.tolist() == g["idxs"]
    assert f.rank_rows.tolist() == g["rank_rows"]
    assert list(oracle.loc_search(f, [0, 1, 2])) == g["loc_search_ACG"]
    assert oracle.reverse_search(f, [0, 1, 2]) == g["reverse_search_ACG"]
    assert [oracle.unc_sa(f, i) for i in range(41)] == g["unc_sa"] == g["sa"]
    assert [oracle.lf(f, i) for i in range(41)] == g["lf"]
    assert [[oracle.rank(f, c, i) for i in range(42)] for c in range(4)] == g["rank"]


def test_lookup_table(oracle):
    assert sha(oracle.lookup_table()) == GOLDEN["lookup_table"]


@pytest.mark.parametrize("rec", GOLDEN["random"], ids=lambda r: f"n{r['n']}-m{r.get('m', 0)}")
def test_random_genomes(oracle, rec, tmp_path):
    n = rec["n"]
    text = oracle.gen_genome(n, GENOME_SEED)
    sa = oracle.histsort(text, n)
    assert sha(sa) == rec["sa"]
    assert oracle.check_sa(text, n, sa) == 0
    f = oracle.make_fmi(text, n, sa)
    assert (f.endloc, f.C) == (rec["endloc"], rec["C"])
    assert sha(f.bwt) == rec["bwt"] and sha(f.idxs) == rec["idxs"] and sha(f.rank_rows) == rec["rank_rows"]
    path = str(tmp_path / "x.idx")
    assert oracle.write_index(f, path) == 0
    assert file_sha(path) == (rec["index_file"], rec["index_file_bytes"])
    assert rec["index_file_bytes"] == 28 + 4 * (1 + n // 32) + (n + 3) // 4
    f2 = oracle.read_index(path)
    assert (f2.len, f2.endloc, f2.C) == (n, f.endloc, f.C) and sha(f2.rank_rows) == rec["rank_rows"]
    rows = probe_rows(rec)
    pos, _ = oracle.locate_rows(f, rows)
    assert sha(pos) == rec["unc_sa"]
    assert sha(np.array([oracle.lf(f, int(r)) for r in rows], np.uint32)) == rec["lf"]
    if "R" not in rec:
        return
    R, m = rec["R"], rec["m"]
    reads, _ = oracle.gen_reads(text, n, R, m, READS_SEED)
    assert sha(reads) == rec["reads"]
    sp, ep = oracle.search_batch(f, reads, m, m, R)
    assert sha(sp) == rec["sp"] and sha(ep) == rec["ep"]
    assert sha(np.array([oracle.reverse_search(f, reads[r]) for r in range(R)], np.int32)) == rec["reverse_search"]
    loc = [oracle.locate(f, reads[r]) if ep[r] - sp[r] == 1 else -2 for r in range(R)]
    assert sha(np.array(loc, np.int32)) == rec["locate_unique"]
    pats = mms_patterns(oracle, text, n, 300, rec["mms_m"], 17)
    assert [list(oracle.mms(f, pats[r])) for r in range(300)] == rec["mms"]


@pytest.mark.parametrize("rec", GOLDEN["tandem"], ids=lambda r: f"n{r['n']}-p{r['period']}-mut{r['mut_every']}")
def test_tandem_repeats(oracle, rec):
    n = rec["n"]
    text = oracle.gen_tandem(n, rec["period"], 11, rec["mut_every"], 3)
    sa = oracle.sa_doubling(text, n)
    assert sha(sa) == rec["sa"]
    if n <= 20000:
        assert sha(oracle.histsort(text, n)) == rec["sa"]
    else:
        # the bucket sort gives up at a depth bound instead of overflowing a stack
        assert oracle.histsort(text, n, max_depth=512) is None
    f = oracle.make_fmi(text, n, sa)
    assert (f.endloc, f.C) == (rec["endloc"], rec["C"])
    assert sha(f.bwt) == rec["bwt"] and sha(f.idxs) == rec["idxs"] and sha(f.rank_rows) == rec["rank_rows"]


def test_four_thread_histsort(oracle):
    rec = GOLDEN["threaded"]
    n = rec["n"]
    text = oracle.gen_genome(n, GENOME_SEED)
    sa = oracle.histsort(text, n, threads=4)
    assert sha(sa) == rec["sa"]
    f = oracle.make_fmi(text, n, sa)
    assert (f.endloc, f.C, sha(f.bwt), sha(f.idxs)) == (rec["endloc"], rec["C"], rec["bwt"], rec["idxs"])


def test_packer_quirk(oracle):
    g = GOLDEN["packer_ACGTTTT"]
    packed = oracle.pack_ascii(b"ACGTTTT")
    assert packed.tobytes().hex() == g["packed_hex"] == "1b00"  # TTT at the end is lost
    f = oracle.make_fmi(packed, 7)
    assert (f.C, f.endloc, f.idxs.tolist(), f.bwt.tobytes().hex()) == (g["C"], g["endloc"], g["idxs"], g["bwt_hex"])
    g = GOLDEN["packer_1003"]
    ascii_text = g["ascii"].encode("latin-1")
    packed = oracle.pack_ascii(ascii_text)
    assert sha(packed) == g["packed"]
    f = oracle.make_fmi(packed, len(ascii_text))
    assert (f.C, f.endloc, sha(f.bwt)) == (g["C"], g["endloc"], g["bwt"])


def test_check_sa_rejects_wrong_arrays(oracle):
    n = 5000
    text = oracle.gen_genome(n, 3)
    sa = oracle.histsort(text, n)
    assert oracle.check_sa(text, n, sa) == 0
    bad = sa.copy()
    bad[[10, 11]] = bad[[11, 10]]
    assert oracle.check_sa(text, n, bad) != 0
    bad = sa.copy()
    bad[7] = bad[8]
    assert oracle.check_sa(text, n, bad) != 0
