"""The oracle (CPU restatement, oracle/pfac_oracle.c) is pinned here -- before anything trusts it:
* against the golden outputs produced by the reference's own host sources (tests/golden/fingerprints.json,
  made by tests/golden/make_golden.py through oracle/_ref),
* against the table statistics and the duplicate rule that the reference's recorded run logs / outputs hold,
* against an independent algorithm (serial Aho-Corasick with failure links) as a set of (start, id) pairs,
* and, when oracle/_ref/libpfacref.so is present, array-for-array against the reference-built tables.
"""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

from orc import Oracle, lib, match_checksum

HERE = os.path.dirname(os.path.abspath(__file__))
FP = json.load(open(os.path.join(HERE, "golden", "fingerprints.json")))
REF_LIB = os.path.join(os.path.dirname(HERE), "oracle", "_ref", "libpfacref.so")


def test_1m_fixture_regenerates(resolve):
    blob = open(resolve("1M"), "rb").read()
    assert len(blob) == FP["inputs"]["1M"]["bytes"]
    assert hashlib.md5(blob).hexdigest() == FP["inputs"]["1M"]["md5"]
    assert hashlib.md5(open(resolve("xaa+xab+xac+xad"), "rb").read()).hexdigest() == FP["inputs"]["xaa+xab+xac+xad"]["md5"]


@pytest.mark.parametrize("case", sorted(FP["cases"]))
def test_oracle_reproduces_reference_outputs(case, resolve, tmp_path):
    """Tile-faithful restatement (4*streamnum pattern chunks, exact FFDM layout, 4096+512-byte tiles, host merge)
    -> byte-identical GPU_match_result.txt.  Also: the un-tiled 'spec' walk gives the same file (parity domain)."""
    c = FP["cases"][case]
    data = open(resolve(c["input"]), "rb").read()[:-1]         # main.cc:138
    o = Oracle(resolve(c["pattern"]), c["streams"], 4)
    o.ffdm(c["width"], exact=True)
    out = tmp_path / "o.txt"
    cnt, nbytes = o.emit(data, str(out))
    blob = out.read_bytes()
    assert (cnt, nbytes, hashlib.md5(blob).hexdigest()) == (c["lines"], c["bytes"], c["md5"])
    if c["verbatim"]:
        assert blob == open(os.path.join(HERE, "golden", "out", case + ".txt"), "rb").read()
    cnt2, _ = o.emit(data, str(out), spec=True)
    assert cnt2 == c["lines"] and hashlib.md5(out.read_bytes()).hexdigest() == c["md5"]
    o.close()


# (pattern set, width) -> statistics printed by the reference's recorded runs (single automaton):
#   tmp.dat:2-12, experiment/{xaa,xab,xac,xad}record:2-12, experiment/englishdicall:2-12
RECORDED = {
    ("experimentpattern", 1024): dict(state_num=6, final=4, keys=4, max_key=1377, r_size=2, max_offset=0, ht_size=513),
    ("xaa", 4096): dict(state_num=7983, final=2600, keys=7978, max_key=2043502, r_size=499),
    ("xab", 4096): dict(state_num=7844, final=2600, keys=7837, max_key=2007929, r_size=491),
    ("xac", 4096): dict(state_num=7662, final=2600, keys=7656, max_key=1961316, r_size=479),
    ("xad", 4096): dict(state_num=493, final=189, keys=491, max_key=126073, r_size=31, max_offset=0, ht_size=3939),
    ("xaa+xab+xac+xad", 4096): dict(state_num=23963, final=7989, keys=23949, max_key=6134393, r_size=1498),
}


@pytest.mark.parametrize("name,width", sorted(RECORDED))
def test_table_statistics_match_recorded_logs(name, width, resolve):
    """state num / #keys / max key / r size as logged by the reference.  (Max Offset / Hash table size of the
    three large logs come from an OLDER build and differ from what the CURRENT reference source computes --
    oracle/_ref shows that, see test_oracle_equals_reference_host_code -- so they are pinned only where the
    current source agrees: experimentpattern and xad.)"""
    o = Oracle(resolve(name), 1, 1)
    o.ffdm(width, exact=True)
    st = o.stats()
    for k, v in RECORDED[(name, width)].items():
        assert st[k] == v, (k, st[k], v)
    o.close()


def test_duplicate_rule_from_recorded_output(resolve):
    """experiment/GPU_match_resultxab.txt reports 'may' (lines 1776 and 1777 of xab) as pattern 1777:
    ids are 1-based line numbers and the LAST duplicate wins."""
    lines = open(resolve("xab"), "rb").read().split(b"\n")
    assert lines[1775] == b"may" and lines[1776] == b"may"
    o = Oracle(resolve("xab"), 1, 1)
    pos, ids = o.scan_spec(b"xx may yy")
    assert (3, 1777) in set(zip(pos.tolist(), ids.tolist())) and 1776 not in ids.tolist()
    o.close()


def test_quirks(tmp_path):
    """SURVEY.md 8(c) probes: intra-chunk duplicates -> last line wins; order at one position = ascending length;
    last input byte is the caller's business (N = filesize-1 is applied by the CLI, not by the scan)."""
    p = tmp_path / "p"
    p.write_bytes(b"ab\nab\nabc\n")
    o = Oracle(str(p), 1, 1)
    pos, ids = o.scan_spec(b"abcab")
    assert list(zip(pos.tolist(), ids.tolist())) == [(0, 2), (0, 3), (3, 2)]
    o.close()
    p.write_bytes(b"aaaa\naa\na\naaa\n")
    o = Oracle(str(p), 1, 4)
    o.ffdm(256)
    pos, ids = o.scan_reference(b"aaaaaaaaaaaais a a a a a ")
    assert ids[:4].tolist() == [3, 2, 4, 1] and pos[:4].tolist() == [0, 0, 0, 0]
    o.close()


@pytest.mark.parametrize("name", ["experimentpattern", "xaa", "xaa+xab+xac+xad", "bytefile/10000byte"])
def test_pfac_equals_serial_aho_corasick_as_a_set(name, resolve):
    """Independent algorithm: failure-link DFA, (end, id) -> (start, id); same SET of matches."""
    L = lib()
    data = np.frombuffer(open(resolve("1M"), "rb").read()[:200000], dtype=np.uint8)
    if name.startswith("bytefile"):
        data = np.frombuffer(open(resolve("bytefile/100000byte"), "rb").read()[:200000], dtype=np.uint8)
    o = Oracle(resolve(name), 1, 1)
    pos, ids = o.scan_spec(data)
    ac = L.ac_build(o.m)
    m = L.orc_matches_new()
    n = L.ac_scan_collect(ac, data.ctypes.data, data.size, m)
    apos = np.ctypeslib.as_array(L.orc_matches_pos(m), (max(n, 1),))[:n].copy()
    aid = np.ctypeslib.as_array(L.orc_matches_id(m), (max(n, 1),))[:n].copy()
    chk = C.c_uint64(0)
    n2 = L.ac_scan_count(ac, data.ctypes.data, data.size, C.byref(chk))
    L.orc_matches_free(m)
    L.ac_free(ac)
    assert n == n2 == pos.size
    assert sorted(zip(apos.tolist(), aid.tolist())) == sorted(zip(pos.tolist(), ids.tolist()))
    assert chk.value == match_checksum(pos, ids)
    o.close()


@pytest.mark.skipif(not os.path.exists(REF_LIB), reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("name,streams,width", [("experimentpattern", 1, 256), ("xaa", 2, 1024), ("bytefile/10000byte", 1, 4096)])
def test_oracle_equals_reference_host_code(name, streams, width, resolve):
    """Array-for-array: tries, id maps, r / HT / val of the reference's REAL create_table_reorder.c + phf.c
    (compiled where they lie, oracle/ref_harness.cc) == the oracle's restatement."""
    ref = C.CDLL(REF_LIB)
    L = lib()
    for lb in (ref,):
        lb.ref_build.restype = C.c_void_p; lb.ref_build.argtypes = [C.c_char_p, C.c_int]
        lb.ref_ffdm.argtypes = [C.c_void_p, C.c_int]
        for f in ("orc_num_chunks",):
            getattr(lb, f).argtypes = [C.c_void_p]
        for f in ("orc_state_num", "orc_final_num", "orc_chunk_max_len"):
            getattr(lb, f).argtypes = [C.c_void_p, C.c_int]
        lb.orc_trie_row.restype = C.POINTER(C.c_int); lb.orc_trie_row.argtypes = [C.c_void_p, C.c_int, C.c_int]
        lb.orc_idmap.restype = C.POINTER(C.c_int); lb.orc_idmap.argtypes = [C.c_void_p, C.c_int]
        lb.orc_phf_stat.argtypes = [C.c_void_p, C.c_int, C.c_int]
        for f in ("orc_phf_r", "orc_phf_HT", "orc_phf_val"):
            getattr(lb, f).restype = C.POINTER(C.c_int); getattr(lb, f).argtypes = [C.c_void_p, C.c_int]
    o = Oracle(resolve(name), streams, 4)
    o.ffdm(width, exact=True)
    b = ref.ref_build(os.fsencode(resolve(name)), streams)
    ref.ref_ffdm(b, width)
    assert ref.orc_num_chunks(b) == o.P
    for c in range(o.P):
        S = L.orc_state_num(o.m, c)
        assert S == ref.orc_state_num(b, c) and L.orc_final_num(o.m, c) == ref.orc_final_num(b, c)
        for s in range(S):
            assert (np.ctypeslib.as_array(L.orc_trie_row(o.m, c, s), (256,)) ==
                    np.ctypeslib.as_array(ref.orc_trie_row(b, c, s), (256,))).all()
        nf = L.orc_final_num(o.m, c)
        if nf:
            assert (o.idmap(c) == np.ctypeslib.as_array(ref.orc_idmap(b, c), (nf,))).all()
        hs = L.orc_phf_stat(o.m, c, 4)
        assert hs == ref.orc_phf_stat(b, c, 4)
        mr = L.orc_phf_stat(o.m, c, 3)
        for f in ("orc_phf_HT", "orc_phf_val"):
            if hs:
                assert (np.ctypeslib.as_array(getattr(L, f)(o.m, c), (hs,)) == np.ctypeslib.as_array(getattr(ref, f)(b, c), (hs,))).all()
        assert (np.ctypeslib.as_array(L.orc_phf_r(o.m, c), (mr,)) == np.ctypeslib.as_array(ref.orc_phf_r(b, c), (mr,))).all()
    o.close()
