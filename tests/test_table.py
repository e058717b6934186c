"""Host-side C table builder (phfpfac_amd/csrc/pfac_table.c through the C-ABI): lookup(state, byte) must equal
the oracle's dense trie in EVERY cell, for every PHF width; plus the reader's error behaviour and the blob image."""
import numpy as np
import pytest

from orc import Oracle
from phfpfac_amd import PfacError, PfacTable


def lookup_all(t):
    S, w, wb = t.state_num, t.width, t.width_bit
    key = (np.arange(S, dtype=np.int64)[:, None] << 8) + np.arange(256)[None, :]
    row, col = key >> wb, key & (w - 1)
    idx = t.r[row] + col
    ok = (idx >= 0) & (idx < t.ht_size)
    idc = np.clip(idx, 0, t.ht_size - 1)
    ok &= t.HT[idc] == row
    return np.where(ok, t.val[idc], -1)


@pytest.mark.parametrize("name", ["experimentpattern", "xaa", "xaa+xab+xac+xad", "bytefile/10000byte", "bytefile/100000byte"])
@pytest.mark.parametrize("width", [256, 1024, 4096, 64, 1])
def test_lookup_equals_dense_trie(name, width, resolve):
    t = PfacTable.from_file(resolve(name), width)
    o = Oracle(resolve(name), 1, 1)
    dense = o.trie()
    assert t.state_num == dense.shape[0] and t.num_final == o.stats()["final"]
    assert t.max_pat_len == o.L.orc_max_len(o.m)
    assert (lookup_all(t) == dense).all()
    assert (t.idmap == o.idmap()).all()
    assert (t.s0 == dense[t.num_final + 1]).all()
    assert t.max_row == t.state_num * 256 // width + 1
    # the C lookup helper agrees on a sample
    rng = np.random.default_rng(0)
    for s, c in zip(rng.integers(0, t.state_num, 200), rng.integers(0, 256, 200)):
        assert t.lookup(int(s), int(c)) == dense[s, c]
    o.close()


def test_snort_scale_set_builds_fast_and_correct(resolve):
    """75 840 patterns / 542 732 states (the reference preallocates 4 GiB per chunk and sorts rows in O(R^2))."""
    import time
    t0 = time.time()
    t = PfacTable.from_file(resolve("bytefile/1000000byte"), 256)
    assert time.time() - t0 < 20
    assert (t.n_patterns, t.state_num, t.max_pat_len) == (75840, 542732, 228)
    o = Oracle(resolve("bytefile/1000000byte"), 1, 1)
    dense = o.trie()
    rng = np.random.default_rng(1)
    S = rng.integers(0, t.state_num, 20000)
    got = lookup_all_rows(t, S)
    assert (got == dense[S]).all()
    o.close()


def lookup_all_rows(t, states):
    key = (states.astype(np.int64)[:, None] << 8) + np.arange(256)[None, :]
    row, col = key >> t.width_bit, key & (t.width - 1)
    idx = t.r[row] + col
    ok = (idx >= 0) & (idx < t.ht_size)
    idc = np.clip(idx, 0, t.ht_size - 1)
    ok &= t.HT[idc] == row
    return np.where(ok, t.val[idc], -1)


def test_blob_round_trip_and_reference_arrays(resolve):
    t = PfacTable.from_file(resolve("xad"), 1024)
    blob = t.blob()
    assert blob.dtype == np.int32 and blob[0] == 0x50464143
    u = PfacTable.from_blob(blob)
    for k in ("width", "width_bit", "n_patterns", "num_final", "state_num", "max_pat_len", "max_row", "ht_size", "n_keys"):
        assert getattr(t, k) == getattr(u, k)
    for k in ("s0", "r", "HT", "val", "idmap"):
        assert (getattr(t, k) == getattr(u, k)).all()
    with pytest.raises(PfacError):
        PfacTable.from_blob(blob[:100])
    bad = blob.copy(); bad[0] = 1
    with pytest.raises(PfacError):
        PfacTable.from_blob(bad)
    # arrays as the reference's FFDM() leaves them (exact layout from the oracle's restatement)
    o = Oracle(resolve("xad"), 1, 1)
    o.ffdm(4096, exact=True)
    st = o.stats()
    L = o.L
    w = PfacTable.from_reference_arrays(
        o.trie()[st["final"] + 1], np.ctypeslib.as_array(L.orc_phf_r(o.m, 0), (st["r_size"],)),
        np.ctypeslib.as_array(L.orc_phf_HT(o.m, 0), (st["ht_size"],)),
        np.ctypeslib.as_array(L.orc_phf_val(o.m, 0), (st["ht_size"],)), o.idmap(), 4096, st["state_num"], st["final"],
        st["ht_size"], L.orc_max_len(o.m))
    assert (lookup_all(w) == o.trie()).all()
    o.close()


def test_reader_errors_are_reported_not_fatal(tmp_path):
    """The reference exit(1)s (or runs into undefined behaviour) on these; the library returns PFAC_E_PATTERN / _ARG / _IO."""
    def build(data, width=256):
        return PfacTable.from_bytes(data, width)
    with pytest.raises(PfacError) as e:
        build(b"abc")                              # no trailing newline
    assert e.value.status == -3
    with pytest.raises(PfacError) as e:
        build(b"abc\n\nde\n")                      # empty line
    assert e.value.status == -3
    with pytest.raises(PfacError) as e:
        build(b"a" * 1023 + b"\n")                 # "Pattern 1 length over 1024.": the reference's reader counts the
    assert e.value.status == -3 and "length over 1024" in str(e.value)      # newline too (ctr.c:72-83) -> 1022 max
    assert build(b"a" * 1022 + b"\n").max_pat_len == 1022
    for w in (0, 3, 8192, -4):
        with pytest.raises(PfacError) as e:
            build(b"a\n", w)
        assert e.value.status == -1
    with pytest.raises(PfacError) as e:
        PfacTable.from_file(str(tmp_path / "missing"), 256)
    assert e.value.status == -2
    t = build(b"b\r\n\xff\x00z\n")                 # \r kept, NUL and high bytes are ordinary pattern bytes
    assert t.n_patterns == 2 and t.max_pat_len == 3


def test_emitter_format(tmp_path, resolve):
    from phfpfac_amd import RECORD_DTYPE, emit_records
    rec = np.array([(4, 0), (12, 1), (9999, 0), (10000, 1), (4000000000, 0)], dtype=RECORD_DTYPE)
    idmap = np.array([3, 77777], dtype=np.int32)
    out = tmp_path / "o.txt"
    n = emit_records(str(out), rec, idmap)
    exp = "".join("At position %4d, match pattern %d\n" % (p, idmap[s]) for p, s in rec.tolist()).encode()
    assert out.read_bytes() == exp and n == len(exp)
    n2 = emit_records(str(out), rec[:2], idmap, base=(1 << 33), append=True)   # 64-bit positions, appended
    tail = "".join("At position %4d, match pattern %d\n" % ((1 << 33) + p, idmap[s]) for p, s in rec[:2].tolist()).encode()
    assert out.read_bytes() == exp + tail and n2 == len(tail)


def test_parallel_emitter_is_byte_identical(tmp_path):
    """pfac_emit_records_mt (size pass + prefix + parallel pwrite) == the serial emitter, also when appending."""
    from phfpfac_amd import RECORD_DTYPE, emit_records
    rng = np.random.default_rng(5)
    n = 700_000
    rec = np.empty(n, dtype=RECORD_DTYPE)
    rec["pos"] = np.sort(rng.integers(0, 2**32 - 1, n, dtype=np.uint64)).astype(np.uint32)
    rec["state"] = rng.integers(0, 5000, n)
    idmap = rng.integers(1, 2_000_000, 5000).astype(np.int32)
    a, b = tmp_path / "a.txt", tmp_path / "b.txt"
    for base in (0, 1 << 33):
        na = emit_records(str(a), rec, idmap, base=base, append=base != 0)
        nb = emit_records(str(b), rec, idmap, base=base, append=base != 0, threads=5)
        assert na == nb
    assert a.read_bytes() == b.read_bytes()


ESCAPED = (b"tab\\there\n"            # \t
           b"nl\\nin\\\\side\n"   # \n and \\ inside one pattern
           b"hex\\x41\\xfe\\x7\n"  # \x41 \xfe \x7
           b"oct\\101\\7\\377\\0end\n"   # \101 \7 \377 \0
           b"odd\\8\\q\\'\\\"\n"        # \8 (no octal digit -> byte 0, then '8'), \q (literal backslash), \' \"
           b"plain\n")


def test_escape_aware_reader(tmp_path):
    """pfac_table_build_file_escaped == the reference's read_pattern_ext/fgetc_ext (dead code there, restated in the
    oracle and -- when oracle/_ref is built -- called for real): same tries, same ids, newline bytes inside patterns."""
    import ctypes as C
    import os
    pf = tmp_path / "esc"
    pf.write_bytes(ESCAPED)
    t = PfacTable.from_file(str(pf), 256, escapes=True)
    o = Oracle(str(pf), 1, 1, escapes=True)
    dense = o.trie()
    assert t.n_patterns == 6 and t.state_num == dense.shape[0]
    assert (lookup_all(t) == dense).all() and (t.idmap == o.idmap()).all()
    # spot-check the unescaped bytes through the automaton: walk "nl\nin\\side" from the root
    s = t.num_final + 1
    for ch in b"nl\nin\\side":
        s = t.lookup(s, ch)
        assert s >= 0
    assert s < t.num_final and t.idmap[s] == 2
    for pat, pid in ((b"hexA\xfe\x07", 3), (b"octA\x07\xff\x00end", 4), (b"odd\x008\\q'\"", 5), (b"tab\there", 1)):
        s = t.num_final + 1
        for ch in pat:
            s = t.lookup(s, ch)
            assert s >= 0, (pat, ch)
        assert t.idmap[s] == pid
    # the plain reader sees the same file as 6 different (longer) patterns
    assert PfacTable.from_file(str(pf), 256).max_pat_len > t.max_pat_len
    ref_lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "libpfacref.so")
    if os.path.exists(ref_lib):
        ref = C.CDLL(ref_lib)
        ref.ref_build_single_ext.restype = C.c_void_p; ref.ref_build_single_ext.argtypes = [C.c_char_p]
        ref.orc_state_num.argtypes = [C.c_void_p, C.c_int]
        ref.orc_trie_row.restype = C.POINTER(C.c_int); ref.orc_trie_row.argtypes = [C.c_void_p, C.c_int, C.c_int]
        ref.orc_idmap.restype = C.POINTER(C.c_int); ref.orc_idmap.argtypes = [C.c_void_p, C.c_int]
        b = ref.ref_build_single_ext(os.fsencode(str(pf)))
        assert ref.orc_state_num(b, 0) == dense.shape[0]
        for st in range(dense.shape[0]):
            assert (np.ctypeslib.as_array(ref.orc_trie_row(b, 0, st), (256,)) == dense[st]).all()
        assert (np.ctypeslib.as_array(ref.orc_idmap(b, 0), (6,)) == o.idmap()).all()
    o.close()


def test_escape_reader_fuzz_against_the_reference_reader(tmp_path):
    """Differential fuzz of the in-memory escape parser (pfac_table.c: getc_escaped / scan_escape_number) against the
    reference's own read_pattern_ext + fgetc_ext, which scan \\ooo and \\xNN with fscanf on the stream (compiled from
    the reference sources into oracle/_ref): pattern files full of backslashes, digits beyond 7, 'x', signs, "0x"
    prefixes and white space after "\\x" -- the inputs on which a hand-written parser and libc's scanf disagree."""
    import ctypes as C
    import os
    ref_lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "libpfacref.so")
    if not os.path.exists(ref_lib):
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    ref = C.CDLL(ref_lib)
    ref.ref_build_single_ext.restype = C.c_void_p; ref.ref_build_single_ext.argtypes = [C.c_char_p]
    ref.orc_state_num.argtypes = [C.c_void_p, C.c_int]
    ref.orc_num_patterns.argtypes = [C.c_void_p]
    ref.orc_trie_row.restype = C.POINTER(C.c_int); ref.orc_trie_row.argtypes = [C.c_void_p, C.c_int, C.c_int]
    ref.orc_idmap.restype = C.POINTER(C.c_int); ref.orc_idmap.argtypes = [C.c_void_p, C.c_int]
    rng = np.random.default_rng(20261004)
    alphabet = [b"\\", b"\\", b"\\", b"x", b"x", b"0", b"1", b"7", b"8", b"9", b"a", b"f", b"g", b"A", b"F", b"q", b"n", b"t",
                b" ", b"\t", b"\r", b"+", b"-", b"X", b"'", b"\"", b"3", b"5", b"z", b"\xfe", b"\x00"]
    for trial in range(60):
        lines = []
        for _ in range(int(rng.integers(1, 12))):
            k = int(rng.integers(1, 14))
            line = b"".join(alphabet[int(i)] for i in rng.integers(0, len(alphabet), k))
            lines.append(line + b"\n")
        if trial % 5 == 0:
            lines.insert(0, b"\\x\n41\\x \t7q\\0x41\\x0x41\\+7\\x-f\\x+\n")     # white space / sign / prefix after an escape
        img = b"".join(lines)
        if img.endswith(b"\\\n") or img.endswith(b"\\x\n"):            # an escape that swallows the final newline: the
            img += b"end\n"                                           # reference then reads past EOF and exits
        pf = tmp_path / f"fz{trial}"
        pf.write_bytes(img)
        b = ref.ref_build_single_ext(os.fsencode(str(pf)))
        n_pat = ref.orc_num_patterns(b)
        S = ref.orc_state_num(b, 0)
        t = PfacTable.from_file(str(pf), 256, escapes=True)
        assert t.n_patterns == n_pat and t.state_num == S, (trial, img)
        dense = np.stack([np.ctypeslib.as_array(ref.orc_trie_row(b, 0, st), (256,)) for st in range(S)])
        assert (lookup_all(t) == dense).all(), (trial, img)
        assert (np.ctypeslib.as_array(ref.orc_idmap(b, 0), (n_pat,)) == t.idmap).all(), (trial, img)


def test_emit_from_record_heap_and_tile_index(tmp_path):
    """pfac_emit_packed prints the compact device form -- 32-bit words scattered over a heap with gaps, ordered only by
    the tile index -- to the same bytes as the 8-byte-record emitters, serial and multi-threaded."""
    from phfpfac_amd import RECORD_DTYPE, emit_packed, emit_records
    rng = np.random.default_rng(11)
    n_tiles = 1600
    counts = rng.integers(0, 1200, n_tiles)
    counts[rng.random(n_tiles) < 0.3] = 0                    # empty tiles
    order = rng.permutation(n_tiles)                         # where each tile's run sits in the heap: any order, with gaps
    first = np.zeros(n_tiles, dtype=np.uint64)
    at = 0
    for t in order:
        at += int(rng.integers(0, 9))
        first[t] = at
        at += int(counts[t])
    words = rng.integers(0, 1 << 32, at + 5, dtype=np.uint64).astype(np.uint32)     # garbage in the gaps
    idmap = rng.integers(1, 3_000_000, 1 << 20).astype(np.int32)
    rec = np.empty(int(counts.sum()), dtype=RECORD_DTYPE)
    k = 0
    for t in range(n_tiles):
        c = int(counts[t])
        pos = np.sort(rng.integers(0, 4096, c)).astype(np.uint32)
        st = rng.integers(0, 1 << 20, c).astype(np.uint32)
        words[int(first[t]): int(first[t]) + c] = pos | (st << np.uint32(12))
        rec["pos"][k: k + c] = pos + np.uint32(t * 4096)
        rec["state"][k: k + c] = st
        k += c
    tix = first | (counts.astype(np.uint64) << np.uint64(40))
    assert rec.size > 4 * (1 << 17)                          # enough for the parallel emitter (it falls back to serial below that)
    a, b, c_ = tmp_path / "a.txt", tmp_path / "b.txt", tmp_path / "c.txt"
    for base in (0, 5 << 32):
        na = emit_records(str(a), rec, idmap, base=base)
        nb = emit_packed(str(b), words, tix, idmap, base=base)
        nc = emit_packed(str(c_), words, tix, idmap, base=base, threads=6)
        assert na == nb == nc
        assert a.read_bytes() == b.read_bytes() == c_.read_bytes()
    # the 16-bit form (automata with at most 16 final states): same tiles, states folded to 4 bits
    rec16 = rec.copy()
    rec16["state"] &= 15
    words16 = (words & np.uint32(0xFFFF)).astype(np.uint16)
    na = emit_records(str(a), rec16, idmap)
    for threads in (1, 6):
        assert emit_packed(str(b), words16, tix, idmap, threads=threads) == na
        assert a.read_bytes() == b.read_bytes()
    # no tiles at all / only empty tiles
    assert emit_packed(str(b), np.zeros(1, np.uint32), np.zeros(0, np.uint64), idmap) == 0
    assert emit_packed(str(b), np.zeros(1, np.uint32), np.zeros(9, np.uint64), idmap) == 0
    # a tile index that points past the heap copy it is paired with (an index copied after an overflowed scan, a short
    # copy): refused, nothing read out of bounds
    need = int((first + counts.astype(np.uint64))[counts > 0].max())
    assert emit_packed(str(b), words[:need], tix, idmap) == emit_records(str(a), rec, idmap)
    for short in (words[: need - 1], words[:1]):
        with pytest.raises(PfacError):
            emit_packed(str(b), short, tix, idmap)
