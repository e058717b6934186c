"""Character-class patterns (second half): pfac_table_build_*_charclass -- parser, subset
construction, PFAC numbering, multi-pattern final states -- against the independent brute-force matcher
oracle/charclass_oracle.py.  Parity with the reference is UNPINNED here (its char-class code cannot be built and it
holds no fixture for it); the grammar is restated from charset_table_reorder.c:131-168."""
import importlib.util
import os

import numpy as np
import pytest

from phfpfac_amd import PfacError, PfacTable

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_spec = importlib.util.spec_from_file_location("charclass_oracle", os.path.join(REPO, "oracle", "charclass_oracle.py"))
cco = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(cco)

PATTERNS = (b"[a-c]x\n"            # a class and a literal
            b"ax\n"                # overlaps pattern 1 on "ax": one final state, two patterns
            b"[^a-z0-9 ]\n"        # negated class, one byte
            b"q[0-9][0-9]\n"       # two classes
            b"[a-c]\n"             # a prefix of pattern 1's language
            b"\\x41[\\x42-\\x44]\\n\n"   # escapes outside and inside a class, pattern ends in a newline BYTE
            b"[-a]z\n"             # leading '-' is a literal
            b"ax[xy]\n"            # extends pattern 2
            b"[a-c]x\n")           # duplicate of pattern 1: both ids are reported


def cpu_walk(t, data):
    """The PFAC walk on the host table (the device lookup contract), expanded through the outputs lists."""
    pos, ids = [], []
    n = data.size
    root = t.num_final + 1
    for i in range(n):
        s = root
        for j in range(i, n):
            s = t.lookup(s, int(data[j]))
            if s < 0:
                break
            if s < t.num_final:
                for k in range(t.out_first[s], t.out_first[s + 1]):
                    pos.append(i)
                    ids.append(int(t.out_ids[k]))
    return np.array(pos, dtype=np.int64), np.array(ids, dtype=np.int32)


def test_charclass_table_matches_brute_force():
    t = PfacTable.from_charclass(PATTERNS, 256)
    assert t.n_patterns == 9 and t.state_num >= t.num_final + 2
    assert t.max_pat_len == 3
    rng = np.random.default_rng(3)
    alphabet = np.frombuffer(b"abcxyzq0123456789 AB\nCD-Z!", dtype=np.uint8)
    data = alphabet[rng.integers(0, alphabet.size, 6000)]
    want_pos, want_ids = cco.match(PATTERNS, data)
    got_pos, got_ids = cpu_walk(t, data)
    assert want_pos.size > 2000
    np.testing.assert_array_equal(got_pos, want_pos)
    np.testing.assert_array_equal(got_ids, want_ids)
    # the overlap: "ax" ends patterns 1, 2 and 9 in one state
    multi = [list(t.out_ids[t.out_first[s]: t.out_first[s + 1]]) for s in range(t.num_final)]
    assert [1, 2, 9] in multi
    assert all(t.idmap[s] == multi[s][0] for s in range(t.num_final))


@pytest.mark.parametrize("width", [64, 256, 4096])
def test_charclass_fuzz_against_brute_force(width):
    rng = np.random.default_rng(100 + width)
    atoms = [b"a", b"b", b"c", b"z", b"0", b"[ab]", b"[^a]", b"[a-c]", b"[0-9a]", b"\\x61", b"[\\x61-\\x63]", b"[^0-9]", b"[b-]", b"\\101"]
    for trial in range(40):
        lines = []
        for _ in range(int(rng.integers(1, 9))):
            lines.append(b"".join(atoms[int(k)] for k in rng.integers(0, len(atoms), int(rng.integers(1, 5)))) + b"\n")
        img = b"".join(lines)
        try:
            cco.parse(img)
        except ValueError:                      # e.g. "[b-]x": the ']' is taken as the range's right end, the class never closes
            with pytest.raises(PfacError):
                PfacTable.from_charclass(img, width)
            continue
        t = PfacTable.from_charclass(img, width)
        data = np.frombuffer(b"abcz09A-]", dtype=np.uint8)[rng.integers(0, 9, 1500)]
        want_pos, want_ids = cco.match(img, data)
        got_pos, got_ids = cpu_walk(t, data)
        np.testing.assert_array_equal(got_pos, want_pos, err_msg=repr(img))
        np.testing.assert_array_equal(got_ids, want_ids, err_msg=repr(img))


def test_charclass_reader_errors():
    for bad in (b"abc", b"a\n\nb\n", b"[abc\n", b"x[a-\n"):
        with pytest.raises(PfacError) as e:
            PfacTable.from_charclass(bad, 256)
        assert e.value.status == -3


@pytest.mark.gpu
def test_charclass_table_scanned_on_the_gpu(tmp_path):
    """The same table on the GPU (the scan kernel is unchanged: it walks lookup(state, byte)); records expanded
    through the outputs lists == the brute-force matcher; the text through pfac_emit_records_multi."""
    from phfpfac_amd import GpuMatcher, emit_records_multi
    pf = tmp_path / "cc.pat"
    pf.write_bytes(PATTERNS)
    t = PfacTable.from_charclass(str(pf), 256)
    rng = np.random.default_rng(9)
    alphabet = np.frombuffer(b"abcxyzq0123456789 AB\nCD-Z!", dtype=np.uint8)
    data = alphabet[rng.integers(0, alphabet.size, 300_000)]
    with GpuMatcher(0, 1) as g:
        g.load_table(t)
        rec = g.scan_bytes(data)
    want_pos, want_ids = cco.match(PATTERNS, data)
    cnt = (t.out_first[1:] - t.out_first[:-1])[rec["state"]]
    got_pos = np.repeat(rec["pos"].astype(np.int64), cnt)
    got_ids = np.concatenate([t.out_ids[t.out_first[s]: t.out_first[s + 1]] for s in rec["state"]]) if rec.size else np.empty(0, np.int32)
    np.testing.assert_array_equal(got_pos, want_pos)
    np.testing.assert_array_equal(got_ids, want_ids)
    out = tmp_path / "cc.txt"
    emit_records_multi(str(out), rec, t)
    want = "".join("At position %4d, match pattern %d\n" % (p, i) for p, i in zip(want_pos, want_ids))
    assert out.read_text() == want
