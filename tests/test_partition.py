"""Pattern-partition mode (reference: create_table_reorder.c:217-247 + main.cc:304-324) on the
host: partition tables must equal the oracle's per-chunk tries cell for cell, and merging the per-partition match
lists must give the reference's merged order.  The product has no CPU scan, so the per-partition match lists of
these CPU tests come from a small walk through the library's host-side lookup helper (``pfac_table_lookup``)."""
import os
import socket
import sys

import numpy as np
import pytest

from orc import Oracle
from phfpfac_amd import RECORD_DTYPE, PfacError, PfacTable, emit_records, merge_partitions
from test_table import lookup_all

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)


def walk_records(t, data, n=None):
    """(pos, final state) records of table ``t`` over ``data`` -- the PFAC rule through the host lookup helper."""
    tr = lookup_all(t)
    n = len(data) if n is None else n
    root = t.num_final + 1
    out = []
    for i in range(n):
        s, p = root, i
        while p < n:
            s = tr[s, data[p]]
            if s < 0:
                break
            if s < t.num_final:
                out.append((i, s))
            p += 1
    rec = np.empty(len(out), dtype=RECORD_DTYPE)
    if out:
        a = np.array(out, dtype=np.int64)
        rec["pos"], rec["state"] = a[:, 0], a[:, 1]
    return rec


@pytest.mark.parametrize("name,streams", [("xaa", 1), ("xaa+xab+xac+xad", 2), ("experimentpattern", 1), ("bytefile/10000byte", 3)])
def test_partition_tables_equal_reference_chunks(name, streams, resolve):
    """P = 4 * streamnum partitions: same patterns, same state numbering, same id maps as the reference's chunks."""
    o = Oracle(resolve(name), streams, 4)
    P = o.P
    assert P == 4 * streams
    full = PfacTable.from_file(resolve(name), 256)
    for c in range(P):
        t = PfacTable.from_file_part(resolve(name), 256, c, P)
        dense = o.trie(c)
        assert t.state_num == dense.shape[0] and t.num_final == o.stats(c)["final"]
        assert (t.idmap == o.idmap(c)).all()
        assert (lookup_all(t) == dense).all()
        assert t.max_pat_len == full.max_pat_len            # the global maximum, ctr.c:238,246
    o.close()


def test_merge_equals_reference_merge(resolve):
    """Per-partition scans merged by pfac_merge_partitions == the reference's kernel-per-chunk + host merge
    (oracle, P = 4) == the single-automaton result; and the emitted text is the same file."""
    name = "xaa+xab+xac+xad"
    data = open(resolve("1M"), "rb").read()[:6000]
    buf = np.frombuffer(data, dtype=np.uint8)
    tabs = [PfacTable.from_file_part(resolve(name), 256, c, 4) for c in range(4)]
    lists = [walk_records(t, buf) for t in tabs]
    merged = merge_partitions(lists, [t.idmap for t in tabs])
    o = Oracle(resolve(name), 1, 4)
    o.ffdm(256)
    pos, ids = o.scan_reference(buf)
    o.close()
    assert merged.size == pos.size
    np.testing.assert_array_equal(merged["pos"].astype(np.int64), pos)
    np.testing.assert_array_equal(merged["state"].astype(np.int32), ids)
    o1 = Oracle(resolve(name), 1, 1)
    pos1, ids1 = o1.scan_spec(buf)
    o1.close()
    np.testing.assert_array_equal(merged["pos"].astype(np.int64), pos1)
    np.testing.assert_array_equal(merged["state"].astype(np.int32), ids1)
    # ids already applied: same result when the lists are translated first
    pre = []
    for t, r in zip(tabs, lists):
        q = r.copy()
        q["state"] = t.idmap[r["state"]]
        pre.append(q)
    assert (merge_partitions(pre, None) == merged).all()


def test_emit_with_ids(resolve, tmp_path):
    full = PfacTable.from_file(resolve("experimentpattern"), 256)
    data = np.frombuffer(open(resolve("experimentinput"), "rb").read()[:-1], dtype=np.uint8)
    rec = walk_records(full, data)
    emit_records(str(tmp_path / "a.txt"), rec, full.idmap)
    ids = rec.copy()
    ids["state"] = full.idmap[rec["state"]]
    emit_records(str(tmp_path / "b.txt"), ids, None)
    a = (tmp_path / "a.txt").read_bytes()
    assert a == (tmp_path / "b.txt").read_bytes()
    assert a == open(os.path.join(HERE, "golden", "out", "exp_x_expinput_s1_w256.txt"), "rb").read()


def test_duplicates_never_straddle_a_cut():
    """8 lines, 4 partitions of 2: the duplicate `b` would open partition 1 in the reference (and overflow position
    slots, main.cc:308-315); here the cut moves past it and the later line wins, as in the single automaton."""
    pats = b"a\nb\nb\nc\nd\ne\nf\ng\n"
    tabs = [PfacTable.from_bytes(pats, 256, part=c, n_parts=4) for c in range(4)]
    assert [list(t.idmap) for t in tabs] == [[1, 2, 3], [4], [5, 6], [7, 8]]
    data = np.frombuffer(b"abcdefg", dtype=np.uint8)
    merged = merge_partitions([walk_records(t, data) for t in tabs], [t.idmap for t in tabs])
    full = PfacTable.from_bytes(pats, 256)
    one = walk_records(full, data)
    assert list(merged["pos"]) == list(one["pos"]) == [0, 1, 2, 3, 4, 5, 6]
    assert list(merged["state"]) == list(full.idmap[one["state"]]) == [1, 3, 4, 5, 6, 7, 8]
    # prefix chain across partitions: within a position the order is partition order == pattern length order
    pats = b"abcd\nab\nabc\na\nabcde\nzz\n"
    tabs = [PfacTable.from_bytes(pats, 256, part=c, n_parts=3) for c in range(3)]
    data = np.frombuffer(b"xabcdeabcdzz", dtype=np.uint8)
    merged = merge_partitions([walk_records(t, data) for t in tabs], [t.idmap for t in tabs])
    full = PfacTable.from_bytes(pats, 256)
    one = walk_records(full, data)
    assert list(merged["pos"]) == list(one["pos"])
    assert list(merged["state"]) == list(full.idmap[one["state"]])
    assert list(merged["state"][:5]) == [4, 2, 3, 1, 5]


def test_partition_edge_cases():
    pats = b"x\ny\n"
    # more partitions than patterns: k = 0, the last partition takes everything (ctr.c:220-222)
    tabs = [PfacTable.from_bytes(pats, 256, part=c, n_parts=5) for c in range(5)]
    assert [t.n_patterns for t in tabs] == [0, 0, 0, 0, 2]
    assert all((t.s0 == -1).all() for t in tabs[:4])
    data = np.frombuffer(b"xyx", dtype=np.uint8)
    merged = merge_partitions([walk_records(t, data) for t in tabs], [t.idmap for t in tabs])
    assert list(merged["pos"]) == [0, 1, 2] and list(merged["state"]) == [1, 2, 1]
    assert merge_partitions([], None).size == 0
    assert merge_partitions([np.empty(0, dtype=RECORD_DTYPE)] * 3, None).size == 0
    with pytest.raises(PfacError):
        PfacTable.from_bytes(pats, 256, part=2, n_parts=2)
    with pytest.raises(PfacError):
        PfacTable.from_bytes(pats, 256, part=0, n_parts=0)
    # random interleavings: the merge is a stable sort by (pos, partition)
    rng = np.random.default_rng(5)
    lists = []
    for k in range(6):
        n = int(rng.integers(0, 4000))
        r = np.empty(n, dtype=RECORD_DTYPE)
        r["pos"] = np.sort(rng.integers(0, 3000, n))
        r["state"] = k * 100000 + np.arange(n)
        lists.append(r)
    merged = merge_partitions(lists, None)
    cat = np.concatenate(lists)
    part = np.concatenate([np.full(l.size, k) for k, l in enumerate(lists)])
    order = np.lexsort((np.arange(cat.size), part, cat["pos"]))
    assert (merged == cat[order]).all()


# ---- world-size-2 gloo: rank g = pattern partition g, input replicated by one broadcast, merge on rank 0 ----
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, pat_path, data_path, n, out_dir):
    sys.path.insert(0, REPO); sys.path.insert(0, HERE)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from phfpfac_amd import dist as pdist
    dev = torch.device("cpu")
    table = PfacTable.from_file_part(pat_path, 256, rank, world)
    src = torch.from_numpy(np.frombuffer(open(data_path, "rb").read()[:n], dtype=np.uint8).copy()) if rank == 0 else None
    buf = pdist.broadcast_input(src, dev, 0).numpy()
    assert buf.size == n
    rec = walk_records(table, buf)                     # stand-in for this rank's GPU scan of the whole input
    merged = pdist.gather_partition_matches(rec, table, dev, dst=0)
    if rank == 0:
        np.save(os.path.join(out_dir, "merged.npy"), merged)
    else:
        assert merged is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_pattern_partition(resolve, tmp_path):
    import torch.multiprocessing as mp
    pat, n = resolve("xaa"), 5000
    mp.spawn(_worker, args=(2, _free_port(), pat, resolve("1M"), n, str(tmp_path)), nprocs=2, join=True)
    got = np.load(tmp_path / "merged.npy")
    o = Oracle(pat, 1, 1)
    pos, ids = o.scan_spec(np.frombuffer(open(resolve("1M"), "rb").read()[:n], dtype=np.uint8))
    o.close()
    np.testing.assert_array_equal(got["pos"].astype(np.int64), pos)
    np.testing.assert_array_equal(got["state"].astype(np.int32), ids)
