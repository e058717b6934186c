"""The N>1 path on CPU: world_size-2 `gloo` processes exercise the sharding plan, the table broadcast and the
ordered record gather of phfpfac_amd/dist.py.  There is no GPU here and the product has no CPU scan, so each
rank's shard records are produced by the CHECKER (the oracle's spec walk over that shard's owned range + halo);
what is under test is that sharding + broadcast + gather reassemble exactly the unsharded result."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
DATA = os.path.join(HERE, "golden", "data")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def pack_records(rec, n_owned, rec_bytes, seed):
    """CPU stand-in for what the scan kernel leaves in HBM (include/pfac.h): a record HEAP of compact words
    (pos & 4095 | state << 12), tiles placed in arbitrary order with gaps between them, plus the ordered tile index
    (first | count << 40 per 4 KiB tile).  Returns (uint8 heap bytes, int64 tile index)."""
    rng = np.random.default_rng(seed)
    n_tiles = (n_owned + 4095) // 4096
    tile = (rec["pos"] >> 12).astype(np.int64)
    counts = np.bincount(tile, minlength=n_tiles).astype(np.int64)
    first = np.zeros(n_tiles, dtype=np.int64)
    at = 0
    for t in rng.permutation(n_tiles):
        at += int(rng.integers(0, 7))
        first[t] = at
        at += int(counts[t])
    dt = np.uint16 if rec_bytes == 2 else np.uint32
    words = rng.integers(0, 1 << (8 * rec_bytes), at + 3, dtype=np.uint64).astype(dt)       # garbage in the gaps
    start = np.cumsum(counts) - counts
    idx = first[tile] + (np.arange(rec.size) - start[tile])
    words[idx] = ((rec["pos"] & 4095) | (rec["state"] << 12)).astype(dt)
    tix = first.astype(np.uint64) | (counts.astype(np.uint64) << np.uint64(40))
    return words.view(np.uint8).copy(), tix.view(np.int64).copy()


def _worker(rank, world, port, pat_path, n_total, out_dir):
    sys.path.insert(0, REPO); sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from orc import Oracle
    from phfpfac_amd import PfacTable, RECORD_DTYPE
    from phfpfac_amd import dist as pdist
    from phfpfac_amd.matcher import tiled_bytes
    dev = torch.device("cpu")
    # rank 0 builds the table on the host and broadcasts the image
    table = PfacTable.from_file(pat_path, 256) if rank == 0 else None
    blob, table = pdist.broadcast_table(table, dev, 0)
    ref = PfacTable.from_file(pat_path, 256)
    assert np.array_equal(blob.numpy(), ref.blob())
    assert table.state_num == ref.state_num and table.max_pat_len == ref.max_pat_len
    # this rank's shard of the global stream (generated locally, as bench.py does on the GPU)
    para = open(os.path.join(DATA, "paragraph402"), "rb").read()
    lo, hi, end = pdist.shard_read_range(n_total, rank, world, table.halo)
    assert lo % 16 == 0 and (hi % 16 == 0 or hi == n_total)
    shard = tiled_bytes(end - lo, para, phase=lo % len(para))
    o = Oracle(pat_path, 1, 1)
    pos, ids = o.scan_spec(shard)                 # walks may use the halo ...
    keep = pos < (hi - lo)                        # ... but only owned offsets report
    inv = {int(v): k for k, v in enumerate(o.idmap())}
    rec = np.empty(int(keep.sum()), dtype=RECORD_DTYPE)
    rec["pos"] = pos[keep]
    rec["state"] = [inv[int(i)] for i in ids[keep]]
    o.close()
    counts = pdist.gather_counts(rec.size, dev)
    assert counts[rank] == rec.size
    t = torch.from_numpy(rec.view(np.int64).copy()) if rec.size else torch.empty(0, dtype=torch.int64)
    gathered = pdist.gather_records(t, rec.size, counts, dst=0)
    if rank == 0:
        allrec = pdist.split_gathered(gathered, counts, n_total, world)
        np.save(os.path.join(out_dir, "gathered.npy"), allrec)
        np.save(os.path.join(out_dir, "counts.npy"), np.array(counts))
    else:
        assert gathered is None
    # the COMPACT gather (what bench.py and a multi-GPU consumer use): heap words + tile index travel as they are
    rec_bytes = 2 if table.num_final <= 16 else 4
    wbytes, tix = pack_records(rec, hi - lo, rec_bytes, seed=100 + rank)
    parts = pdist.gather_packed_tensors(torch.from_numpy(wbytes), torch.from_numpy(tix), rec_bytes, rec.size, dst=0)
    if rank == 0:
        assert [p["n_matches"] for p in parts] == counts and all(p["rec_bytes"] == rec_bytes for p in parts)
        moved = sum(p["words"].numel() + 8 * p["tix"].numel() for p in parts)
        assert moved < 8 * sum(counts) + 8 * sum(p["tix"].numel() for p in parts) + 64 * world       # fewer bytes than the 8-byte form
        got = np.concatenate([pdist.packed_to_records(p["words"].numpy(), p["tix"].numpy(), p["rec_bytes"],
                                                      base=pdist.shard_range(n_total, r, world)[0]) for r, p in enumerate(parts)])
        np.save(os.path.join(out_dir, "packed.npy"), got)
        nbytes = pdist.emit_gathered(os.path.join(out_dir, "packed.txt"), parts, table.idmap, n_total, threads=2)
        assert nbytes == os.path.getsize(os.path.join(out_dir, "packed.txt"))
    else:
        assert parts is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("pattern,n_total", [("xaa", 100003), ("experimentpattern", 65536 + 402)])
def test_two_rank_sharded_scan_reassembles(pattern, n_total, tmp_path):
    from orc import Oracle
    from phfpfac_amd import PfacTable
    from phfpfac_amd.matcher import tiled_bytes
    pat_path = os.path.join(DATA, pattern)
    port = _free_port()
    mp.spawn(_worker, args=(2, port, pat_path, n_total, str(tmp_path)), nprocs=2, join=True)
    got = np.load(tmp_path / "gathered.npy")
    para = open(os.path.join(DATA, "paragraph402"), "rb").read()
    whole = tiled_bytes(n_total, para)
    o = Oracle(pat_path, 1, 1)
    pos, ids = o.scan_spec(whole)
    table = PfacTable.from_file(pat_path, 256)
    assert got.size == pos.size
    np.testing.assert_array_equal(got["pos"].astype(np.int64), pos)
    np.testing.assert_array_equal(table.idmap[got["state"]], ids)
    assert (np.diff(got["pos"].astype(np.int64)) >= 0).all()
    # the compact gather reassembles the same sequence, and prints the reference's text from it
    packed = np.load(tmp_path / "packed.npy")
    np.testing.assert_array_equal(packed["pos"].astype(np.int64), pos)
    np.testing.assert_array_equal(table.idmap[packed["state"]], ids)
    exp = tmp_path / "expected.txt"
    o.emit(whole, str(exp), spec=True)
    assert (tmp_path / "packed.txt").read_bytes() == exp.read_bytes()
    o.close()


def test_shard_plan_properties():
    from phfpfac_amd.dist import shard_range, shard_read_range
    for n in (0, 1, 15, 16, 17, 1000, 1 << 20, (1 << 35) + 5):
        for world in (1, 2, 3, 4, 8):
            prev = 0
            for r in range(world):
                lo, hi = shard_range(n, r, world)
                assert lo == prev and lo <= hi <= n and (lo % 16 == 0 or lo == n)
                prev = hi
                lo2, hi2, end = shard_read_range(n, r, world, 227)
                assert (lo2, hi2) == (lo, hi) and hi <= end <= min(n, hi + 227)
            assert prev == n


def test_bench_spawns_one_process_per_gpu(tmp_path):
    """`python bench.py --gpus 2` from a bare command line: the parent starts two fresh children with the
    torch.distributed environment of one node (it never touches a GPU itself), passes rank 0's stdout through and
    returns the worst exit code.  PFAC_BENCH_SPAWN_TEST makes the children report their environment instead of
    benchmarking (no GPU here)."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["PFAC_BENCH_SPAWN_TEST"] = str(tmp_path)
    bench = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "3"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["rank"] == "0"         # only rank 0 prints to the parent's stdout
    got = [json.load(open(tmp_path / f"rank{k}.json")) for k in range(2)]
    assert [g["rank"] for g in got] == ["0", "1"] and [g["local_rank"] for g in got] == ["0", "1"]
    assert all(g["world"] == "2" and g["master"].startswith("127.0.0.1:") for g in got)
    assert got[0]["master"] == got[1]["master"] and got[0]["argv"] == ["--gpus", "2", "--steps", "3"]
