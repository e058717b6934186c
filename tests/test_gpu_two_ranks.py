"""Two (and four) RANKS on hardware (run with -m gpu): the N > 1 path of phfpfac_amd/dist.py with real scans.

The builder's box has ONE MI355X, and RCCL refuses two ranks on one device, so the two processes of this test share
cuda:0 and talk over `gloo` (which carries device tensors): everything but the transport is what `bench.py --gpus 2`
and a two-GPU consumer run -- rank 0 builds the table and broadcasts its image into device memory, every rank installs
it with pfac_table_upload_device, generates ITS shard of the global byte stream (owned range + max_pat_len-1 bytes of
halo), scans it with the HIP kernel, the ranks exchange counts and rank 0 gathers the COMPACT records (heap words + tile
index as the kernel wrote them).  Rank 0 then expands them with global positions and prints them: both must equal the
oracle's result for the UNSHARDED stream, bit for bit -- including the matches that straddle the cut between the shards.
"""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
DATA = os.path.join(HERE, "golden", "data")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank(rank, world, port, pat_path, n_total, kind, out_dir):
    sys.path.insert(0, REPO); sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from phfpfac_amd import GpuMatcher, PfacTable
    from phfpfac_amd import dist as pdist
    torch.cuda.set_device(0)                                   # both ranks on the one device (see the module docstring)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        table = PfacTable.from_file(pat_path, 256) if rank == 0 else None
        blob, table = pdist.broadcast_table(table, dev, 0)
        para = open(os.path.join(DATA, "paragraph402"), "rb").read()
        lo, hi, end = pdist.shard_read_range(n_total, rank, world, table.halo)
        n_owned, n_avail = hi - lo, end - lo
        with GpuMatcher(0, 1) as g:
            g.load_table_device(blob, blob.numel(), 0, host_table=table)
            buf = torch.empty(n_avail + 4096, dtype=torch.uint8, device=dev)
            if kind == "text":
                g.fill_tiled(buf, n_avail, para, phase=lo % len(para))
            else:
                g.fill_random(buf, (n_avail + 7) // 8 * 8, 0x5048465046414331 + lo // 8)
            g.reserve(0, 0, max(n_owned // 2, 1 << 16))
            n = g.scan_resident(n_owned, n_avail, d_input=buf)
            counts = pdist.gather_counts(n, dev)
            assert counts[rank] == n
            parts = pdist.gather_packed(g, dev, slot=0, dst=0)
        if rank == 0:
            assert [p["n_matches"] for p in parts] == counts
            rec = np.concatenate([pdist.packed_to_records(p["words"].cpu().numpy(), p["tix"].cpu().numpy(), p["rec_bytes"],
                                                          base=pdist.shard_range(n_total, r, world)[0]) for r, p in enumerate(parts)])
            np.save(os.path.join(out_dir, "rec.npy"), rec)
            np.save(os.path.join(out_dir, "idmap.npy"), table.idmap)
            pdist.emit_gathered(os.path.join(out_dir, "gathered.txt"), parts, table.idmap, n_total, threads=4)
        else:
            assert parts is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,pattern,kind,n_total", [(2, "xaa", "text", (24 << 20) + 12345), (2, "experimentpattern", "text", (64 << 20) + 7),
                                                         (2, "bytefile_10000byte", "rand", (16 << 20) + 401), (4, "xaa", "text", (24 << 20) + 999)])
def test_two_ranks_sharded_scan_on_the_gpu(world, pattern, kind, n_total, tmp_path):
    import torch.multiprocessing as mp
    from orc import Oracle
    from phfpfac_amd.matcher import splitmix64_bytes, tiled_bytes
    pat_path = os.path.join(DATA, pattern)
    mp.spawn(_rank, args=(world, _free_port(), pat_path, n_total, kind, str(tmp_path)), nprocs=world, join=True)
    rec = np.load(tmp_path / "rec.npy")
    idmap = np.load(tmp_path / "idmap.npy")
    para = open(os.path.join(DATA, "paragraph402"), "rb").read()
    whole = tiled_bytes(n_total, para) if kind == "text" else splitmix64_bytes(n_total, 0x5048465046414331)
    o = Oracle(pat_path, 1, 1)
    pos, ids = o.scan_spec(whole)
    assert rec.size == pos.size
    np.testing.assert_array_equal(rec["pos"].astype(np.int64), pos)
    np.testing.assert_array_equal(idmap[rec["state"]], ids)
    exp = tmp_path / "expected.txt"
    o.emit(whole, str(exp), spec=True)
    o.close()
    assert (tmp_path / "gathered.txt").read_bytes() == exp.read_bytes()
    # the cut between the shards lies inside the stream: matches start on both sides of it
    cut = -(-n_total // world)
    cut = (cut + 15) // 16 * 16
    if pos.size > 100:
        assert (pos < cut).any() and (pos >= cut).any()
