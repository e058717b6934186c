"""Multi-GPU plumbing: one process per GPU, ``torch.distributed`` (backend "nccl" == RCCL over xGMI).

The reference has no inter-GPU communication at all: it partitions the PATTERN set and lets every
stream scan the whole input (create_table_reorder.c:217-247, master_kernel.cu:359).  Here the INPUT
byte stream is sharded instead -- each start offset's walk reads only ``input[i : i+max_pat_len)`` and
read-only tables, so shards are independent units:

* rank g owns start offsets ``[lo_g, hi_g)`` (contiguous, 16-byte aligned cuts) and additionally
  READS ``max_pat_len - 1`` bytes past ``hi_g`` (the halo; none past the global end);
* the packed table image is broadcast ONCE from rank 0 (``broadcast_table``);
* per scan the only exchange is an all-gather of one match count per rank (``gather_counts``), which
  gives every rank its record offset; records are gathered to one rank only when a single ordered
  stream is wanted (``gather_records``) -- rank order == position order, so concatenation is sorted.

No data-path collective touches the input bytes.

Fallback, for automata that outgrow L2/MALL: the reference's own scheme, PATTERN
partitioning -- rank g builds the table of partition g (``PfacTable.from_file_part``), the input is replicated with
one broadcast (``broadcast_input``), every rank scans all of it, and ``gather_partition_matches`` merges the ranks'
match lists by (position, partition) on one rank, which is the reference's host merge (main.cc:304-324).

All functions work with the ``gloo`` backend on CPU
tensors as well (that is how tests/test_dist_cpu.py exercises them with world_size 2).
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist

from .table import RECORD_DTYPE, PfacTable, merge_partitions

ALIGN = 16


def shard_range(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """Owned start-offset range of ``rank``: contiguous, cut at multiples of 16 bytes."""
    per = -(-n_total // world)
    per = (per + ALIGN - 1) // ALIGN * ALIGN
    lo = min(n_total, rank * per)
    hi = min(n_total, lo + per)
    return lo, hi


def shard_read_range(n_total: int, rank: int, world: int, halo: int) -> Tuple[int, int, int]:
    """(lo, hi, end): owned ``[lo, hi)``, readable ``[lo, end)`` with ``end = min(n_total, hi + halo)``."""
    lo, hi = shard_range(n_total, rank, world)
    return lo, hi, min(n_total, hi + halo)


def broadcast_table(table: Optional[PfacTable], device: torch.device, src: int = 0) -> Tuple[torch.Tensor, PfacTable]:
    """Rank ``src`` passes its ``PfacTable``; every rank returns (int32 image on ``device``, host table).

    Two collectives: the image length, then the image itself (one flat int32 buffer: root row +
    r + HT + val + idmap, <= ~17 MiB for the largest pattern sets) -- a single broadcast over xGMI.
    """
    rank = dist.get_rank()
    n = torch.zeros(1, dtype=torch.int64, device=device)
    blob_host = None
    if rank == src:
        if table is None:
            raise ValueError("the source rank must pass a table")
        blob_host = table.blob()
        n[0] = blob_host.size
    dist.broadcast(n, src)
    words = int(n.item())
    if rank == src:
        blob = torch.from_numpy(blob_host).to(device)
    else:
        blob = torch.empty(words, dtype=torch.int32, device=device)
    dist.broadcast(blob, src)
    if rank != src:
        table = PfacTable.from_blob(blob.cpu().numpy())
    return blob, table


def gather_counts(n_local: int, device: torch.device) -> List[int]:
    """All-gather of one int64 per rank: every rank learns every shard's match count."""
    world = dist.get_world_size()
    mine = torch.tensor([int(n_local)], dtype=torch.int64, device=device)
    parts = [torch.empty(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(parts, mine)
    return [int(p.item()) for p in parts]


def gather_counts_async(n_local: int, device: torch.device):
    """Non-blocking form: returns (work, tensor of world int64 counts).  The exchange only places records, so a
    scan loop can overlap it with the next scan and ``work.wait()`` when the offsets are needed."""
    world = dist.get_world_size()
    mine = torch.tensor([int(n_local)], dtype=torch.int64, device=device)
    out = torch.empty(world, dtype=torch.int64, device=device)
    work = dist.all_gather_into_tensor(out, mine, async_op=True)
    return work, out


def gather_records(records: torch.Tensor, n_local: int, counts: List[int], dst: int = 0) -> Optional[torch.Tensor]:
    """Ordered gather of compact records to rank ``dst``.

    ``records`` is this rank's record buffer viewed as int64 (one 8-byte ``pfac_record`` per element,
    at least ``n_local`` long).  Returns, on ``dst``, one int64 tensor of ``sum(counts)`` records in rank
    (== position) order; ``None`` elsewhere.  Positions stay shard-relative: add ``shard_range(...)[0]``
    of the owning rank (``split_gathered``).
    """
    rank, world = dist.get_rank(), dist.get_world_size()
    if rank == dst:
        out = torch.empty(sum(counts), dtype=torch.int64, device=records.device)
        offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        reqs = []
        for r in range(world):
            if counts[r] == 0:
                continue
            piece = out[offs[r]: offs[r + 1]]
            if r == dst:
                piece.copy_(records[:n_local])
            else:
                reqs.append(dist.irecv(piece, src=r))
        for q in reqs:
            q.wait()
        return out
    if n_local:
        dist.send(records[:n_local].contiguous(), dst=dst)
    return None


def split_gathered(gathered: torch.Tensor, counts: List[int], n_total: int, world: int) -> np.ndarray:
    """Host-side view of a gathered record stream with GLOBAL positions: structured array (pos u64, state u32)."""
    raw = gathered.cpu().numpy().view(RECORD_DTYPE)
    out = np.empty(raw.size, dtype=np.dtype([("pos", np.uint64), ("state", np.uint32)]))
    out["state"] = raw["state"]
    k = 0
    for r in range(world):
        lo, _ = shard_range(n_total, r, world)
        out["pos"][k: k + counts[r]] = raw["pos"][k: k + counts[r]].astype(np.uint64) + np.uint64(lo)
        k += counts[r]
    return out


# ---------------------------------------------------------------------------
# pattern-partition mode (the reference's scheme, create_table_reorder.c:217-247 + main.cc:304-324)

def broadcast_input(data: Optional[torch.Tensor], device: torch.device, src: int = 0) -> torch.Tensor:
    """Replicate the input bytes: rank ``src`` passes a uint8 tensor, every rank returns it on ``device``.
    (The reference copies the whole input to every stream, master_kernel.cu:359; here it is ONE broadcast.)"""
    rank = dist.get_rank()
    n = torch.zeros(1, dtype=torch.int64, device=device)
    if rank == src:
        if data is None:
            raise ValueError("the source rank must pass the input")
        n[0] = data.numel()
    dist.broadcast(n, src)
    buf = data.to(device).contiguous() if rank == src else torch.empty(int(n.item()), dtype=torch.uint8, device=device)
    dist.broadcast(buf, src)
    return buf


def gather_partition_matches(records: np.ndarray, table: PfacTable, device: torch.device, dst: int = 0) -> Optional[np.ndarray]:
    """``records``: this rank's (== this pattern partition's) records over the WHOLE input.  Returns on ``dst`` the
    merged match list -- ordered by (position, partition) = (position, pattern length), ``state`` = pattern id --
    and None elsewhere.  One count all-gather + one ordered record gather; the merge runs in the host library."""
    world = dist.get_world_size()
    ids = np.ascontiguousarray(records, dtype=RECORD_DTYPE).copy()
    if ids.size:
        ids["state"] = table.idmap[ids["state"]].astype(np.uint32)   # partitions number their final states separately
    counts = gather_counts(ids.size, device)
    t = torch.from_numpy(ids.view(np.int64)).to(device) if ids.size else torch.empty(0, dtype=torch.int64, device=device)
    gathered = gather_records(t, ids.size, counts, dst=dst)
    if gathered is None:
        return None
    raw = gathered.cpu().numpy().view(RECORD_DTYPE)
    offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    return merge_partitions([raw[offs[r]: offs[r + 1]] for r in range(world)], None)
