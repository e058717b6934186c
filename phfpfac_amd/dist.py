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
  stream is wanted -- in the COMPACT form the scan wrote (``gather_packed``: 2 or 4 bytes per match + 8 bytes per
  4 KiB tile; rank order == position order and each rank's tile index is ordered, so the parts print or expand
  in sequence), or as expanded 8-byte records (``expanded_records`` + ``gather_records``).

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


def _wire(device: torch.device) -> torch.device:
    """Where tensors live while they travel: the device itself over RCCL ("nccl"), host memory over gloo (whose
    point-to-point operations take CPU tensors only) -- results are moved back to ``device`` by the callers."""
    return torch.device("cpu") if dist.get_backend() == "gloo" else device


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
    wire = _wire(device)
    n = torch.zeros(1, dtype=torch.int64, device=wire)
    blob_host = None
    if rank == src:
        if table is None:
            raise ValueError("the source rank must pass a table")
        blob_host = table.blob()
        n[0] = blob_host.size
    dist.broadcast(n, src)
    words = int(n.item())
    if rank == src:
        blob = torch.from_numpy(blob_host).to(wire)
    else:
        blob = torch.empty(words, dtype=torch.int32, device=wire)
    dist.broadcast(blob, src)
    if rank != src:
        table = PfacTable.from_blob(blob.cpu().numpy())
    return blob.to(device), table


def gather_counts(n_local: int, device: torch.device) -> List[int]:
    """All-gather of one int64 per rank: every rank learns every shard's match count."""
    world = dist.get_world_size()
    wire = _wire(device)
    mine = torch.tensor([int(n_local)], dtype=torch.int64, device=wire)
    parts = [torch.empty(1, dtype=torch.int64, device=wire) for _ in range(world)]
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
    """Ordered gather of EXPANDED records to rank ``dst`` (8 bytes per match; prefer ``gather_packed``, which moves the
    compact form -- a quarter of the bytes for a 16-bit automaton).

    ``records`` must hold this rank's SORTED 8-byte ``pfac_record``s viewed as int64, at least ``n_local`` of them:
    the output of ``GpuMatcher.expand_records`` (``pfac_records_expand``) -- NOT the slot's record buffer, which is the
    unordered heap of 2- or 4-byte compact words (``pfac.h``); ``expanded_records`` below does the expansion.
    Returns, on ``dst``, one int64 tensor of ``sum(counts)`` records in rank (== position) order; ``None`` elsewhere.
    Positions stay shard-relative: add ``shard_range(...)[0]`` of the owning rank (``split_gathered``).
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


def expanded_records(matcher, n_local: int, device: torch.device, slot: int = 0, d_records=None) -> torch.Tensor:
    """This rank's records of the slot's last scan as ``gather_records`` wants them: sorted 8-byte ``pfac_record``s in
    a fresh int64 tensor on ``device`` (``pfac_records_expand``; the slot's stream is synchronised)."""
    wide = torch.empty(max(int(n_local), 1), dtype=torch.int64, device=device)
    if n_local:
        matcher.expand_records(int(n_local), wide, slot=slot, d_records=d_records)
    matcher.sync(slot)
    return wide


# ---------------------------------------------------------------------------
# the compact gather: what the scan wrote is what travels

def gather_packed_tensors(words: torch.Tensor, tile_index: torch.Tensor, rec_bytes: int, n_matches: int, dst: int = 0):
    """Ordered gather of the COMPACT record form to rank ``dst``: every rank passes its record heap as a uint8 tensor
    (``used * rec_bytes`` bytes: 2- or 4-byte words ``pos_in_tile:12 | final state``) and its tile index (int64, one
    ``first | count << 40`` entry per 4 KiB tile -- the index is what is ordered, ``pfac.h``).  One all-gather of four
    int64 per rank (bytes, tiles, record width, matches), then one send per tensor per rank to ``dst``.
    Returns on ``dst`` a list with one dict per rank, in rank (== position) order:
    ``{"words": uint8 tensor, "tix": int64 tensor, "rec_bytes": int, "n_matches": int}``; ``None`` elsewhere.
    ``pfac_emit_packed`` prints a rank's part as it is (``emit_gathered``); ``packed_to_records`` expands it.
    Works on CPU tensors with gloo as well (tests/test_dist_cpu.py)."""
    rank, world = dist.get_rank(), dist.get_world_size()
    device = _wire(words.device)
    words, tile_index = words.to(device), tile_index.to(device)
    mine = torch.tensor([int(words.numel()), int(tile_index.numel()), int(rec_bytes), int(n_matches)], dtype=torch.int64, device=device)
    meta = torch.empty(world * 4, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(meta, mine)
    meta = meta.view(world, 4).cpu().tolist()
    if rank != dst:
        if words.numel():
            dist.send(words.contiguous(), dst=dst)
        if tile_index.numel():
            dist.send(tile_index.contiguous(), dst=dst)
        return None
    parts, reqs = [], []
    for r in range(world):
        nb, nt, rb, nm = (int(v) for v in meta[r])
        if r == dst:
            w, t = words, tile_index
        else:
            w = torch.empty(nb, dtype=torch.uint8, device=device)
            t = torch.empty(nt, dtype=torch.int64, device=device)
            if nb:
                reqs.append(dist.irecv(w, src=r))
            if nt:
                reqs.append(dist.irecv(t, src=r))
        parts.append({"words": w, "tix": t, "rec_bytes": rb, "n_matches": nm})
    for q in reqs:
        q.wait()
    return parts


def gather_packed(matcher, device: torch.device, slot: int = 0, d_records=None, dst: int = 0):
    """``gather_packed_tensors`` for the slot's last finished scan: the heap words and the tile index are copied into
    torch-owned device buffers (``pfac_records_packed_device``), the slot's stream is synchronised, then they travel."""
    rb, n_tiles, used = matcher.scan_format(slot)
    if rb == 8:
        raise ValueError("automata beyond 2^20 final states write 8-byte records: use expanded_records + gather_records")
    words = torch.empty(int(used) * rb, dtype=torch.uint8, device=device)
    tix = torch.empty(int(n_tiles), dtype=torch.int64, device=device)
    matcher.packed_to_device(words, tix, slot=slot, d_records=d_records)
    matcher.sync(slot)
    n_matches = int(matcher.last_count(slot))
    return gather_packed_tensors(words, tix, rb, n_matches, dst=dst)


def packed_to_records(words: np.ndarray, tile_index: np.ndarray, rec_bytes: int, base: int = 0) -> np.ndarray:
    """Host-side expansion of one rank's compact part: structured array (pos u64, state u32), sorted by (position,
    pattern length), positions = ``base`` + tile * 4096 + (word & 4095).  (numpy; for tests and small consumers --
    ``emit_gathered`` prints without expanding.)"""
    w = np.ascontiguousarray(words).view(np.uint16 if rec_bytes == 2 else np.uint32)
    tix = np.ascontiguousarray(tile_index).view(np.uint64)
    cnt = (tix >> np.uint64(40)).astype(np.int64)
    first = (tix & np.uint64((1 << 40) - 1)).astype(np.int64)
    n = int(cnt.sum())
    out = np.empty(n, dtype=np.dtype([("pos", np.uint64), ("state", np.uint32)]))
    if n == 0:
        return out
    tile_of = np.repeat(np.arange(tix.size, dtype=np.int64), cnt)
    start = np.cumsum(cnt) - cnt
    idx = first[tile_of] + (np.arange(n, dtype=np.int64) - start[tile_of])
    ww = w[idx].astype(np.uint32)
    out["pos"] = (np.uint64(base) + tile_of.astype(np.uint64) * np.uint64(4096) + (ww & np.uint32(4095)).astype(np.uint64))
    out["state"] = ww >> np.uint32(12)
    return out


def emit_gathered(path, parts, idmap, n_total: int, threads: int = 1) -> int:
    """``GPU_match_result.txt`` (main.cc:335-350) from a ``gather_packed`` result, rank by rank (== position order),
    printed straight from the compact form by ``pfac_emit_packed`` with each rank's shard offset as base."""
    import ctypes as C
    import os
    from ._ffi import PfacError, host_lib
    L = host_lib()
    libc = C.CDLL(None)
    libc.fopen.restype = C.c_void_p
    libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
    libc.fclose.argtypes = [C.c_void_p]
    idm = np.ascontiguousarray(idmap, dtype=np.int32)
    f = libc.fopen(os.fsencode(path), b"wb")
    if not f:
        raise PfacError(-2, f"cannot open {path}")
    total = 0
    try:
        for r, p in enumerate(parts):
            lo, _ = shard_range(n_total, r, len(parts))
            w = np.ascontiguousarray(p["words"].cpu().numpy())
            t = np.ascontiguousarray(p["tix"].cpu().numpy()).view(np.uint64)
            n = L.pfac_emit_packed(f, w.ctypes.data, w.size // p["rec_bytes"], p["rec_bytes"], t.ctypes.data, t.size, int(lo),
                                   idm.ctypes.data, int(threads))
            if n < 0:
                raise PfacError(int(n), "pfac_emit_packed")
            total += int(n)
    finally:
        libc.fclose(f)
    return total


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
