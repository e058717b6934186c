"""GPU matcher: the host-side mirror of the reference's device seam.

``GpuMatcher`` owns one ``pfac_ctx`` (one GPU) and plays the role of
``GPU_Malloc_Memory`` / ``GPU_TraceTable`` / ``GPU_Free_memory``
(main.cc:35-37, master_kernel.cu:188-524) through the C-ABI of
``libpfac_hip.so``.  Nothing here computes matches on the host: every scan is a
launch of the HIP kernel, and a missing library or GPU raises.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np

from ._ffi import PFAC_E_OVERFLOW, CRecord, PfacError, hip_lib
from .table import RECORD_DTYPE, PfacTable


def device_count() -> int:
    n = C.c_int(0)
    rc = hip_lib().pfac_device_count(C.byref(n))
    if rc:
        raise PfacError(rc, (hip_lib().pfac_last_error(None) or b"").decode())
    return n.value


def _ptr(x) -> int:
    """Device pointer of a torch tensor / int / None."""
    if x is None:
        return 0
    if isinstance(x, int):
        return x
    if hasattr(x, "data_ptr"):
        return int(x.data_ptr())
    raise TypeError(f"cannot take a device pointer from {type(x)!r}")


class GpuMatcher:
    """One GPU, ``n_streams`` pipeline slots (the reference's "streams per GPU", argv[2])."""

    def __init__(self, device: int = 0, n_streams: int = 1):
        self._L = hip_lib()
        self._ctx = C.c_void_p()
        self.device = device
        self.n_streams = n_streams
        self.table: Optional[PfacTable] = None
        self._keep = {}
        self._last_n = {}
        rc = self._L.pfac_ctx_create(int(device), int(n_streams), C.byref(self._ctx))
        if rc:
            self._ctx = C.c_void_p()
            raise PfacError(rc, (self._L.pfac_last_error(None) or b"").decode())

    # -- plumbing ---------------------------------------------------------
    def _check(self, rc: int, allow_overflow: bool = False) -> int:
        if rc and not (allow_overflow and rc == PFAC_E_OVERFLOW):
            raise PfacError(rc, (self._L.pfac_last_error(self._ctx) or b"").decode())
        return rc

    def close(self):
        if getattr(self, "_ctx", None):
            self._L.pfac_ctx_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- table ------------------------------------------------------------
    def load_table(self, table) -> None:
        """Upload a ``PfacTable`` (or its int32 image) -- master_kernel.cu:365-383."""
        if isinstance(table, PfacTable):
            self.table = table
            blob = table.blob()
        else:
            blob = np.ascontiguousarray(table, dtype=np.int32)
            self.table = PfacTable.from_blob(blob)
        self._check(self._L.pfac_table_upload(self._ctx, blob.ctypes.data, blob.size))

    def load_table_device(self, d_blob, n_words: int, stream: int = 0, host_table: Optional[PfacTable] = None) -> None:
        """Install a table image that already sits in this GPU's memory (e.g. after an RCCL broadcast)."""
        self._check(self._L.pfac_table_upload_device(self._ctx, _ptr(d_blob), int(n_words), stream))
        if host_table is not None:
            self.table = host_table

    def info(self) -> dict:
        v, t, g, l = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        self._check(self._L.pfac_scan_info(self._ctx, C.byref(v), C.byref(t), C.byref(g), C.byref(l)))
        nb, cap = C.c_int(), C.c_uint32()
        self._check(self._L.pfac_scan_staging(self._ctx, C.byref(nb), C.byref(cap)))
        return {"variant": "tables_in_lds" if v.value == 0 else "tables_via_l2", "tile_bytes": t.value,
                "grid_blocks": g.value, "lds_bytes": l.value, "staging_buffers": nb.value, "staging_records": cap.value}

    # -- buffers ----------------------------------------------------------
    def reserve(self, slot: int = 0, input_bytes: int = 0, record_capacity: int = 0) -> None:
        self._check(self._L.pfac_slot_reserve(self._ctx, slot, int(input_bytes), int(record_capacity)))

    def input_ptr(self, slot: int = 0) -> int:
        return int(self._L.pfac_slot_input(self._ctx, slot) or 0)

    def records_ptr(self, slot: int = 0) -> int:
        return int(self._L.pfac_slot_records(self._ctx, slot) or 0)

    def stream_handle(self, slot: int = 0) -> int:
        return int(self._L.pfac_slot_stream(self._ctx, slot) or 0)

    def set_stream(self, slot: int, stream_handle: int) -> None:
        self._check(self._L.pfac_slot_set_stream(self._ctx, slot, stream_handle))

    def h2d(self, host: np.ndarray, slot: int = 0, dst_offset: int = 0) -> None:
        host = np.ascontiguousarray(host, dtype=np.uint8)
        self._keep.setdefault(slot, []).append(host)   # the copy is asynchronous: released by sync()/scan_finish()
        self._check(self._L.pfac_slot_h2d(self._ctx, slot, host.ctypes.data, host.size, int(dst_offset)))

    def sync(self, slot: int = 0) -> None:
        self._check(self._L.pfac_slot_sync(self._ctx, slot))
        self._keep.pop(slot, None)

    # -- the scan ---------------------------------------------------------
    def scan_async(self, n_owned: int, n_avail: Optional[int] = None, d_input=None, d_records=None,
                   capacity: int = 0, slot: int = 0) -> None:
        """Launch the kernel (master_kernel.cu:406).  ``d_input`` / ``d_records`` None = the slot's buffers."""
        n_avail = n_owned if n_avail is None else n_avail
        self._check(self._L.pfac_scan_async(self._ctx, slot, _ptr(d_input), int(n_owned), int(n_avail),
                                            _ptr(d_records), int(capacity)))

    def scan_finish(self, slot: int = 0, allow_overflow: bool = False) -> Tuple[int, bool]:
        n = C.c_uint64(0)
        rc = self._check(self._L.pfac_scan_finish(self._ctx, slot, C.byref(n)), allow_overflow=allow_overflow)
        # the scan's end event is ordered after the slot's H2D copies on its stream: their host arrays can go
        # (a streaming caller that never calls sync() would otherwise keep every chunk it ever uploaded alive)
        self._keep.pop(slot, None)
        self._last_n[slot] = n.value
        return n.value, rc == PFAC_E_OVERFLOW

    def last_count(self, slot: int = 0) -> int:
        """Match count of the slot's last finished scan."""
        return self._last_n[slot]

    def elapsed_ms(self, slot: int = 0) -> float:
        ms = C.c_float(0)
        self._check(self._L.pfac_scan_elapsed_ms(self._ctx, slot, C.byref(ms)))
        return ms.value

    def records_to_host(self, n: int, slot: int = 0, d_records=None, first: int = 0) -> np.ndarray:
        out = np.empty(int(n), dtype=RECORD_DTYPE)
        if n:
            self._check(self._L.pfac_records_d2h(self._ctx, slot, _ptr(d_records), out.ctypes.data, int(first), int(n)))
            self.sync(slot)
        return out

    def scan_format(self, slot: int = 0) -> Tuple[int, int, int]:
        """(record_bytes, n_tiles, used) of the slot's last scan: record_bytes = 2 or 4 (compact words in the record
        heap) or 8 (pfac_record), n_tiles = entries of the tile index, used = heap records in use (gaps included)."""
        rb, nt, used = C.c_int(0), C.c_uint64(0), C.c_uint64(0)
        self._check(self._L.pfac_scan_format(self._ctx, slot, C.byref(rb), C.byref(nt), C.byref(used)))
        return rb.value, nt.value, used.value

    def capacity_hint(self, slot: int = 0) -> int:
        """A record capacity the slot's last scan fits (after an overflow)."""
        c = C.c_uint64(0)
        self._check(self._L.pfac_scan_capacity_hint(self._ctx, slot, C.byref(c)))
        return c.value

    def expand_records(self, n: int, d_out, slot: int = 0, d_records=None, first: int = 0) -> None:
        """Records [first, first+n) of the slot's last scan as 8-byte ``pfac_record`` into the DEVICE buffer ``d_out``
        (asynchronous on the slot's stream) -- what the RCCL record gather sends."""
        self._check(self._L.pfac_records_expand(self._ctx, slot, _ptr(d_records), int(first), int(n), _ptr(d_out)))

    def packed_to_host(self, slot: int = 0, d_records=None) -> Tuple[np.ndarray, np.ndarray]:
        """The compact device form itself: (uint16 or uint32 heap words [used], uint64 tile index [n_tiles])."""
        rb, nt, used = self.scan_format(slot)
        words = np.empty(int(used), dtype=np.uint16 if rb == 2 else np.uint32)
        tix = np.empty(max(nt, 1), dtype=np.uint64)
        self._check(self._L.pfac_records_d2h_packed(self._ctx, slot, _ptr(d_records), words.ctypes.data, int(used), tix.ctypes.data))
        self.sync(slot)
        return words, tix[:nt]

    def packed_to_host_into(self, host_words, host_tile_index, slot: int = 0, d_records=None) -> Tuple[int, int, int]:
        """``packed_to_host`` into buffers the caller owns (e.g. pinned torch tensors: uint8[used * record_bytes],
        int64[n_tiles]); synchronous.  Returns (record_bytes, n_tiles, used)."""
        rb, nt, used = self.scan_format(slot)
        self._check(self._L.pfac_records_d2h_packed(self._ctx, slot, _ptr(d_records), _ptr(host_words), int(used),
                                                    _ptr(host_tile_index)))
        self.sync(slot)
        return rb, nt, used

    def packed_to_device(self, d_words_out, d_tile_index_out, slot: int = 0, d_records=None) -> Tuple[int, int, int]:
        """The compact form into DEVICE buffers the caller owns (torch tensors): heap words [0, used) -> ``d_words_out``
        (None: leave them where the scan wrote them) and the tile index -> ``d_tile_index_out`` (int64[n_tiles]).
        Asynchronous on the slot's stream.  Returns (record_bytes, n_tiles, used) -- what ``dist.gather_packed`` sends."""
        rb, nt, used = self.scan_format(slot)
        self._check(self._L.pfac_records_packed_device(self._ctx, slot, _ptr(d_records), _ptr(d_words_out), int(used),
                                                       _ptr(d_tile_index_out)))
        return rb, nt, used

    def emit_text_device(self, base: int = 0, slot: int = 0, d_records=None) -> int:
        """Format the slot's last finished scan into ``GPU_match_result.txt`` lines ON THE GPU (main.cc:335-350); returns
        the byte count.  ``text_to_host`` fetches them."""
        n = C.c_uint64(0)
        self._check(self._L.pfac_emit_text_device(self._ctx, slot, _ptr(d_records), int(base), C.byref(n)))
        return n.value

    def text_to_host(self, n_bytes: int, slot: int = 0, first: int = 0) -> bytes:
        out = np.empty(int(n_bytes), dtype=np.uint8)
        if n_bytes:
            self._check(self._L.pfac_text_d2h(self._ctx, slot, out.ctypes.data, int(first), int(n_bytes)))
            self.sync(slot)
        return out.tobytes()

    def checksum(self, n: int, base: int = 0, slot: int = 0, d_records=None) -> int:
        s = C.c_uint64(0)
        self._check(self._L.pfac_records_checksum(self._ctx, slot, _ptr(d_records), int(n), int(base), C.byref(s)))
        return s.value

    def scan_resident(self, n_owned: int, n_avail: Optional[int] = None, d_input=None, slot: int = 0) -> int:
        """Scan input already in HBM into the slot's record buffer, growing it on overflow.  Returns #matches."""
        n_avail = n_owned if n_avail is None else n_avail
        self.scan_async(n_owned, n_avail, d_input=d_input, slot=slot)
        n, over = self.scan_finish(slot, allow_overflow=True)
        cap = 0
        for _ in range(4):
            if not over:
                break
            cap = max(self.capacity_hint(slot), 2 * cap)
            self.reserve(slot, 0, cap)
            self.scan_async(n_owned, n_avail, d_input=d_input, slot=slot)
            n, over = self.scan_finish(slot, allow_overflow=True)
        if over:
            raise PfacError(PFAC_E_OVERFLOW, "record heap still too small after four attempts")
        return n

    def scan_bytes(self, data, n_owned: Optional[int] = None, slot: int = 0) -> np.ndarray:
        """H2D + scan + D2H of one host buffer.  ``n_owned`` < len(data) leaves the rest as read-only halo."""
        buf = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data
        n_avail = int(buf.size)
        n_owned = n_avail if n_owned is None else int(n_owned)
        self.reserve(slot, max(n_avail, 1), max(n_avail // 8, 4096))
        if n_avail:
            self.h2d(buf, slot)
        n = self.scan_resident(n_owned, n_avail, slot=slot)
        return self.records_to_host(n, slot)

    def scan_partitioned(self, tables, data, slot: int = 0) -> np.ndarray:
        """Pattern-partition mode on ONE GPU: the input is copied once, every partition's table (``tables[k]`` =
        ``PfacTable.from_file_part(..., k, len(tables))``) scans it in turn, and the per-partition record lists are
        merged as main.cc:304-324 does.  Returns records ordered by (position, pattern length) whose ``state``
        field holds the PATTERN ID (emit with ``idmap=None``)."""
        from .table import merge_partitions
        buf = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data
        n = int(buf.size)
        self.reserve(slot, max(n, 1), max(n // 8, 4096))
        if n:
            self.h2d(buf, slot)
        lists = []
        for t in tables:
            self.load_table(t)
            cnt = self.scan_resident(n, n, slot=slot)
            lists.append(self.records_to_host(cnt, slot))
        return merge_partitions(lists, [t.idmap for t in tables])

    # -- synthetic inputs (device resident) --------------------------------
    def fill_tiled(self, d_dst, n: int, pattern: bytes, phase: int = 0, slot: int = 0) -> None:
        pat = np.frombuffer(pattern, dtype=np.uint8)
        self._check(self._L.pfac_fill_tiled(self._ctx, slot, _ptr(d_dst), int(n), pat.ctypes.data, pat.size, int(phase)))

    def fill_random(self, d_dst, n: int, seed: int, slot: int = 0) -> None:
        self._check(self._L.pfac_fill_random(self._ctx, slot, _ptr(d_dst), int(n), int(seed) & (2**64 - 1)))


def trace_table_compat(input_bytes: np.ndarray, table: PfacTable, device: int = 0) -> np.ndarray:
    """Call the reference-shaped one-shot seam (``pfac_trace_table_compat``) and return its dense
    ``input_size x max_pat_len`` result array (0xFFFFFFFF = empty), as ``GPU_TraceTable`` fills it."""
    from ._ffi import CThreadData
    L = hip_lib()
    inp = np.ascontiguousarray(input_bytes, dtype=np.uint8)
    n = int(inp.size)
    dense = np.empty((max(n, 1), table.max_pat_len), dtype=np.uint32)
    s0 = np.ascontiguousarray(table.s0); r = np.ascontiguousarray(table.r)
    HT = np.ascontiguousarray(table.HT); val = np.ascontiguousarray(table.val)
    d = CThreadData(inp.ctypes.data, n, table.state_num, table.num_final, dense.ctypes.data, table.ht_size,
                    table.width, s0.ctypes.data, table.max_pat_len, r.ctypes.data, HT.ctypes.data, val.ctypes.data)
    rc = L.pfac_trace_table_compat(C.byref(d), int(device))
    if rc:
        raise PfacError(rc, (L.pfac_last_error(None) or b"").decode())
    return dense[:n]


def splitmix64_bytes(n: int, seed: int) -> np.ndarray:
    """CPU twin of ``pfac_fill_random`` (byte i = byte i&7 of splitmix64(seed + i>>3)); test helper."""
    words = (n + 7) // 8
    x = (np.arange(words, dtype=np.uint64) + np.uint64(seed & (2**64 - 1)))
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        x = x ^ (x >> np.uint64(31))
    return x.view(np.uint8)[:n].copy()


def tiled_bytes(n: int, pattern: bytes, phase: int = 0) -> np.ndarray:
    """CPU twin of ``pfac_fill_tiled``."""
    pat = np.frombuffer(pattern, dtype=np.uint8)
    idx = (np.arange(n, dtype=np.int64) + phase) % pat.size
    return pat[idx]
