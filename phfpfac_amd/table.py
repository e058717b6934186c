"""Host-side table build: pattern file -> PHF-compressed PFAC transition table.

Thin wrapper over ``libpfac_host.so`` (``csrc/pfac_table.c``), i.e. over the
replacement of the reference's ``create_PFAC_table_reorder()`` + ``FFDM()``
(main.cc:108,125).  All arithmetic happens in the C library.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from ._ffi import CTable, PfacError, host_lib

RECORD_DTYPE = np.dtype([("pos", np.uint32), ("state", np.uint32)])


class PfacTable:
    """One automaton over the whole pattern file, in the reference's table terms.

    Attributes mirror ``struct thread_data`` (main.cc:19-32): ``s0`` (root row),
    ``r``, ``HT``, ``val`` (the perfect hash), ``width``, ``ht_size`` (HTSize),
    ``state_num``, ``num_final`` (final_state_num), ``max_pat_len``; ``idmap``
    is ``patternIdMaps`` (final state -> 1-based pattern line number).
    """

    def __init__(self, ptr):
        self._ptr = ptr
        t = ptr.contents
        for name in ("width", "width_bit", "n_patterns", "num_final", "state_num", "max_pat_len", "max_row",
                     "ht_size", "n_keys"):
            setattr(self, name, int(getattr(t, name)))
        self.s0 = np.ctypeslib.as_array(t.s0, (256,))
        self.r = np.ctypeslib.as_array(t.r, (self.max_row,))
        self.HT = np.ctypeslib.as_array(t.HT, (self.ht_size,))
        self.val = np.ctypeslib.as_array(t.val, (self.ht_size,))
        self.idmap = np.ctypeslib.as_array(t.idmap, (max(self.num_final, 1),))[: self.num_final]

    # -- construction -----------------------------------------------------
    @classmethod
    def from_file(cls, pattern_file: str, width: int = 256, escapes: bool = False) -> "PfacTable":
        """``escapes=True`` reads the file like the reference's ``read_pattern_ext`` (backslash escapes)."""
        L = host_lib()
        ptr = C.POINTER(CTable)()
        err = C.create_string_buffer(256)
        fn = L.pfac_table_build_file_escaped if escapes else L.pfac_table_build_file
        rc = fn(os.fsencode(pattern_file), int(width), C.byref(ptr), err, 256)
        if rc:
            raise PfacError(rc, err.value.decode(errors="replace"))
        return cls(ptr)

    @classmethod
    def from_bytes(cls, patterns: bytes, width: int = 256, part: int = 0, n_parts: int = 1) -> "PfacTable":
        L = host_lib()
        ptr = C.POINTER(CTable)()
        err = C.create_string_buffer(256)
        buf = C.create_string_buffer(patterns, len(patterns))
        rc = L.pfac_table_build_mem_part(buf, len(patterns), int(width), int(part), int(n_parts), C.byref(ptr), err, 256)
        if rc:
            raise PfacError(rc, err.value.decode(errors="replace"))
        return cls(ptr)

    @classmethod
    def from_file_part(cls, pattern_file: str, width: int, part: int, n_parts: int) -> "PfacTable":
        """Partition ``part`` of ``n_parts`` of the sorted pattern list -- the reference's pattern partitioning
        (create_table_reorder.c:217-247): every partition scans the whole input, ``merge_partitions`` merges."""
        L = host_lib()
        ptr = C.POINTER(CTable)()
        err = C.create_string_buffer(256)
        rc = L.pfac_table_build_file_part(os.fsencode(pattern_file), int(width), int(part), int(n_parts), C.byref(ptr),
                                          err, 256)
        if rc:
            raise PfacError(rc, err.value.decode(errors="replace"))
        return cls(ptr)

    @classmethod
    def from_charclass(cls, patterns, width: int = 256) -> "PfacTable":
        """Character-class pattern file (path) or image (bytes): single characters and ``[...]`` / ``[^...]`` classes
        with ``l-r`` ranges, escape-aware (charset_table_reorder.c:45-168).  The table's final states can stand for
        several patterns: ``out_first`` / ``out_ids`` list them (``idmap`` holds the first); print with
        ``emit_records_multi``."""
        from ._ffi import COutputs
        L = host_lib()
        ptr, optr = C.POINTER(CTable)(), C.POINTER(COutputs)()
        err = C.create_string_buffer(256)
        if isinstance(patterns, (bytes, bytearray)):
            buf = C.create_string_buffer(bytes(patterns), len(patterns))
            rc = L.pfac_table_build_mem_charclass(buf, len(patterns), int(width), C.byref(ptr), C.byref(optr), err, 256)
        else:
            rc = L.pfac_table_build_file_charclass(os.fsencode(patterns), int(width), C.byref(ptr), C.byref(optr), err, 256)
        if rc:
            raise PfacError(rc, err.value.decode(errors="replace"))
        t = cls(ptr)
        o = optr.contents
        t.out_first = np.ctypeslib.as_array(o.first, (o.n_states + 1,)).copy()
        t.out_ids = np.ctypeslib.as_array(o.ids, (max(int(t.out_first[-1]), 1),))[: int(t.out_first[-1])].copy()
        L.pfac_outputs_free(optr)
        return t

    @classmethod
    def from_blob(cls, blob: np.ndarray) -> "PfacTable":
        L = host_lib()
        blob = np.ascontiguousarray(blob, dtype=np.int32)
        ptr = C.POINTER(CTable)()
        rc = L.pfac_table_from_blob(blob.ctypes.data, blob.size, C.byref(ptr))
        if rc:
            raise PfacError(rc, "bad table image")
        return cls(ptr)

    @classmethod
    def from_reference_arrays(cls, s0, r, HT, val, idmap, width, state_num, num_final, ht_size, max_pat_len):
        """Wrap arrays produced by the reference's own FFDM() (main.cc:72-76,125)."""
        L = host_lib()
        arrs = [np.ascontiguousarray(a, dtype=np.int32) for a in (s0, r, HT, val, idmap)]
        ptr = C.POINTER(CTable)()
        rc = L.pfac_table_from_reference_arrays(*[a.ctypes.data for a in arrs], int(width), int(state_num),
                                                int(num_final), int(ht_size), int(max_pat_len), C.byref(ptr))
        if rc:
            raise PfacError(rc, "bad reference arrays")
        return cls(ptr)

    # -- use --------------------------------------------------------------
    def lookup(self, state: int, ch: int) -> int:
        """The device lookup (master_kernel.cu:52-63) evaluated on the host."""
        return int(host_lib().pfac_table_lookup(self._ptr, int(state), int(ch)))

    def blob(self) -> np.ndarray:
        """Flat int32 image (what gets uploaded / broadcast between ranks)."""
        L = host_lib()
        n = int(L.pfac_table_blob_words(self._ptr))
        out = np.empty(n, dtype=np.int32)
        rc = L.pfac_table_to_blob(self._ptr, out.ctypes.data, n)
        if rc:
            raise PfacError(rc, "pfac_table_to_blob")
        return out

    @property
    def halo(self) -> int:
        """Bytes a shard must be able to read past its owned range."""
        return max(self.max_pat_len - 1, 0)

    def pattern_ids(self, records: np.ndarray) -> np.ndarray:
        return self.idmap[records["state"]]

    def __del__(self):
        ptr = getattr(self, "_ptr", None)
        if ptr is not None:
            try:
                host_lib().pfac_table_free(ptr)
            except Exception:
                pass
            self._ptr = None


def merge_partitions(record_lists, idmaps=None) -> np.ndarray:
    """The reference's host merge (main.cc:304-324) on compact records: ``record_lists[k]`` are the position-sorted
    records of pattern partition k; returns one array ordered by (position, partition) whose ``state`` field holds
    the PATTERN ID (``idmaps[k]`` applied; None = the lists already hold ids)."""
    L = host_lib()
    lists = [np.ascontiguousarray(r, dtype=RECORD_DTYPE) for r in record_lists]
    k = len(lists)
    maps = [None if idmaps is None or m is None else np.ascontiguousarray(m, dtype=np.int32) for m in (idmaps or [None] * k)]
    ptrs = (C.c_void_p * max(k, 1))(*[r.ctypes.data if r.size else None for r in lists])
    counts = (C.c_uint64 * max(k, 1))(*[r.size for r in lists])
    mptrs = (C.c_void_p * max(k, 1))(*[None if m is None else m.ctypes.data for m in maps])
    total = sum(r.size for r in lists)
    out = np.empty(total, dtype=RECORD_DTYPE)
    n = L.pfac_merge_partitions(ptrs, counts, mptrs, k, out.ctypes.data, total)
    if n != total:
        raise PfacError(int(n), "pfac_merge_partitions")
    return out


def emit_records(path_or_file, records: np.ndarray, idmap, base: int = 0, append: bool = False,
                 threads: int = 1) -> int:
    """Write ``At position %4d, match pattern %d`` lines (main.cc:335-350).  Returns bytes written.
    ``threads`` > 1 uses the parallel emitter (same bytes).  ``idmap`` None: ``records["state"]`` already holds
    pattern ids (the output of ``merge_partitions``)."""
    L = host_lib()
    libc = C.CDLL(None)
    libc.fopen.restype = C.c_void_p
    libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
    libc.fclose.argtypes = [C.c_void_p]
    records = np.ascontiguousarray(records, dtype=RECORD_DTYPE)
    idmap = None if idmap is None else np.ascontiguousarray(idmap, dtype=np.int32)
    idmap_ptr = None if idmap is None else idmap.ctypes.data
    libc.fseek.argtypes = [C.c_void_p, C.c_long, C.c_int]
    # append = open for update and seek to the end (an "a" stream is O_APPEND, which defeats positioned writes)
    f = libc.fopen(os.fsencode(path_or_file), b"r+b" if append and os.path.exists(path_or_file) else b"wb")
    if not f:
        raise PfacError(-2, f"cannot open {path_or_file}")
    if append:
        libc.fseek(f, 0, 2)
    try:
        if threads > 1:
            n = L.pfac_emit_records_mt(f, records.ctypes.data, records.size, int(base), idmap_ptr, int(threads))
        else:
            n = L.pfac_emit_records(f, records.ctypes.data, records.size, int(base), idmap_ptr)
    finally:
        libc.fclose(f)
    if n < 0:
        raise PfacError(int(n), "pfac_emit_records")
    return int(n)


def emit_packed(path, words: np.ndarray, tile_index: np.ndarray, idmap, base: int = 0, threads: int = 1) -> int:
    """The same text straight from the compact device form (``GpuMatcher.packed_to_host``): record heap + ordered tile
    index (include/pfac.h).  Returns bytes written."""
    L = host_lib()
    libc = C.CDLL(None)
    libc.fopen.restype = C.c_void_p
    libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
    libc.fclose.argtypes = [C.c_void_p]
    words = np.ascontiguousarray(words)
    if words.dtype not in (np.uint16, np.uint32):
        raise TypeError("compact records are uint16 or uint32 words")
    tix = np.ascontiguousarray(tile_index, dtype=np.uint64)
    idmap = None if idmap is None else np.ascontiguousarray(idmap, dtype=np.int32)
    f = libc.fopen(os.fsencode(path), b"wb")
    if not f:
        raise PfacError(-2, f"cannot open {path}")
    try:
        n = L.pfac_emit_packed(f, words.ctypes.data, int(words.size), int(words.dtype.itemsize), tix.ctypes.data, tix.size, int(base),
                               None if idmap is None else idmap.ctypes.data, int(threads))
    finally:
        libc.fclose(f)
    if n < 0:
        raise PfacError(int(n), "pfac_emit_packed")
    return int(n)


def emit_records_multi(path, records: np.ndarray, table: "PfacTable", base: int = 0) -> int:
    """Text for a character-class table: one line per (record, pattern that ends in the record's final state)."""
    from ._ffi import COutputs
    L = host_lib()
    libc = C.CDLL(None)
    libc.fopen.restype = C.c_void_p
    libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
    libc.fclose.argtypes = [C.c_void_p]
    records = np.ascontiguousarray(records, dtype=RECORD_DTYPE)
    first = np.ascontiguousarray(table.out_first, dtype=np.int32)
    ids = np.ascontiguousarray(table.out_ids if table.out_ids.size else np.zeros(1, np.int32), dtype=np.int32)
    o = COutputs(int(first.size - 1), first.ctypes.data_as(C.POINTER(C.c_int32)), ids.ctypes.data_as(C.POINTER(C.c_int32)))
    f = libc.fopen(os.fsencode(path), b"wb")
    if not f:
        raise PfacError(-2, f"cannot open {path}")
    try:
        n = L.pfac_emit_records_multi(f, records.ctypes.data, records.size, int(base), C.byref(o))
    finally:
        libc.fclose(f)
    if n < 0:
        raise PfacError(int(n), "pfac_emit_records_multi")
    return int(n)
