"""ctypes bindings of the C-ABI declared in include/pfac.h.

Two in-tree shared libraries are bound:

* ``lib/libpfac_host.so`` -- plain C, pattern file -> PHF-compressed table (no GPU needed)
* ``lib/libpfac_hip.so``  -- HIP/gfx950: contexts, upload, the scan kernel, records

There is NO CPU fallback for the scan: if ``libpfac_hip.so`` is missing or no
GPU is usable the calls raise (``PfacError`` / ``OSError``) instead of computing
anything on the host.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(_HERE, "lib")

PFAC_OK = 0
PFAC_E_ARG = -1
PFAC_E_IO = -2
PFAC_E_PATTERN = -3
PFAC_E_NOMEM = -4
PFAC_E_NO_DEVICE = -5
PFAC_E_HIP = -6
PFAC_E_STATE = -7
PFAC_E_OVERFLOW = -8
PFAC_E_INTERNAL = -9

_STATUS_NAMES = {
    0: "PFAC_OK", -1: "PFAC_E_ARG", -2: "PFAC_E_IO", -3: "PFAC_E_PATTERN", -4: "PFAC_E_NOMEM",
    -5: "PFAC_E_NO_DEVICE", -6: "PFAC_E_HIP", -7: "PFAC_E_STATE", -8: "PFAC_E_OVERFLOW", -9: "PFAC_E_INTERNAL",
}


class PfacError(RuntimeError):
    """A C-ABI call returned a negative pfac_status."""

    def __init__(self, status: int, message: str = ""):
        self.status = status
        name = _STATUS_NAMES.get(status, str(status))
        super().__init__(f"{name}: {message}" if message else name)


class CTable(C.Structure):
    """struct pfac_table (include/pfac.h)."""
    _fields_ = [(n, C.c_int32) for n in
                ("width", "width_bit", "n_patterns", "num_final", "state_num", "max_pat_len", "max_row",
                 "ht_size", "n_keys")] + \
               [(n, C.POINTER(C.c_int32)) for n in ("s0", "r", "HT", "val", "idmap")]


class COutputs(C.Structure):
    """struct pfac_outputs (character-class tables: final state -> pattern ids)."""
    _fields_ = [("n_states", C.c_int32), ("first", C.POINTER(C.c_int32)), ("ids", C.POINTER(C.c_int32))]


class CRecord(C.Structure):
    """struct pfac_record: start offset relative to the scanned range + final state."""
    _fields_ = [("pos", C.c_uint32), ("state", C.c_uint32)]


class CThreadData(C.Structure):
    """struct pfac_thread_data == the reference's struct thread_data (main.cc:19-32)."""
    _fields_ = [("input_string", C.c_void_p), ("input_size", C.c_int), ("state_num", C.c_int),
                ("final_state_num", C.c_int), ("match_result", C.c_void_p), ("HTSize", C.c_int),
                ("width", C.c_int), ("s0Table", C.c_void_p), ("max_pat_len", C.c_int), ("r", C.c_void_p),
                ("HT", C.c_void_p), ("val", C.c_void_p)]


HOST_SYMBOLS = (
    "pfac_table_build_file", "pfac_table_build_file_escaped", "pfac_table_build_mem", "pfac_table_build_file_part",
    "pfac_table_build_mem_part", "pfac_merge_partitions", "pfac_table_free", "pfac_table_lookup",
    "pfac_table_blob_words", "pfac_table_to_blob", "pfac_table_from_blob", "pfac_table_from_reference_arrays",
    "pfac_emit_records", "pfac_emit_records_mt", "pfac_emit_packed", "pfac_table_build_file_charclass",
    "pfac_table_build_mem_charclass", "pfac_outputs_free", "pfac_emit_records_multi",
)
HIP_SYMBOLS = (
    "pfac_device_count", "pfac_ctx_create", "pfac_ctx_destroy", "pfac_last_error", "pfac_table_upload",
    "pfac_table_upload_device", "pfac_host_alloc", "pfac_host_free", "pfac_slot_reserve", "pfac_slot_input",
    "pfac_slot_records", "pfac_slot_stream", "pfac_slot_set_stream", "pfac_slot_h2d", "pfac_scan_async",
    "pfac_scan_finish", "pfac_scan_elapsed_ms", "pfac_records_d2h", "pfac_slot_sync", "pfac_records_checksum",
    "pfac_fill_tiled", "pfac_fill_random", "pfac_scan_info", "pfac_scan_staging", "pfac_trace_table_compat", "pfac_scan_format",
    "pfac_records_expand", "pfac_records_d2h_packed", "pfac_scan_capacity_hint", "pfac_records_packed_device",
    "pfac_emit_text_device", "pfac_text_d2h", "pfac_slot_text", "pfac_slot_h2d_wait", "pfac_slot_h2d_done", "pfac_host_register", "pfac_host_unregister",
)

_host = None
_hip = None


def host_lib() -> C.CDLL:
    """libpfac_host.so (table builder + emitter)."""
    global _host
    if _host is None:
        path = os.path.join(LIB_DIR, "libpfac_host.so")
        if not os.path.exists(path):
            raise OSError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          f"or `make -C phfpfac_amd/csrc`")
        L = C.CDLL(path)
        TP = C.POINTER(CTable)
        L.pfac_table_build_file.argtypes = [C.c_char_p, C.c_int, C.POINTER(TP), C.c_char_p, C.c_size_t]
        L.pfac_table_build_file_escaped.argtypes = [C.c_char_p, C.c_int, C.POINTER(TP), C.c_char_p, C.c_size_t]
        L.pfac_table_build_mem.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.POINTER(TP), C.c_char_p, C.c_size_t]
        L.pfac_table_build_file_part.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(TP), C.c_char_p, C.c_size_t]
        L.pfac_table_build_mem_part.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.POINTER(TP), C.c_char_p,
                                                C.c_size_t]
        L.pfac_merge_partitions.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_uint64]
        L.pfac_merge_partitions.restype = C.c_int64
        L.pfac_table_free.argtypes = [TP]
        L.pfac_table_free.restype = None
        L.pfac_table_lookup.argtypes = [TP, C.c_int32, C.c_int32]
        L.pfac_table_lookup.restype = C.c_int32
        L.pfac_table_blob_words.argtypes = [TP]
        L.pfac_table_blob_words.restype = C.c_size_t
        L.pfac_table_to_blob.argtypes = [TP, C.c_void_p, C.c_size_t]
        L.pfac_table_from_blob.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(TP)]
        L.pfac_table_from_reference_arrays.argtypes = [C.c_void_p] * 5 + [C.c_int32] * 5 + [C.POINTER(TP)]
        L.pfac_emit_records.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]
        L.pfac_emit_records.restype = C.c_int64
        L.pfac_emit_records_mt.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_int]
        L.pfac_emit_records_mt.restype = C.c_int64
        L.pfac_emit_packed.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_int]
        L.pfac_emit_packed.restype = C.c_int64
        OP = C.POINTER(COutputs)
        L.pfac_table_build_file_charclass.argtypes = [C.c_char_p, C.c_int, C.POINTER(TP), C.POINTER(OP), C.c_char_p, C.c_size_t]
        L.pfac_table_build_mem_charclass.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.POINTER(TP), C.POINTER(OP), C.c_char_p, C.c_size_t]
        L.pfac_outputs_free.argtypes = [OP]
        L.pfac_outputs_free.restype = None
        L.pfac_emit_records_multi.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, OP]
        L.pfac_emit_records_multi.restype = C.c_int64
        _host = L
    return _host


def hip_lib() -> C.CDLL:
    """libpfac_hip.so (the GPU path).  Raises OSError when it was not built."""
    global _hip
    if _hip is None:
        path = os.environ.get("PFAC_HIP_LIB") or os.path.join(LIB_DIR, "libpfac_hip.so")   # override: A/B tuning builds
        if not os.path.exists(path):
            raise OSError(f"{path} is missing: the HIP extension was not built and there is no CPU fallback "
                          f"(run __graft_entry__.build() or `make -C phfpfac_amd/csrc`)")
        # One HIP runtime per process: libpfac_hip.so needs libamdhip64.so.7 by soname, and PyTorch
        # bundles its own copy.  If ours pulled in /opt/rocm's first, torch.cuda would later find a
        # second, conflicting HSA runtime ("No HIP GPUs are available").  Importing torch first makes
        # the loader bind this library to the copy torch already mapped.  (The C CLI links /opt/rocm.)
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(path)
        vp, u64, i = C.c_void_p, C.c_uint64, C.c_int
        L.pfac_device_count.argtypes = [C.POINTER(i)]
        L.pfac_ctx_create.argtypes = [i, i, C.POINTER(vp)]
        L.pfac_ctx_destroy.argtypes = [vp]
        L.pfac_ctx_destroy.restype = None
        L.pfac_last_error.argtypes = [vp]
        L.pfac_last_error.restype = C.c_char_p
        L.pfac_table_upload.argtypes = [vp, vp, C.c_size_t]
        L.pfac_table_upload_device.argtypes = [vp, vp, C.c_size_t, vp]
        L.pfac_host_alloc.argtypes = [C.POINTER(vp), C.c_size_t]
        L.pfac_host_register.argtypes = [vp, C.c_size_t]
        L.pfac_host_unregister.argtypes = [vp]
        L.pfac_host_free.argtypes = [vp]
        L.pfac_host_free.restype = None
        L.pfac_slot_reserve.argtypes = [vp, i, u64, u64]
        L.pfac_slot_input.argtypes = [vp, i]
        L.pfac_slot_input.restype = vp
        L.pfac_slot_records.argtypes = [vp, i]
        L.pfac_slot_records.restype = vp
        L.pfac_slot_stream.argtypes = [vp, i]
        L.pfac_slot_stream.restype = vp
        L.pfac_slot_set_stream.argtypes = [vp, i, vp]
        L.pfac_slot_h2d.argtypes = [vp, i, vp, u64, u64]
        L.pfac_slot_h2d_wait.argtypes = [vp, i]
        L.pfac_slot_h2d_done.argtypes = [vp, i]
        L.pfac_scan_async.argtypes = [vp, i, vp, u64, u64, vp, u64]
        L.pfac_scan_finish.argtypes = [vp, i, C.POINTER(u64)]
        L.pfac_scan_elapsed_ms.argtypes = [vp, i, C.POINTER(C.c_float)]
        L.pfac_records_d2h.argtypes = [vp, i, vp, vp, u64, u64]
        L.pfac_scan_format.argtypes = [vp, i, C.POINTER(i), C.POINTER(u64), C.POINTER(u64)]
        L.pfac_scan_capacity_hint.argtypes = [vp, i, C.POINTER(u64)]
        L.pfac_records_expand.argtypes = [vp, i, vp, u64, u64, vp]
        L.pfac_records_d2h_packed.argtypes = [vp, i, vp, vp, u64, vp]
        L.pfac_records_packed_device.argtypes = [vp, i, vp, vp, u64, vp]
        L.pfac_emit_text_device.argtypes = [vp, i, vp, u64, C.POINTER(u64)]
        L.pfac_text_d2h.argtypes = [vp, i, vp, u64, u64]
        L.pfac_slot_text.argtypes = [vp, i]
        L.pfac_slot_text.restype = vp
        L.pfac_slot_sync.argtypes = [vp, i]
        L.pfac_records_checksum.argtypes = [vp, i, vp, u64, u64, C.POINTER(u64)]
        L.pfac_fill_tiled.argtypes = [vp, i, vp, u64, vp, C.c_uint32, u64]
        L.pfac_fill_random.argtypes = [vp, i, vp, u64, u64]
        L.pfac_scan_info.argtypes = [vp, C.POINTER(i), C.POINTER(i), C.POINTER(i), C.POINTER(i)]
        L.pfac_scan_staging.argtypes = [vp, C.POINTER(i), C.POINTER(C.c_uint32)]
        L.pfac_trace_table_compat.argtypes = [C.POINTER(CThreadData), i]
        _hip = L
    return _hip
