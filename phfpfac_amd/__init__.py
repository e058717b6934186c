"""phfpfac_amd -- MI355X-native PFAC (Parallel Failureless Aho-Corasick) multi-pattern matcher.

Host side (C, ``csrc/pfac_table.c``) builds the perfect-hash-compressed state-transition table;
the scan runs in a hand-written gfx950 HIP kernel (``csrc/pfac_hip.hip``) behind the C-ABI of
``include/pfac.h``.  This package is the thin Python mirror of that ABI plus the
``torch.distributed`` plumbing that shards the input byte stream across GPUs.
"""
from ._ffi import PfacError  # noqa: F401
from .table import RECORD_DTYPE, PfacTable, emit_packed, emit_records, emit_records_multi, merge_partitions  # noqa: F401
from .matcher import GpuMatcher, device_count  # noqa: F401

__all__ = ["PfacError", "PfacTable", "GpuMatcher", "RECORD_DTYPE", "emit_records", "emit_packed", "emit_records_multi", "merge_partitions", "device_count"]
