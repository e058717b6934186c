"""Whole-shard parity at BASELINE size (run with -m gpu): for every bench workload at 1 GiB, and for the C4 / C5
pattern sets at their 4 GiB shard size, the GPU's (match count, record checksum) over the WHOLE resident buffer must
equal one serial Aho-Corasick pass (oracle/ac_serial.c, the checker) over the very same bytes, copied back from HBM.

Reference semantics being checked: every (start offset, pattern) the walk of master_kernel.cu:37-74 reports and
main.cc:341-349 prints.  The checksum is a sum of 64-bit hashes of (position, pattern id) -- one missing, doubled or
mis-attributed record anywhere in the shard changes it.  Record ORDER is what the golden and prefix tests pin
(tests/test_gpu_parity.py); here every record of the shard takes part.  Integer work: bit-exact.
"""
import os

import numpy as np
import pytest

import bench
from orc import ac_whole_shard
from phfpfac_amd import GpuMatcher, PfacTable

pytestmark = pytest.mark.gpu

GIB = 1 << 30
SEED = 0x5048465046414331


def gpu_count_checksum(g, table, kind, para, n_owned, n_avail, lo):
    """Fill a device buffer as bench.py does for the shard that starts at global offset `lo`, scan it, and return
    (count, checksum with global positions, host copy of the bytes the kernel could read)."""
    import torch
    buf = torch.empty(n_avail + 4096, dtype=torch.uint8, device="cuda:0")
    if kind == "text":
        g.fill_tiled(buf, n_avail, para, phase=lo % len(para))
    else:
        g.fill_random(buf, (n_avail + 7) // 8 * 8, SEED + lo // 8)
    g.reserve(0, 0, max(n_owned // 8, 1 << 20))
    n = g.scan_resident(n_owned, n_avail, d_input=buf)
    chk = g.checksum(n, base=lo)
    host = buf[:n_avail].cpu().numpy()
    del buf
    torch.cuda.empty_cache()
    return n, chk, host


@pytest.mark.parametrize("name", sorted(bench.WORKLOADS))
def test_whole_shard_1gib_equals_serial_ac(name, tmp_path):
    """All six bench workloads at BASELINE config-2 size (1 GiB resident), n_owned == n_avail."""
    pat_name, kind, _ = bench.WORKLOADS[name]
    ppath = bench.pattern_path(pat_name, str(tmp_path))
    para = open(os.path.join(bench.DATA, "paragraph402"), "rb").read()
    table = PfacTable.from_file(ppath, 256)
    with GpuMatcher(0, 1) as g:
        g.load_table(table)
        n, chk, host = gpu_count_checksum(g, table, kind, para, GIB, GIB, 0)
    cnt, want = ac_whole_shard(ppath, host)
    assert (n, chk) == (cnt, want)


@pytest.mark.parametrize("name", ["text1g_experimentpattern", "rand1g_snort75k"])
def test_whole_shard_4gib_with_halo_equals_serial_ac(name, tmp_path):
    """The C4 / C5 shard as rank 3 of 8 sees it: n_owned = 2^32 start offsets, max_pat_len-1 bytes of halo behind them
    that are read but start no match, global positions (base 3 * 2^32) in the checksum."""
    pat_name, kind, _ = bench.WORKLOADS[name]
    ppath = bench.pattern_path(pat_name, str(tmp_path))
    para = open(os.path.join(bench.DATA, "paragraph402"), "rb").read()
    table = PfacTable.from_file(ppath, 256)
    n_owned = 1 << 32
    n_avail = n_owned + table.halo
    lo = 3 * n_owned
    with GpuMatcher(0, 1) as g:
        g.load_table(table)
        n, chk, host = gpu_count_checksum(g, table, kind, para, n_owned, n_avail, lo)
    cnt, want = ac_whole_shard(ppath, host, n_owned=n_owned, base=lo)
    assert (n, chk) == (cnt, want)


def test_second_scan_in_adapted_staging_mode_keeps_parity(tmp_path):
    """The dictionary on text makes the context switch to dense staging after its first scan;
    the scan that runs in the adapted mode -- dense mode's second form, at BASELINE size -- must produce the same
    whole-shard count and checksum."""
    import torch
    pat_name, kind, _ = bench.WORKLOADS["text1g_dictionary"]
    ppath = bench.pattern_path(pat_name, str(tmp_path))
    para = open(os.path.join(bench.DATA, "paragraph402"), "rb").read()
    table = PfacTable.from_file(ppath, 256)
    N = GIB
    with GpuMatcher(0, 1) as g:
        g.load_table(table)
        buf = torch.empty(N + 4096, dtype=torch.uint8, device="cuda:0")
        g.fill_tiled(buf, N, para)
        g.reserve(0, 0, N // 2)
        before = g.info()["staging_buffers"]
        n1 = g.scan_resident(N, N, d_input=buf)
        c1 = g.checksum(n1)
        after = g.info()["staging_buffers"]
        after_records = g.info()["staging_records"]
        n2 = g.scan_resident(N, N, d_input=buf)
        c2 = g.checksum(n2)
        host = buf[:N].cpu().numpy()
    cnt, want = ac_whole_shard(ppath, host)
    assert before != after == 1          # sparse layout first, dense (one buffer) afterwards
    assert after_records == 4096         # ... in its second form (refilled walker slots, record log): not the classic fallback
    assert (n1, c1) == (cnt, want) and (n2, c2) == (cnt, want)
