"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C-ABI of
include/pfac.h, against (a) the committed golden outputs of the reference pipeline and (b) the CPU
oracle on the same seeded inputs.  Everything here is integer/byte work: the bar is BIT-EXACT.
"""
import hashlib
import json
import os

import numpy as np
import pytest

from orc import Oracle, match_checksum
from phfpfac_amd import GpuMatcher, PfacError, PfacTable, emit_records
from phfpfac_amd.matcher import splitmix64_bytes, tiled_bytes, trace_table_compat

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
FP = json.load(open(os.path.join(HERE, "golden", "fingerprints.json")))
TILE = 16384


def gpu_records(table, data, n_owned=None, n_streams=1):
    with GpuMatcher(0, n_streams) as g:
        g.load_table(table)
        return g.scan_bytes(data, n_owned)


def oracle_pairs(pattern_file, data, n=None):
    o = Oracle(pattern_file, 1, 1)          # one automaton; the spec walk needs no PHF
    pos, ids = o.scan_spec(data, n)
    o.close()
    return pos, ids


def assert_same(table, rec, pos, ids):
    assert rec.size == pos.size, f"match count {rec.size} != oracle {pos.size}"
    np.testing.assert_array_equal(rec["pos"].astype(np.int64), pos)
    np.testing.assert_array_equal(table.idmap[rec["state"]], ids)


# ---------------------------------------------------------------------------
@pytest.mark.parametrize("case", sorted(FP["cases"]))
def test_golden_outputs_byte_identical(case, resolve, tmp_path):
    """GPU_match_result.txt of the whole pipeline == the reference's, for every golden case
    (the output is independent of stream count and PHF width inside the parity domain)."""
    c = FP["cases"][case]
    table = PfacTable.from_file(resolve(c["pattern"]), c["width"])
    raw = open(resolve(c["input"]), "rb").read()
    data = raw[:-1]                                           # the reference drops the last byte (main.cc:138)
    rec = gpu_records(table, data, n_streams=c["streams"])
    out = tmp_path / "GPU_match_result.txt"
    nbytes = emit_records(str(out), rec, table.idmap)
    blob = out.read_bytes()
    assert rec.size == c["lines"]
    assert nbytes == len(blob) == c["bytes"]
    assert hashlib.md5(blob).hexdigest() == c["md5"]
    if c["verbatim"]:
        assert blob == open(os.path.join(HERE, "golden", "out", case + ".txt"), "rb").read()


def test_gphf_cli_config1(resolve, tmp_path):
    """The C driver end to end: gphf experimentpattern 2 256 1M -> the BASELINE config-1 golden."""
    import subprocess
    exe = os.path.join(os.path.dirname(HERE), "phfpfac_amd", "bin", "gphf")
    env = dict(os.environ, PFAC_CHUNK_MB="1")                  # force several pipeline chunks on a 1 MiB input
    subprocess.check_call([exe, resolve("experimentpattern"), "2", "256", resolve("1M")], cwd=tmp_path, env=env,
                          stdout=subprocess.DEVNULL)
    blob = (tmp_path / "GPU_match_result.txt").read_bytes()
    assert hashlib.md5(blob).hexdigest() == FP["cases"]["exp_x_1M_s1_w256"]["md5"]
    r = subprocess.run([exe, "a", "b"], cwd=tmp_path, capture_output=True)
    assert r.returncode != 0 and b"usage:" in r.stderr        # argc check, main.cc:93-96
    # streaming ingest + in-order emitter over many chunks: 6 MiB of text, 1 MiB chunks, 3 pipeline slots,
    # dictionary patterns (matches straddle chunk boundaries); expected file from the CPU oracle
    para = open(resolve("paragraph402"), "rb").read()
    big = tmp_path / "big.txt"
    n = 6 * (1 << 20) + 12345
    big.write_bytes(tiled_bytes(n + 1, para).tobytes())        # + the byte the CLI drops (main.cc:138)
    env5 = dict(env, PFAC_READ_THREADS="5", PFAC_EMIT="host")  # a pool of 5 readers; the host formatter
    subprocess.check_call([exe, resolve("xaa"), "3", "1024", str(big)], cwd=tmp_path, env=env5, stdout=subprocess.DEVNULL)
    o = Oracle(resolve("xaa"), 1, 1)
    exp = tmp_path / "expected.txt"
    o.emit(tiled_bytes(n, para), str(exp), spec=True)
    o.close()
    assert (tmp_path / "GPU_match_result.txt").read_bytes() == exp.read_bytes()


@pytest.mark.parametrize("workers,streams,emit", [(2, 1, "device"), (3, 2, "host"), (4, 3, "device")])
def test_gphf_several_workers_share_the_emitter(workers, streams, emit, resolve, tmp_path):
    """The CLI's multi-GPU path (one host thread + context per worker, chunks dealt round-robin over the workers,
    ONE in-order emitter, the bounded window between them -- gphf.c, replacing the OpenMP fan-out of main.cc:180-241)
    with MORE THAN ONE worker: PFAC_WORKERS_PER_GPU runs them on this box's single device.  16 chunks of 1 MiB of the
    reference's text under the 2 600-word dictionary (matches straddle every chunk boundary); the file must equal
    the oracle's text byte for byte, whichever worker finished first."""
    import subprocess
    exe = os.path.join(os.path.dirname(HERE), "phfpfac_amd", "bin", "gphf")
    para = open(resolve("paragraph402"), "rb").read()
    n = 16 * (1 << 20) - 777
    big = tmp_path / "big16.txt"
    big.write_bytes(tiled_bytes(n + 1, para).tobytes())        # + the byte the CLI drops (main.cc:138)
    env = dict(os.environ, PFAC_CHUNK_MB="1", PFAC_WORKERS_PER_GPU=str(workers), PFAC_READ_THREADS="2", PFAC_EMIT_THREADS="4",
               PFAC_EMIT=emit)
    out = subprocess.run([exe, resolve("xaa"), str(streams), "256", str(big)], cwd=tmp_path, env=env, capture_output=True,
                         text=True, check=True).stdout
    assert f"({workers} worker(s);" in out
    assert sum(l.startswith("5.worker") for l in out.splitlines()) == workers
    o = Oracle(resolve("xaa"), 1, 1)
    exp = tmp_path / "expected.txt"
    o.emit(tiled_bytes(n, para), str(exp), spec=True)
    o.close()
    assert (tmp_path / "GPU_match_result.txt").read_bytes() == exp.read_bytes()


def test_bench_rank_code_at_the_c4_shard_size(tmp_path):
    """Exactly what the driver's scaling run executes per rank: `bench.py --gpus 1` with the RCCL path forced
    (table broadcast, count all-gather, compact record gather) at the C4 / C5 shard size of 4 GiB per GPU.  Its own
    whole-shard parity check (count + checksum == serial Aho-Corasick over the 4 GiB copied back) must pass."""
    import subprocess
    import sys
    repo = os.path.dirname(HERE)
    env = dict(os.environ, PFAC_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29547")
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                        "--bytes-per-gpu", str(1 << 32), "--no-cpu-baseline", "--no-extra", "--no-end-to-end", "--sustain-seconds", "0.5"], env=env, cwd=tmp_path,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["config"]["bytes_per_gpu"] == 1 << 32
    assert line["config"]["parity"].startswith("whole shard: count + checksum == serial AC")
    assert line["config"]["matches_per_step"] > 300_000_000
    assert "gather_ms" in line["config"] and line["value"] > 1000


@pytest.mark.parametrize("width,env", [(256, {}), (1024, {}), (64, {}), (256, {"PFAC_NO_D1PACK": "1"}),
                                       (4096, {"PFAC_NO_NW4": "1"}), (256, {"PFAC_NO_FUSE": "1"}), (256, {"PFAC_NO_D1": "1"}),
                                       (256, {"PFAC_NO_DENSE2": "1"}), (2048, {"PFAC_NO_DENSE2": "1"}),
                                       (256, {"PFAC_D2_LOGCAP": "64"}), (512, {"PFAC_D2_LOGCAP": "1500"}), (256, {"PFAC_NWB": "4"})])
def test_dictionary_dense_mode_kernels(width, env, resolve, monkeypatch):
    """Dense staging mode on L2 tables: its second form (refilled walker slots, records ordered on the way out: fused
    tables with packed dense rows, width >= 256), its fallback pass (a record log too small for the tile), the classic
    fused-slot kernels with four walks per lane, the unfused two-walk kernel (width 64) and every fallback knob -- all
    must give the oracle's records."""
    monkeypatch.setenv("PFAC_DENSE", "1")
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    table = PfacTable.from_file(resolve("xaa+xab+xac+xad"), width)
    data = open(resolve("1M"), "rb").read()[:200001]
    rec = gpu_records(table, data)
    pos, ids = oracle_pairs(resolve("xaa+xab+xac+xad"), data)
    assert_same(table, rec, pos, ids)


def test_dense_mode_more_than_15_patterns_at_one_offset(tmp_path, monkeypatch):
    """The second form of dense mode counts a position's records in 4 bits: a tile where 16 or more patterns start at
    one offset is done again the classic way (nested prefixes a, aa, aaa, ... on runs of a), and so is a tile with more
    records than the wave's log holds; tiles without such an offset in the same input keep the fast path."""
    monkeypatch.setenv("PFAC_DENSE", "1")
    monkeypatch.setenv("PFAC_FORCE_L2", "1")
    pats = [b"a" * k for k in range(1, 25)] + [b"ab", b"abc", b"bca", b"cab", b"b", b"zq", b"qzq"]
    pf = tmp_path / "nested.pat"
    pf.write_bytes(b"\n".join(pats) + b"\n")
    rng = np.random.default_rng(5)
    data = rng.choice(np.frombuffer(b"abcqz", dtype=np.uint8), 300_000).astype(np.uint8)
    data[5000:5040] = ord("a")                   # one run of 40: 24 patterns at its first offsets
    data[150_000:150_018] = ord("a")             # 18 patterns at one offset
    data[220_000:220_015] = ord("a")             # exactly 15: still the fast path
    table = PfacTable.from_file(str(pf), 256)
    rec = gpu_records(table, data.tobytes())
    pos, ids = oracle_pairs(str(pf), data.tobytes())
    assert_same(table, rec, pos, ids)
    assert np.bincount(pos).max() == 24


@pytest.mark.parametrize("env", [{"PFAC_FORCE_L2": "1"}, {"PFAC_FORCE_L2": "1", "PFAC_DENSE": "1"},
                                 {"PFAC_FORCE_L2": "1", "PFAC_NO_D1": "1", "PFAC_DENSE": "1"},
                                 {"PFAC_FORCE_L2": "1", "PFAC_NO_FUSE": "1"}, {"PFAC_NWB": "5", "PFAC_DENSE": "1"},
                                 {"PFAC_NWB": "3"}, {"PFAC_NO_D1": "1"},
                                 {"PFAC_WIDE": "1"},                      # 8-byte records in the heap (automata beyond 2^20 final states)
                                 {"PFAC_REC_BYTES": "4"},                 # 32-bit records where 16 bits would do
                                 {"PFAC_WIDE": "1", "PFAC_FORCE_L2": "1", "PFAC_DENSE": "1"},
                                 {"PFAC_LAG": "1"}, {"PFAC_LAG": "2"},    # two / three staging buffers (emission one / two rounds late)
                                 {"PFAC_LAG": "2", "PFAC_REC_BYTES": "4"},
                                 {"PFAC_L2F": "0"}, {"PFAC_L2F": "2"},    # level-2 filter off / lookup form where the SWAR form applies
                                 {"PFAC_L2F": "2", "PFAC_FORCE_L2": "1"}, {"PFAC_NO_SECF": "1", "PFAC_FORCE_L2": "1"}])
@pytest.mark.parametrize("case", ["exp_x_1M_s1_w256", "xaa_x_1M_s1_w256", "all_x_1M_s3_w1024"])
def test_golden_under_every_kernel_variant(case, env, resolve, tmp_path, monkeypatch):
    """The tuning knobs select other kernel instantiations / LDS layouts / record forms (tables through L2, fused slots,
    four walks per lane, dense staging, fewer waves, no dense rows, 8-byte records, every form of the level-2 filter);
    the golden files must come out of every one of them."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    c = FP["cases"][case]
    table = PfacTable.from_file(resolve(c["pattern"]), c["width"])
    data = open(resolve(c["input"]), "rb").read()[:-1]
    rec = gpu_records(table, data)
    out = tmp_path / "GPU_match_result.txt"
    emit_records(str(out), rec, table.idmap)
    assert hashlib.md5(out.read_bytes()).hexdigest() == c["md5"]


@pytest.mark.parametrize("n_parts", [4, 7])
def test_pattern_partition_mode_golden(n_parts, resolve, tmp_path):
    """Pattern-partition fallback (the reference's own scheme, ctr.c:217-247 + main.cc:304-324): every partition's
    table scans the whole input on the GPU, pfac_merge_partitions merges -> the same GPU_match_result.txt."""
    c = FP["cases"]["all_x_1M_s1_w256"]
    tabs = [PfacTable.from_file_part(resolve(c["pattern"]), 256, k, n_parts) for k in range(n_parts)]
    data = open(resolve(c["input"]), "rb").read()[:-1]
    with GpuMatcher(0, 1) as g:
        merged = g.scan_partitioned(tabs, data)
    out = tmp_path / "GPU_match_result.txt"
    emit_records(str(out), merged, None)
    blob = out.read_bytes()
    assert merged.size == c["lines"] and len(blob) == c["bytes"]
    assert hashlib.md5(blob).hexdigest() == c["md5"]


def test_pattern_partition_mode_edge(resolve):
    """More partitions than patterns (empty partitions scan with an edgeless root) and duplicates at a cut."""
    pats = b"ab\nab\nabc\nb\nb\nca\n"
    data = (b"abcab cabbabc" * 700)[:8191]
    full = PfacTable.from_bytes(pats, 256)
    one = gpu_records(full, data)
    for n_parts in (2, 3, 9):
        tabs = [PfacTable.from_bytes(pats, 256, part=k, n_parts=n_parts) for k in range(n_parts)]
        with GpuMatcher(0, 1) as g:
            merged = g.scan_partitioned(tabs, data)
        assert merged.size == one.size
        np.testing.assert_array_equal(merged["pos"], one["pos"])
        np.testing.assert_array_equal(merged["state"].astype(np.int32), full.idmap[one["state"]])


@pytest.mark.parametrize("width", [256, 64, 1024, 4096])
def test_dictionary_vs_oracle_widths(width, resolve, work_dir):
    """7 989-word dictionary (tables via L2) on text, every PHF width: records == oracle, in order."""
    table = PfacTable.from_file(resolve("xaa+xab+xac+xad"), width)
    data = open(resolve("1M"), "rb").read()[:300000]
    rec = gpu_records(table, data)
    pos, ids = oracle_pairs(resolve("xaa+xab+xac+xad"), data)
    assert_same(table, rec, pos, ids)


def test_edge_cases(tmp_path):
    """Edge cases the reference handles in specific ways."""
    pat = tmp_path / "p"
    # prefix chain, single-byte pattern, high bytes, duplicate line (last wins), pattern with \r
    pat.write_bytes(b"ab\nabcd\nb\nab\n\xff\xfe\nxyz\r\nabcdefghijklmnop\n")
    table = PfacTable.from_file(str(pat), 256)
    cases = [
        b"",                                   # empty input
        b"b",                                  # one byte
        b"ab",                                 # match ending exactly at N
        b"abc",                                # "abcd" would run past N: must not be reported
        b"xxabcdxx\xff\xfe\xffab",             # several, high bytes
        b"xyz\r" + b"ab" * 40,
        (b"abcdefghijklmnop" * 3000)[:TILE * 2 + 5],     # matches straddling tile boundaries
        b"q" * (TILE - 1) + b"abcd",           # pattern starts in the last byte of a tile
        b"q" * (TILE - 3) + b"abcdefghijklmnop" + b"q" * 7,
    ]
    for data in cases:
        rec = gpu_records(table, data)
        pos, ids = oracle_pairs(str(pat), data)
        assert_same(table, rec, pos, ids)
    # duplicate rule: "ab" is lines 1 and 4 -> reported as 4 (create_table_reorder.c:366)
    rec = gpu_records(table, b"ab")
    assert [(int(p), int(i)) for p, i in zip(rec["pos"], table.idmap[rec["state"]])] == [(0, 4), (1, 3)]


def test_ragged_sizes_and_alignment(resolve):
    """Every input length around tile / 16-byte boundaries (the buffer-load tail path)."""
    table = PfacTable.from_file(resolve("experimentpattern"), 256)
    base = tiled_bytes(3 * TILE + 64, b"aaaab aa a xaaaaaaa")
    with GpuMatcher(0, 1) as g:
        g.load_table(table)
        for n in [1, 2, 15, 16, 17, 1023, 1024, 1025, 4095, 4097, TILE - 1, TILE, TILE + 1, TILE + 15, TILE + 17,
                  2 * TILE - 1, 2 * TILE, 3 * TILE + 33]:
            rec = g.scan_bytes(base[:n])
            pos, ids = oracle_pairs(resolve("experimentpattern"), base[:n])
            assert_same(table, rec, pos, ids)


def test_owned_range_and_shard_invariance(resolve):
    """Scanning [0,N) in one piece == concatenating K shards that own disjoint ranges and read
    max_pat_len-1 bytes of halo; shard boundaries land inside matches."""
    table = PfacTable.from_file(resolve("xaa+xab+xac+xad"), 256)
    data = np.frombuffer(open(resolve("1M"), "rb").read()[:200003], dtype=np.uint8)
    N = data.size
    with GpuMatcher(0, 1) as g:
        g.load_table(table)
        whole = g.scan_bytes(data)
        for K in (2, 3, 8):
            per = -(-N // K)
            parts = []
            for k in range(K):
                lo, hi = k * per, min(N, (k + 1) * per)
                end = min(N, hi + table.halo)
                rec = g.scan_bytes(data[lo:end], n_owned=hi - lo).copy()
                rec["pos"] += np.uint32(lo)
                parts.append(rec)
            cat = np.concatenate(parts)
            np.testing.assert_array_equal(cat["pos"], whole["pos"])
            np.testing.assert_array_equal(cat["state"], whole["state"])


def test_random_patterns_vs_oracle(tmp_path):
    """Seeded random pattern sets over a small alphabet (dense matches, many >2-deep prefix chains,
    record-buffer overflow + rescan) and long patterns (halo up to 1021 bytes)."""
    rng = np.random.default_rng(1234)
    for trial, (npat, maxlen, alpha) in enumerate([(50, 6, 2), (300, 12, 3), (40, 1022, 2), (2000, 9, 4)]):
        pats = set()
        while len(pats) < npat:
            L = int(rng.integers(1, maxlen + 1))
            p = bytes(rng.integers(97, 97 + alpha, L, dtype=np.uint8))
            pats.add(p)
        pf = tmp_path / f"rp{trial}"
        pf.write_bytes(b"\n".join(sorted(pats, key=lambda x: rng.random())) + b"\n")
        table = PfacTable.from_file(str(pf), 256)
        data = rng.integers(97, 97 + alpha, 70001, dtype=np.uint8)
        if maxlen > 100:                      # plant a long pattern across a tile boundary
            longest = max(pats, key=len)
            data[TILE - 500:TILE - 500 + len(longest)] = np.frombuffer(longest, dtype=np.uint8)
        rec = gpu_records(table, data)
        pos, ids = oracle_pairs(str(pf), data)
        assert_same(table, rec, pos, ids)


def test_snort_scale_table_via_l2(resolve):
    """75 840 patterns / 542 732 states (BASELINE config 5's set): tables far beyond LDS."""
    table = PfacTable.from_file(resolve("bytefile/1000000byte"), 256)
    data = open(resolve("bytefile/1000000byte"), "rb").read()[:400000]
    with GpuMatcher(0, 1) as g:
        g.load_table(table)
        assert g.info()["variant"] == "tables_via_l2"
        rec = g.scan_bytes(data)
    pos, ids = oracle_pairs(resolve("bytefile/1000000byte"), data)
    assert_same(table, rec, pos, ids)


def test_full_size_properties(resolve):
    """BASELINE config-2 size (1 GiB resident in HBM): properties that need no 1 GiB oracle run.
    * tiled text: the input has period 402, so matches repeat with the period -> count and checksum
      follow from one oracle pass over a few periods;
    * checksum of the whole == sum of checksums of 4 shards with halo (linearity / shard invariance);
    * records are sorted by position."""
    import torch
    para = open(resolve("paragraph402"), "rb").read()
    N = 1 << 30
    table = PfacTable.from_file(resolve("experimentpattern"), 256)
    with GpuMatcher(0, 1) as g:
        g.load_table(table)
        if not os.environ.get("PFAC_FORCE_L2"):                 # (tuning knob used by variant sweeps)
            assert g.info()["variant"] == "tables_in_lds"
        buf = torch.empty(N + 1024, dtype=torch.uint8, device="cuda:0")
        g.fill_tiled(buf, N, para)
        head = buf[:4096].cpu().numpy()
        np.testing.assert_array_equal(head, tiled_bytes(4096, para))
        g.reserve(0, 0, N // 8)
        n = g.scan_resident(N, N, d_input=buf)
        # oracle on one period-aligned window: per-period match count, with the global end effect
        reps = 8
        win = tiled_bytes(402 * reps, para)
        pos, ids = oracle_pairs(resolve("experimentpattern"), win)
        inner = (pos >= 402) & (pos < 804)                    # one full period away from both ends
        per_period = int(inner.sum())
        full, tail = divmod(N, 402)
        assert tail >= 3                                      # no match of the last full period is cut by N
        # matches in the last partial period: oracle over the true tail bytes
        last = tiled_bytes(tail, para, phase=0)
        lpos, _ = oracle_pairs(resolve("experimentpattern"), last)
        assert n == per_period * full + lpos.size
        total_sum = g.checksum(n)
        # sortedness of a large prefix and of the tail, checked on the host
        rec_head = g.records_to_host(min(n, 1 << 20))
        assert (np.diff(rec_head["pos"].astype(np.int64)) >= 0).all()
        # linearity: 4 shards with halo, each into its own record region
        K, per = 4, N // 4
        s = 0
        cnt = 0
        for k in range(K):
            lo, hi = k * per, (k + 1) * per
            end = min(N, hi + table.halo)
            nk = g.scan_resident(hi - lo, end - lo, d_input=int(buf.data_ptr()) + lo)
            s = (s + g.checksum(nk, base=lo)) % (1 << 64)
            cnt += nk
        assert cnt == n and s == total_sum
        # and the checksum formula itself agrees with the oracle on a small prefix
        m = 1 << 20
        nm = g.scan_resident(m, m, d_input=buf)
        opos, oids = oracle_pairs(resolve("experimentpattern"), tiled_bytes(m, para))
        assert nm == opos.size and g.checksum(nm) == match_checksum(opos, oids)


def test_maximum_shard_4gib_positions_past_2_31(resolve, tmp_path):
    """The largest shard one scan takes: n_owned = 2^32 (C4/C5 shard size).  Record positions are unsigned 32-bit
    and must be exact past 2^31 and up to 2^32 - 1; count from the input's period, tail records against the oracle
    on the last bytes of the stream, text of those records, checksum linearity over two 2 GiB halves."""
    import torch
    para = open(resolve("paragraph402"), "rb").read()
    N = 1 << 32
    pat = resolve("experimentpattern")
    table = PfacTable.from_file(pat, 256)
    with GpuMatcher(0, 1) as g:
        g.load_table(table)
        buf = torch.empty(N + 1024, dtype=torch.uint8, device="cuda:0")
        g.fill_tiled(buf, N, para)
        g.reserve(0, 0, 340_000_000)
        n = g.scan_resident(N, N, d_input=buf)
        win = tiled_bytes(402 * 8, para)
        pos, _ = oracle_pairs(pat, win)
        per_period = int(((pos >= 402) & (pos < 804)).sum())
        full, tail = divmod(N, 402)
        assert tail >= 3
        lpos, _ = oracle_pairs(pat, tiled_bytes(tail, para))
        assert n == per_period * full + lpos.size
        total_sum = g.checksum(n)
        # the last records: oracle over the last W bytes of the stream (same end of input, so the same cut-offs)
        W = 6000
        wdata = tiled_bytes(W, para, phase=(N - W) % 402)
        opos, oids = oracle_pairs(pat, wdata)
        assert opos.size > 100
        rec = g.records_to_host(opos.size, first=n - opos.size)
        np.testing.assert_array_equal(rec["pos"].astype(np.int64), opos + (N - W))
        np.testing.assert_array_equal(table.idmap[rec["state"]], oids)
        assert int(rec["pos"][0]) > (1 << 31) and int(rec["pos"][-1]) >= N - 402
        out = tmp_path / "tail.txt"
        emit_records(str(out), rec, table.idmap)
        want = "".join("At position %4d, match pattern %d\n" % (p + N - W, i) for p, i in zip(opos, oids))
        assert out.read_text() == want
        # sortedness around the 2^31 boundary
        k = int(n // 2)
        mid = g.records_to_host(1 << 16, first=k - (1 << 15))
        assert (np.diff(mid["pos"].astype(np.int64)) > -1).all()
        # linearity: two 2 GiB halves (second with base 2^31)
        half = N // 2
        n0 = g.scan_resident(half, half + table.halo, d_input=buf)
        s0 = g.checksum(n0, base=0)
        n1 = g.scan_resident(half, half, d_input=int(buf.data_ptr()) + half)
        s1 = g.checksum(n1, base=half)
        assert n0 + n1 == n and (s0 + s1) % (1 << 64) == total_sum


def test_back_to_back_scans_of_varied_sizes(resolve):
    """Hundreds of scans of random lengths / offsets on two pipeline slots sharing a stream, counts checked against
    the oracle's per-phase match table of the periodic text: the control words of a scan are zeroed by the slot's
    previous scan (or by a fallback memset when the size jumps), and a stale word would derail the look-back."""
    import torch
    para = open(resolve("paragraph402"), "rb").read()
    pat = resolve("experimentpattern")
    table = PfacTable.from_file(pat, 256)
    o = Oracle(pat, 1, 1)
    pos, _ = o.scan_spec(tiled_bytes(402 * 4, para))
    per_phase = np.bincount(pos[(pos >= 402) & (pos < 804)] - 402, minlength=402)
    cum = np.concatenate([[0], np.cumsum(per_phase)])

    def expected(start, n_owned, n_avail):
        full, rem = divmod(n_owned, 402)
        idx = (start % 402 + np.arange(rem)) % 402
        cnt = int(full * cum[402] + per_phase[idx].sum())
        k = min(n_owned, 8)                                   # the last offsets may lose matches to the end of the range
        tail_lo = start + n_owned - k
        tp, _ = o.scan_spec(tiled_bytes(start + n_avail - tail_lo, para, phase=tail_lo % 402))
        return cnt - int(per_phase[(tail_lo % 402 + np.arange(k)) % 402].sum()) + int((tp < k).sum())

    N = 1 << 28
    rng = np.random.default_rng(11)
    buf = torch.empty(N + 4096, dtype=torch.uint8, device="cuda:0")
    with GpuMatcher(0, 2) as g:
        g.set_stream(1, g.stream_handle(0))
        g.load_table(table)
        g.fill_tiled(buf, N, para)
        g.reserve(0, 0, N // 8)
        g.reserve(1, 0, N // 8)
        inflight = []
        for k in range(400):
            kind = k % 4
            n_owned = [int(rng.integers(1, 1 << 14)), int(rng.integers(1 << 14, 1 << 22)), int(rng.integers(1 << 22, 1 << 27)),
                       int(rng.integers(1, 64)) * 4096 + int(rng.integers(-17, 18))][kind]
            start = int(rng.integers(0, (N - n_owned) // 16 + 1)) * 16
            n_avail = min(N - start, n_owned + int(rng.integers(0, 4)))
            g.scan_async(n_owned, n_avail, d_input=int(buf.data_ptr()) + start, slot=k & 1)
            inflight.append((k & 1, start, n_owned, n_avail))
            if len(inflight) == 2:
                sl, st, no, na = inflight.pop(0)
                cnt, over = g.scan_finish(sl)
                assert not over and cnt == expected(st, no, na), (k, st, no, na, cnt)
        for sl, st, no, na in inflight:
            cnt, over = g.scan_finish(sl)
            assert not over and cnt == expected(st, no, na)
    o.close()


def test_first_scan_of_fresh_contexts(resolve):
    """The very first scan of a new context uses control buffers that were allocated and zeroed a moment ago: the
    zeroing must be ordered before the scan on the slot's (non-blocking) stream.  Many fresh contexts, one scan each."""
    import torch
    para = open(resolve("paragraph402"), "rb").read()
    table = PfacTable.from_file(resolve("bytefile/10000byte"), 256)
    n = 1 << 22
    buf = torch.empty(n, dtype=torch.uint8, device="cuda:0")
    with GpuMatcher(0, 1) as g0:
        g0.fill_random(buf, n, 0x5048465046414331)
        g0.sync(0)
    junk = [torch.full((1 << 20,), 0x7F, dtype=torch.uint8, device="cuda:0") for _ in range(8)]   # dirty memory to hand back
    del junk
    for k in range(60):
        with GpuMatcher(0, 1) as g:
            g.load_table(table)
            g.reserve(0, 0, 1 << 12)
            assert g.scan_resident(n, n, d_input=buf) == 1, k


def test_random_fill_matches_cpu_twin_and_oracle(resolve):
    import torch
    n = 1 << 22
    table = PfacTable.from_file(resolve("bytefile/10000byte"), 256)
    with GpuMatcher(0, 1) as g:
        g.load_table(table)
        buf = torch.empty(n, dtype=torch.uint8, device="cuda:0")
        g.fill_random(buf, n, 0x5048465046414331)
        host = buf.cpu().numpy()
        np.testing.assert_array_equal(host, splitmix64_bytes(n, 0x5048465046414331))
        g.reserve(0, 0, 1 << 16)
        cnt = g.scan_resident(n, n, d_input=buf)
        rec = g.records_to_host(cnt)
    pos, ids = oracle_pairs(resolve("bytefile/10000byte"), host)
    assert_same(table, rec, pos, ids)


def test_compat_seam_dense_layout(resolve):
    """pfac_trace_table_compat fills the reference's dense input_size x max_pat_len array
    (master_kernel.cu:104-115,236) from tables in the reference's own r/HT/val form."""
    o = Oracle(resolve("xaa"), 1, 1)
    o.ffdm(1024, exact=True)                                  # the reference's exact FFDM layout
    L = o.L
    st = o.stats()
    trie = o.trie()
    nf = st["final"]
    table = PfacTable.from_reference_arrays(
        trie[nf + 1], np.ctypeslib.as_array(L.orc_phf_r(o.m, 0), (st["r_size"],)),
        np.ctypeslib.as_array(L.orc_phf_HT(o.m, 0), (st["ht_size"],)),
        np.ctypeslib.as_array(L.orc_phf_val(o.m, 0), (st["ht_size"],)), o.idmap(), 1024, st["state_num"], nf,
        st["ht_size"], L.orc_max_len(o.m))
    data = np.frombuffer(open(resolve("1M"), "rb").read()[:50000], dtype=np.uint8)
    dense = trace_table_compat(data, table, 0)
    pos, ids = o.scan_spec(data)
    exp = np.full_like(dense, 0xFFFFFFFF)
    inv = {int(v): k for k, v in enumerate(o.idmap())}
    fill = {}
    for p, i in zip(pos.tolist(), ids.tolist()):
        j = fill.get(p, 0)
        exp[p, j] = inv[i]
        fill[p] = j + 1
    np.testing.assert_array_equal(dense, exp)
    o.close()


def test_errors_are_loud(resolve):
    with GpuMatcher(0, 1) as g:
        with pytest.raises(PfacError):
            g.scan_async(16)                                   # scan before a table upload
    with pytest.raises(PfacError):
        GpuMatcher(99, 1)                                      # no such device
    import torch
    table = PfacTable.from_file(resolve("experimentpattern"), 256)
    with GpuMatcher(0, 1) as g:
        g.load_table(table)
        buf = torch.zeros(4096 + 64, dtype=torch.uint8, device="cuda:0")
        with pytest.raises(PfacError):
            g.scan_async((1 << 32) + 1, (1 << 32) + 1, d_input=buf)     # more than one scan's 2^32 positions
        with pytest.raises(PfacError):
            g.scan_async(100, 50, d_input=buf)                 # n_owned > n_avail
        with pytest.raises(PfacError):
            g.scan_async(64, 64, d_input=int(buf.data_ptr()) + 4)       # input not 16-byte aligned
        with pytest.raises(PfacError):
            g.scan_finish(0)                                   # nothing was scanned
        # a record buffer that is too small: the count is still exact, the caller learns it has to grow
        buf[:4096] = torch.from_numpy(tiled_bytes(4096, open(resolve("paragraph402"), "rb").read()).copy()).to("cuda:0")
        g.reserve(0, 0, 16)
        g.scan_async(4096, 4096, d_input=buf)
        n, over = g.scan_finish(0, allow_overflow=True)
        pos, _ = oracle_pairs(resolve("experimentpattern"), tiled_bytes(4096, open(resolve("paragraph402"), "rb").read()))
        assert over and n == pos.size > 16
        # ... and records of an overflowed scan, or records past the match count, are refused, not delivered as garbage
        with pytest.raises(PfacError):
            g.records_to_host(8)
        g.reserve(0, 0, 4096)
        assert g.scan_resident(4096, 4096, d_input=buf) == n
        assert g.records_to_host(n).size == n
        with pytest.raises(PfacError):
            g.records_to_host(n + 1)
        with pytest.raises(PfacError):
            g.records_to_host(2, first=n - 1)


def test_rccl_path_single_rank(resolve):
    """The N>1 plumbing on the real backend ("nccl" == RCCL) with one rank: table image broadcast into device
    memory, pfac_table_upload_device, scan into a torch-owned record buffer, count all-gather, ordered gather."""
    import torch
    import torch.distributed as dist
    from phfpfac_amd import dist as pdist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        table = PfacTable.from_file(resolve("xaa"), 256)
        blob, table2 = pdist.broadcast_table(table, dev, 0)
        torch.cuda.synchronize()
        data = np.frombuffer(open(resolve("1M"), "rb").read()[:150001], dtype=np.uint8)
        n_total = data.size
        lo, hi, end = pdist.shard_read_range(n_total, 0, 1, table2.halo)
        buf = torch.zeros(end - lo + 64, dtype=torch.uint8, device=dev)
        buf[: end - lo] = torch.from_numpy(data[lo:end].copy()).to(dev)
        rec_t = torch.empty(1 << 17, dtype=torch.int64, device=dev)       # torch-owned record buffer (either form fits)
        wide_t = torch.empty(1 << 17, dtype=torch.int64, device=dev)      # 8-byte records for the gather
        with GpuMatcher(0, 1) as g:
            g.load_table_device(blob, blob.numel(), 0, host_table=table2)
            g.scan_async(hi - lo, end - lo, d_input=buf, d_records=rec_t, capacity=rec_t.numel())
            n, over = g.scan_finish(0)
            assert not over
            rec_bytes, n_tiles, used = g.scan_format(0)                   # compact words in a heap + tile index
            assert rec_bytes == 4 and n_tiles == -(-(hi - lo) // 4096) and n <= used <= rec_t.numel()
            g.expand_records(n, wide_t, d_records=rec_t)                  # -> pfac_record, still on the device
            g.sync(0)
            # the compact gather: heap words + tile index as the kernel wrote them (what bench.py --gpus N moves)
            parts = pdist.gather_packed(g, dev, slot=0, d_records=rec_t, dst=0)
        counts = pdist.gather_counts(n, dev)
        assert counts == [n]
        gathered = pdist.gather_records(wide_t, n, counts, dst=0)
        got = pdist.split_gathered(gathered, counts, n_total, 1)
        assert len(parts) == 1 and parts[0]["n_matches"] == n and parts[0]["rec_bytes"] == 4
        assert parts[0]["words"].numel() < 8 * n
        got2 = pdist.packed_to_records(parts[0]["words"].cpu().numpy(), parts[0]["tix"].cpu().numpy(), 4, base=lo)
    finally:
        dist.destroy_process_group()
    pos, ids = oracle_pairs(resolve("xaa"), data)
    assert got.size == pos.size
    np.testing.assert_array_equal(got["pos"].astype(np.int64), pos)
    np.testing.assert_array_equal(table.idmap[got["state"]], ids)
    np.testing.assert_array_equal(got2["pos"].astype(np.int64), pos)
    np.testing.assert_array_equal(table.idmap[got2["state"]], ids)


def test_deeply_nested_patterns_every_offset_matches_hundreds(tmp_path):
    """Worst case for the record path: patterns a, aa, ..., a^300 on an all-'a' input -> up to 300 matches per
    start offset (the >2-matches re-walk, staging overflow + urgent look-back on every tile), millions of
    records, still globally ordered by (position, length) and bit-exact."""
    L = 300
    pf = tmp_path / "nest"
    order = np.random.default_rng(7).permutation(L) + 1
    pf.write_bytes(b"".join(b"a" * int(k) + b"\n" for k in order))
    table = PfacTable.from_file(str(pf), 256)
    data = np.full(3 * TILE + 777, ord("a"), dtype=np.uint8)
    data[TILE + 5] = ord("b")                                   # one break in the run
    rec = gpu_records(table, data)
    pos, ids = oracle_pairs(str(pf), data)
    assert rec.size == pos.size and rec.size > 10_000_000
    assert_same(table, rec, pos, ids)


# ---------------------------------------------------------------------------
# round 2: the rows the round-1 review found untested behind the HIP scan

def test_reference_seam_by_its_own_names(resolve, tmp_path):
    """GPU_Malloc_Memory / GPU_TraceTable / GPU_Free_memory with the reference's parameter lists
    (main.cc:35-37), compiled (libpfac_seam.so) and driven by a program with the shape of the reference's main():
    4 x streamnum pattern chunks, thread_data per chunk, dense input_size x max_pat_len results, the position-major
    merge and the fprintf loop of main.cc:304-350.  Its GPU_match_result.txt must be the golden file."""
    import subprocess
    exe = os.path.join(os.path.dirname(HERE), "phfpfac_amd", "bin", "gphf_seam")
    for case, streams in (("exp_x_1M_s1_w256", 1), ("all_x_1M_s2_w256", 2), ("xaa_x_1M_s1_w256", 1)):
        c = FP["cases"][case]
        subprocess.check_call([exe, resolve(c["pattern"]), str(streams), str(c["width"]), resolve(c["input"])], cwd=tmp_path)
        blob = (tmp_path / "GPU_match_result.txt").read_bytes()
        assert len(blob) == c["bytes"] and hashlib.md5(blob).hexdigest() == c["md5"], case
    lib = os.path.join(os.path.dirname(HERE), "phfpfac_amd", "lib", "libpfac_seam.so")
    syms = subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True).stdout
    for name in ("GPU_Malloc_Memory", "GPU_TraceTable", "GPU_Free_memory"):
        assert name in syms


@pytest.mark.parametrize("emit", ["host", "device"])
def test_gphf_parallel_emitter_behind_the_scan(emit, resolve, tmp_path):
    """SURVEY 8(f)1 behind the HIP scan: ONE 64 MiB chunk that yields ~5 M records (195 MB of text) goes through (host)
    the multi-threaded host emitter (PFAC_EMIT_THREADS=8, well above its 524 288-record threshold), printing straight
    from the compact device form (record heap + tile index), or (device, the default) the GPU-side emitter: text
    formatted on the device, returned through the ring of 64 MiB pinned buffers and pwrite()n by the writer pool;
    the file must equal the oracle's text byte for byte."""
    import subprocess
    exe = os.path.join(os.path.dirname(HERE), "phfpfac_amd", "bin", "gphf")
    para = open(resolve("paragraph402"), "rb").read()
    n = (64 << 20) - 4097
    big = tmp_path / "big64.txt"
    big.write_bytes(tiled_bytes(n + 1, para).tobytes())        # + the byte the CLI drops (main.cc:138)
    env = dict(os.environ, PFAC_EMIT_THREADS="8", PFAC_EMIT=emit, PFAC_CHUNK_MB="64")
    out = subprocess.run([exe, resolve("experimentpattern"), "1", "256", str(big)], cwd=tmp_path, env=env,
                         capture_output=True, text=True, check=True).stdout
    matches = int([l for l in out.splitlines() if l.startswith("4.Time for  emit")][0].split()[3])
    assert matches > 4 * (1 << 17)
    o = Oracle(resolve("experimentpattern"), 1, 1)
    exp = tmp_path / "expected.txt"
    cnt, _ = o.emit(tiled_bytes(n, para), str(exp), spec=True)
    o.close()
    assert cnt == matches
    got = hashlib.md5((tmp_path / "GPU_match_result.txt").read_bytes()).hexdigest()
    assert got == hashlib.md5(exp.read_bytes()).hexdigest()


def test_compact_form_to_text(resolve, tmp_path):
    """The compact device form itself (heap words + tile index, pfac_records_d2h_packed) printed by pfac_emit_packed
    == the sorted 8-byte records printed by pfac_emit_records, on a golden case; the heap holds every record once."""
    from phfpfac_amd import emit_packed
    c = FP["cases"]["all_x_1M_s1_w256"]
    table = PfacTable.from_file(resolve(c["pattern"]), c["width"])
    data = np.frombuffer(open(resolve(c["input"]), "rb").read()[:-1], dtype=np.uint8)
    with GpuMatcher(0, 1) as g:
        g.load_table(table)
        rec = g.scan_bytes(data)
        rec_bytes, n_tiles, used = g.scan_format(0)
        words, tix = g.packed_to_host(0)
    assert rec_bytes == 4 and words.dtype == np.uint32 and n_tiles == -(-data.size // 4096) and words.size == used >= rec.size
    cnt = (tix >> np.uint64(40)).astype(np.int64)
    first = (tix & np.uint64((1 << 40) - 1)).astype(np.int64)
    assert int(cnt.sum()) == rec.size == c["lines"]
    spans = sorted((int(f), int(f + n)) for f, n in zip(first, cnt) if n)
    assert all(a[1] <= b[0] for a, b in zip(spans, spans[1:])) and spans[-1][1] <= used     # tiles never overlap in the heap
    out = tmp_path / "packed.txt"
    for threads in (1, 5):
        assert emit_packed(str(out), words, tix, table.idmap, threads=threads) == c["bytes"]
        assert hashlib.md5(out.read_bytes()).hexdigest() == c["md5"]
    # an automaton with at most 16 final states gets 16-bit records (4 patterns here); PFAC_REC_BYTES=4 widens them
    c = FP["cases"]["exp_x_1M_s1_w256"]
    table = PfacTable.from_file(resolve(c["pattern"]), c["width"])
    data = np.frombuffer(open(resolve(c["input"]), "rb").read()[:-1], dtype=np.uint8)
    for knob, want_bytes, dt in ((None, 2, np.uint16), ("4", 4, np.uint32)):
        if knob:
            os.environ["PFAC_REC_BYTES"] = knob
        try:
            with GpuMatcher(0, 1) as g:
                g.load_table(table)
                rec = g.scan_bytes(data)
                rec_bytes, _, used = g.scan_format(0)
                words, tix = g.packed_to_host(0)
        finally:
            os.environ.pop("PFAC_REC_BYTES", None)
        assert rec_bytes == want_bytes and words.dtype == dt and words.size == used and rec.size == c["lines"]
        assert emit_packed(str(out), words, tix, table.idmap, threads=3) == c["bytes"]
        assert hashlib.md5(out.read_bytes()).hexdigest() == c["md5"]


ESCAPED_PATTERNS = (b"tab\\there\n" b"nl\\nin\\\\side\n" b"hex\\x41\\xfe\\x7\n" b"oct\\101\\7\\377\\0end\n"
                    b"odd\\8\\q\\'\\\"\n" b"plain\n" b"\\xff\\xfe\n" b"a\\nb\n" b"\\0\n")


def test_escaped_pattern_file_scanned_on_the_gpu(tmp_path):
    """SURVEY 8(f)4 behind the HIP scan: a pattern file with \\n \\xNN \\ooo escapes (patterns that contain newline and
    NUL bytes) built with pfac_table_build_file_escaped, scanned on the GPU, records == the oracle's escape-aware
    pipeline on the same bytes (create_table_reorder.c:131-185)."""
    pf = tmp_path / "esc"
    pf.write_bytes(ESCAPED_PATTERNS)
    table = PfacTable.from_file(str(pf), 256, escapes=True)
    assert table.n_patterns == 9
    rng = np.random.default_rng(5)
    raw = [b"tab\there", b"nl\nin\\side", b"hexA\xfe\x07", b"octA\x07\xff\x00end", b"odd\x008\\q'\"", b"plain", b"\xff\xfe",
           b"a\nb", b"\x00"]
    pieces = []
    for _ in range(30000):
        k = int(rng.integers(0, len(raw) + 3))
        pieces.append(raw[k] if k < len(raw) else bytes(rng.integers(0, 256, int(rng.integers(1, 9)), dtype=np.uint8)))
    data = np.frombuffer(b"".join(pieces), dtype=np.uint8)
    rec = gpu_records(table, data)
    o = Oracle(str(pf), 1, 1, escapes=True)
    pos, ids = o.scan_spec(data)
    o.close()
    assert pos.size > 20000 and set(np.unique(ids)) == set(range(1, 10))
    assert_same(table, rec, pos, ids)


def test_config3_four_slots_streamed_at_size(resolve):
    """BASELINE config 3 at size: 4 GiB of text in four 1 GiB chunks through FOUR pipeline slots on their own HIP
    streams -- H2D from host memory (hipMemcpyAsync), scan, count -- all four in flight together.  Count from the
    input's period; per-chunk checksums (linearity, base = chunk offset) against the same chunks scanned one at a
    time out of a device-filled buffer."""
    import torch
    para = open(resolve("paragraph402"), "rb").read()
    pat = resolve("experimentpattern")
    table = PfacTable.from_file(pat, 256)
    chunk, K = 1 << 30, 4
    N = chunk * K
    host = tiled_bytes(chunk + 402 + 64, para)                  # chunk k = host[phase_k : phase_k + n_avail]
    win = tiled_bytes(402 * 8, para)
    pos, _ = oracle_pairs(pat, win)
    per_period = int(((pos >= 402) & (pos < 804)).sum())
    full, tail = divmod(N, 402)
    lpos, _ = oracle_pairs(pat, tiled_bytes(tail, para))
    expect_total = per_period * full + lpos.size
    with GpuMatcher(0, K) as g:
        g.load_table(table)
        assert len({g.stream_handle(s) for s in range(K)}) == K      # four distinct streams
        counts, sums = [], []
        for s in range(K):
            g.reserve(s, chunk + table.halo, chunk // 8)
        for s in range(K):                                      # everything enqueued before anything is waited for
            lo = s * chunk
            n_avail = min(N, lo + chunk + table.halo) - lo
            ph = lo % 402
            g.h2d(host[ph: ph + n_avail], slot=s)
            g.scan_async(chunk, n_avail, slot=s)
        for s in range(K):
            n, over = g.scan_finish(s)
            assert not over
            counts.append(n)
            sums.append(g.checksum(n, base=s * chunk, slot=s))
        assert sum(counts) == expect_total
        # the same chunks, one at a time, from a buffer filled on the device
        buf = torch.empty(chunk + 4096, dtype=torch.uint8, device="cuda:0")
        for s in range(K):
            lo = s * chunk
            n_avail = min(N, lo + chunk + table.halo) - lo
            g.fill_tiled(buf, n_avail, para, phase=lo % 402)
            n = g.scan_resident(chunk, n_avail, d_input=buf)
            assert n == counts[s] and g.checksum(n, base=lo) == sums[s]


def test_protocol_timeout_is_reported_not_hidden(resolve, monkeypatch):
    """The error channel: with PFAC_FAULT=1 workgroup 1 never publishes the record bases of its second round, so its
    waves run into the (shortened) bounded wait; the flags travel through device memory to the host and
    pfac_scan_finish must return PFAC_E_INTERNAL -- and the next scan of the same context must be exact again."""
    import torch
    from phfpfac_amd._ffi import PFAC_E_INTERNAL
    para = open(resolve("paragraph402"), "rb").read()
    pat = resolve("experimentpattern")
    N = 64 << 20
    buf = torch.empty(N + 4096, dtype=torch.uint8, device="cuda:0")
    pos, _ = oracle_pairs(pat, tiled_bytes(1 << 20, para))
    monkeypatch.setenv("PFAC_FAULT", "1")
    monkeypatch.setenv("PFAC_SPIN_MAX", "20000")
    with GpuMatcher(0, 1) as g:
        g.load_table(PfacTable.from_file(pat, 256))            # knobs are read when a table is installed
        g.fill_tiled(buf, N, para)
        g.reserve(0, 0, N // 8)
        g.scan_async(N, N, d_input=buf)
        with pytest.raises(PfacError) as e:
            g.scan_finish(0)
        assert e.value.status == PFAC_E_INTERNAL and "timeout" in str(e.value)
        monkeypatch.delenv("PFAC_FAULT")
        monkeypatch.delenv("PFAC_SPIN_MAX")
        g.load_table(PfacTable.from_file(pat, 256))
        assert g.scan_resident(1 << 20, 1 << 20, d_input=buf) == pos.size


def test_two_slots_on_their_own_streams_large_scans(resolve):
    """Two persistent grids at once: two slots on DISTINCT non-blocking streams, 256 MiB each, enqueued back to back
    so their workgroups compete for the CUs (what gphf does with streamnum >= 2).  Counts must be exact every time."""
    import torch
    para = open(resolve("paragraph402"), "rb").read()
    pat = resolve("experimentpattern")
    n = 256 << 20
    pos, _ = oracle_pairs(pat, tiled_bytes(402 * 8, para))
    per_period = int(((pos >= 402) & (pos < 804)).sum())
    with GpuMatcher(0, 2) as g:
        g.load_table(PfacTable.from_file(pat, 256))
        assert g.stream_handle(0) != g.stream_handle(1)
        bufs = [torch.empty(n + 4096, dtype=torch.uint8, device="cuda:0") for _ in range(2)]
        for s in range(2):
            g.fill_tiled(bufs[s], n, para, phase=100 * s)
            g.reserve(s, 0, n // 8)
        ref = [g.scan_resident(n, n, d_input=bufs[s], slot=s) for s in range(2)]
        assert all(abs(r - per_period * n / 402) < per_period + 8 for r in ref)
        for _ in range(6):
            for s in range(2):
                g.scan_async(n, n, d_input=bufs[s], slot=s)
            assert [g.scan_finish(s)[0] for s in range(2)] == ref


def test_snort_scale_table_on_random_input(resolve):
    """BASELINE config 5's own input kind: the 75 840-pattern set (tables through L2, level-2 filter in its lookup
    form with the second-byte pre-filter) on splitmix64 random bytes, records == oracle; and with the filters off."""
    table = PfacTable.from_file(resolve("bytefile/1000000byte"), 256)
    n = 3 << 20
    data = splitmix64_bytes(n, 0x5048465046414331)
    pos, ids = oracle_pairs(resolve("bytefile/1000000byte"), data)
    assert pos.size > 50000
    rec = gpu_records(table, data)
    assert_same(table, rec, pos, ids)
    for knob in ("PFAC_NO_SECF", "PFAC_L2F"):
        os.environ[knob] = "1" if knob == "PFAC_NO_SECF" else "0"
        try:
            assert_same(table, gpu_records(table, data), pos, ids)
        finally:
            del os.environ[knob]


def test_staging_layout_follows_the_match_density(tmp_path):
    """Tables in LDS have two sparse staging layouts: three buffers (records leave at the top of a round; smaller
    buffers) and two.  A context starts with three, falls back to two when more than 1/16 of a scan's tiles held more
    records than the small buffers take (those tiles are walked twice), and returns when the input thins out.  Inputs
    on either side of the three-buffer capacity (384 records per 4 KiB tile) and well past the two-buffer one, scanned
    back to back by ONE context: every scan's records must equal the oracle's."""
    pat = tmp_path / "p"
    pat.write_bytes(b"a\nab\nabc\n")
    table = PfacTable.from_file(str(pat), 256)
    rng = np.random.default_rng(11)
    n = 3 * 1024 * 1024 + 77
    with GpuMatcher(0, 1) as g:
        g.load_table(table)
        assert g.info()["variant"] == "tables_in_lds"
        seen = []
        assert g.info()["staging_buffers"] == 3
        for density in (0.02, 0.08, 0.08, 0.02, 0.02, 0.3, 0.08, 0.02):
            u = rng.random(n)
            data = np.where(u < density, ord("a"), np.where(u < density + 0.3, ord("b"), ord("c"))).astype(np.uint8).tobytes()
            rec = g.scan_bytes(data)
            seen.append(g.info()["staging_buffers"])            # the layout the NEXT scan gets
            pos, ids = oracle_pairs(str(pat), data)
            assert_same(table, rec, pos, ids)
        assert seen == [3, 2, 2, 3, 3, 1, 2, 3], seen


def test_malformed_table_image_is_refused_at_upload(resolve):
    """The fused walk indexes its slot array unchecked (the array is padded to every index a well-formed image can
    produce), so an image whose next states, root states or row displacements point outside the tables must not get
    onto the device: the upload checks them all (pfac_repack_kernel) and fails with PFAC_E_ARG."""
    table = PfacTable.from_file(resolve("xaa"), 256)
    blob = np.array(table.blob(), dtype=np.int32)
    max_row, ht = int(blob[8]), int(blob[9])
    base_r, base_ht = 16 + 256, 16 + 256 + max_row
    owned = np.flatnonzero(blob[base_ht:base_ht + ht] >= 0)
    bad_images = []
    b = blob.copy(); b[base_ht + ht + owned[len(owned) // 2]] = int(blob[6]) + 5; bad_images.append(b)     # a next state that is no state
    b = blob.copy(); b[16 + ord("a")] = 1 << 30; bad_images.append(b)                                       # a root edge to nowhere
    b = blob.copy(); b[base_r + 1] = ht + 7; bad_images.append(b)                                           # a row displaced past the table
    b = blob.copy(); b[base_r + 2] = -(1 << 20); bad_images.append(b)                                       # ... or far before it
    with GpuMatcher(0, 1) as g:
        for b in bad_images:
            with pytest.raises(PfacError, match="outside the tables"):
                g.load_table(b)
        g.load_table(blob)                                                                                   # the image itself is fine
        data = open(resolve("1M"), "rb").read()[:50001]
        rec = g.scan_bytes(data)
    pos, ids = oracle_pairs(resolve("xaa"), data)
    assert_same(table, rec, pos, ids)


@pytest.mark.parametrize("force_l2", [False, True])
@pytest.mark.parametrize("n_first,n_second", [(120, 40), (30, 256), (100, 256), (255, 3), (256, 5)])
def test_dense_depth1_rows_by_column(n_first, n_second, force_l2, tmp_path, monkeypatch):
    """The dense rows of the depth-1 states sit in LDS with only the columns of bytes that are some pattern's second
    byte (+ a "no edge" column): many first bytes x few second bytes (120 x 40), every byte a second byte (no spare
    column: 30 x 256), too much for LDS (100 x 256: the walk hashes its second byte), 255 depth-1 states, and one too
    many for a row index (256) -- tables in LDS and through L2, records == oracle."""
    if force_l2:
        monkeypatch.setenv("PFAC_FORCE_L2", "1")
    rng = np.random.default_rng(n_first * 1000 + n_second)
    firsts = rng.permutation(256)[:n_first].astype(np.uint8)
    firsts = firsts[firsts != 10][: n_first] if n_first < 256 else np.array([b for b in range(256) if b != 10], dtype=np.uint8)
    seconds = rng.permutation(np.array([b for b in range(256) if b != 10], dtype=np.uint8))[: min(n_second, 255)]
    pats = set()
    for f in firsts:                                       # every first byte with a few of the second bytes, some going deeper
        for sb in rng.choice(seconds, size=min(6, len(seconds)), replace=False):
            tail = bytes(int(x) for x in rng.choice(seconds, size=int(rng.integers(0, 4))))
            pats.add(bytes([int(f), int(sb)]) + tail)
    for sb in seconds:                                     # ... and every second byte used at least once
        pats.add(bytes([int(firsts[int(rng.integers(0, len(firsts)))]), int(sb)]))
    pf = tmp_path / "p"
    pf.write_bytes(b"\n".join(sorted(pats)) + b"\n")
    table = PfacTable.from_file(str(pf), 256)
    alphabet = np.union1d(firsts, seconds)
    data = alphabet[rng.integers(0, alphabet.size, 150_001)].astype(np.uint8)
    plist = sorted(pats)
    for at in rng.integers(0, data.size - 8, 4000):       # ... with patterns planted all over it
        pt = plist[int(rng.integers(0, len(plist)))]
        data[at:at + len(pt)] = np.frombuffer(pt, dtype=np.uint8)
    rec = gpu_records(table, data)
    pos, ids = oracle_pairs(str(pf), data)
    assert pos.size > 3000
    assert_same(table, rec, pos, ids)


# ---------------------------------------------------------------------------
# round 3: the GPU-side text emitter (main.cc:335-350 on the device)

@pytest.mark.parametrize("case", sorted(FP["cases"]))
def test_gpu_text_emitter_goldens(case, resolve):
    """pfac_emit_text_device: the lines of GPU_match_result.txt formatted ON THE GPU from the compact record heap + tile
    index (sizes per 64 tiles, prefix sum, format), copied back as finished text: md5-identical to the reference's file
    for every golden case."""
    c = FP["cases"][case]
    table = PfacTable.from_file(resolve(c["pattern"]), c["width"])
    data = np.frombuffer(open(resolve(c["input"]), "rb").read()[:-1], dtype=np.uint8)     # main.cc:138
    with GpuMatcher(0, 1) as g:
        g.load_table(table)
        g.reserve(0, max(data.size, 1), max(data.size // 2, 4096))
        if data.size:
            g.h2d(data, 0)
        n = g.scan_resident(data.size, data.size)
        nbytes = g.emit_text_device(0)
        text = g.text_to_host(nbytes)
    assert n == c["lines"] and nbytes == len(text) == c["bytes"]
    assert hashlib.md5(text).hexdigest() == c["md5"]


@pytest.mark.parametrize("base,knob", [(0, {}), (999_999_000, {}), (5 << 32, {}), (10**17, {}), (9_990, {"PFAC_WIDE": "1"}),
                                       (99_999_999_000, {"PFAC_DENSE": "1"})])
def test_gpu_text_emitter_digit_boundaries_and_record_forms(base, knob, resolve, tmp_path, monkeypatch):
    """Line length changes with the digit count of the position (%4d pads to 4, grows to 18 digits) and of the pattern
    id: bases that put 10^k boundaries inside the scanned range, the 8-byte record form (PFAC_WIDE), dense staging;
    sparse tiles (chunks of a few lines) and dense ones (hundreds of lines per tile).  Must equal the host emitter's
    bytes, which the goldens pin."""
    for k, v in knob.items():
        monkeypatch.setenv(k, v)
    para = open(resolve("paragraph402"), "rb").read()
    data = tiled_bytes(300_001, para)
    data[100_000:101_000] = 0                                   # a stretch without matches: empty tiles, tiny chunks
    table = PfacTable.from_file(resolve("xaa"), 256)
    with GpuMatcher(0, 1) as g:
        g.load_table(table)
        rec = g.scan_bytes(data)
        nbytes = g.emit_text_device(base)
        text = g.text_to_host(nbytes)
        assert g.text_to_host(min(nbytes, 1000), first=max(nbytes - 1000, 0)) == text[-1000:]
        with pytest.raises(PfacError):
            g.text_to_host(nbytes + 1)
    out = tmp_path / "host.txt"
    assert emit_records(str(out), rec, table.idmap, base=base) == nbytes
    assert out.read_bytes() == text


def test_gpu_text_emitter_at_size(resolve):
    """1 GiB x experimentpattern (80.1 M lines, 3.1 GB of text): byte count from the closed form of the periodic input,
    head and tail of the text against the host emitter, a checksum of the whole text against the multi-threaded host
    emitter's file."""
    import torch
    para = open(resolve("paragraph402"), "rb").read()
    N = 1 << 30
    table = PfacTable.from_file(resolve("experimentpattern"), 256)
    with GpuMatcher(0, 1) as g:
        g.load_table(table)
        buf = torch.empty(N + 1024, dtype=torch.uint8, device="cuda:0")
        g.fill_tiled(buf, N, para)
        g.reserve(0, 0, N // 8)
        n = g.scan_resident(N, N, d_input=buf)
        nbytes = g.emit_text_device(0)
        # expected size: every record is one line; digits(pos) from the record positions themselves (host, vectorised)
        head = g.text_to_host(1 << 20)
        tail = g.text_to_host(1 << 20, first=nbytes - (1 << 20))
        k = head.count(b"\n")
        rec_head = g.records_to_host(k + 1)
        rec_tail_n = tail.count(b"\n") - 1                        # (the slice begins inside a line: that one is not compared)
        rec_tail = g.records_to_host(rec_tail_n, first=n - rec_tail_n)
    want_head = "".join("At position %4d, match pattern %d\n" % (p, i) for p, i in zip(rec_head["pos"][:k], table.idmap[rec_head["state"][:k]]))
    assert head.decode().startswith(want_head) and len(want_head) > (1 << 20) - 64
    want_tail = "".join("At position %4d, match pattern %d\n" % (p, i) for p, i in zip(rec_tail["pos"], table.idmap[rec_tail["state"]]))
    assert tail.decode().endswith(want_tail) and tail.endswith(b"\n") and len(want_tail) > (1 << 20) - 64
    # the total, exactly: the input has period 402, so the match set is {off + 402 q} for the (offset, pattern) pairs of one
    # period, cut where a match would cross N; a line is 12 + max(4, digits(pos)) + 16 + digits(id) + 1 bytes
    win = tiled_bytes(402 * 8, para)
    pos, ids = oracle_pairs(resolve("experimentpattern"), win)
    sel = (pos >= 402) & (pos < 804)
    plen = {i + 1: len(l) for i, l in enumerate(open(resolve("experimentpattern"), "rb").read().split(b"\n")[:-1])}
    total = lines = 0
    for off, pid in zip(pos[sel] - 402, ids[sel]):
        q_max = (N - plen[int(pid)] - int(off)) // 402          # last q with off + 402 q + len <= N
        for d in range(4, 11):
            lo, hi = (0 if d == 4 else 10 ** (d - 1)), 10 ** d     # positions with this many printed digits
            q_lo = max(0, -(-(lo - int(off)) // 402))
            q_hi = min(q_max, (hi - 1 - int(off)) // 402)
            c = max(0, q_hi - q_lo + 1)
            lines += c
            total += c * (12 + d + 16 + len(str(int(pid))) + 1)
    assert lines == n and total == nbytes


def test_knobs_are_opt_in(resolve):
    """A production process cannot have its scans altered by a stray environment variable: the kernel's tuning and test
    knobs are read only when PFAC_ENABLE_KNOBS=1 (this test suite sets it in conftest.py).  Without it PFAC_FORCE_L2 and
    PFAC_FAULT are ignored; with it they act."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); from phfpfac_amd import GpuMatcher, PfacTable\n"
            "g = GpuMatcher(0, 1); g.load_table(PfacTable.from_file(%r, 256)); print(g.info()['variant'])"
            % (os.path.dirname(HERE), resolve("experimentpattern")))
    base = {k: v for k, v in os.environ.items() if not k.startswith("PFAC_")}
    off = subprocess.run([sys.executable, "-c", code], env=dict(base, PFAC_FORCE_L2="1", PFAC_FAULT="1"), capture_output=True, text=True, check=True)
    on = subprocess.run([sys.executable, "-c", code], env=dict(base, PFAC_FORCE_L2="1", PFAC_ENABLE_KNOBS="1"), capture_output=True, text=True, check=True)
    assert off.stdout.split()[-1] == "tables_in_lds" and on.stdout.split()[-1] == "tables_via_l2"


@pytest.mark.parametrize("ingest", ["mmap", "pread"])
def test_gphf_edge_sizes(ingest, resolve, tmp_path):
    """The CLI on degenerate and boundary-sized inputs, both ingest paths: an empty file, one and two bytes (N = size - 1,
    main.cc:138), sizes around a page, around one chunk and around a registration piece boundary (PFAC_CHUNK_MB=1: a piece
    is 256 chunks), a file whose last chunk is one byte long."""
    import subprocess
    exe = os.path.join(os.path.dirname(HERE), "phfpfac_amd", "bin", "gphf")
    para = open(resolve("paragraph402"), "rb").read()
    o = Oracle(resolve("xaa"), 1, 1)
    env = dict(os.environ, PFAC_CHUNK_MB="1", PFAC_INGEST=ingest, PFAC_READ_THREADS="3")
    for size in (0, 1, 2, 5, 4095, 4096, 4097, (1 << 20), (1 << 20) + 1, (1 << 20) + 2, 3 * (1 << 20) + 17):
        f = tmp_path / f"in_{size}"
        f.write_bytes(tiled_bytes(size, para).tobytes())
        r = subprocess.run([exe, resolve("xaa"), "2", "256", str(f)], cwd=tmp_path, env=env, capture_output=True, text=True)
        assert r.returncode == 0, (size, r.stderr[-500:])
        exp = tmp_path / "expected.txt"
        o.emit(tiled_bytes(max(size - 1, 0), para), str(exp), spec=True)
        assert (tmp_path / "GPU_match_result.txt").read_bytes() == exp.read_bytes(), size
    o.close()
