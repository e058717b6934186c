"""GPU soak: many back-to-back scans of random lengths / offsets over resident inputs, two slots in flight, every
count checked against a closed form (text input has period 402); scans below 1 MiB also have their expanded records
compared with the CPU oracle record for record (heap placement, tile index, chunk boundaries).
Looks for rare protocol failures (timeouts surface as PFAC_E_INTERNAL), count drift and misplaced records.
usage: soak.py [seconds] [seed] [own] [dict]      ("own": the two slots keep their own streams -> two grids at once;
"dict": the 7 989-word dictionary instead of experimentpattern -- tables via L2, dense staging, four walks per lane)"""
import os
os.environ.setdefault("PFAC_ENABLE_KNOBS", "1")     # tuning / test knobs of libpfac_hip.so are opt-in
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from phfpfac_amd import GpuMatcher, PfacTable
from phfpfac_amd.matcher import tiled_bytes
from orc import Oracle

DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "data")
para = open(os.path.join(DATA, "paragraph402"), "rb").read()
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)

N = 1 << 30
buf = torch.empty(N + 4096, dtype=torch.uint8, device="cuda:0")
pat = os.path.join(DATA, "experimentpattern")
if "dict" in sys.argv:
    import tempfile
    pat = os.path.join(tempfile.mkdtemp(), "all.pat")
    open(pat, "wb").write(b"".join(open(os.path.join(DATA, p), "rb").read() for p in ("xaa", "xab", "xac", "xad")))
table = PfacTable.from_file(pat, 256)
HALO = max(table.max_pat_len, 8)

# matches starting at each phase of the period (walks never cross more than max_pat_len bytes)
o = Oracle(pat, 1, 1)
win = tiled_bytes(402 * 4, para)
pos, _ = o.scan_spec(win)
per_phase = np.bincount(pos[(pos >= 402) & (pos < 804)] - 402, minlength=402)
cum = np.concatenate([[0], np.cumsum(per_phase)])


def expected(start, n_owned, n_avail):
    """matches with start offset in [start, start+n_owned) of the infinite periodic text, cut at start+n_avail"""
    full, rem = divmod(n_owned, 402)
    ph = start % 402
    idx = (ph + np.arange(rem)) % 402
    cnt = int(full * cum[402] + per_phase[idx].sum())
    # cut-off at the end of the readable range: recount the last few offsets exactly
    k = min(n_owned, HALO)
    tail_lo = start + n_owned - k
    data = tiled_bytes(start + n_avail - tail_lo, para, phase=tail_lo % 402)
    tp, _ = o.scan_spec(data)
    exact_tail = int((tp < k).sum())
    approx_tail = int(per_phase[(tail_lo % 402 + np.arange(k)) % 402].sum())
    return cnt - approx_tail + exact_tail


with GpuMatcher(0, 2) as g:
    if "own" not in sys.argv:
        g.set_stream(1, g.stream_handle(0))
    g.load_table(table)
    g.fill_tiled(buf, N, para)
    g.reserve(0, 0, N // 2 if "dict" in sys.argv else N // 8)
    g.reserve(1, 0, N // 2 if "dict" in sys.argv else N // 8)
    t0 = time.time()
    last_note = t0
    scans = 0
    checked = 0
    inflight = []
    while time.time() - t0 < seconds:
        kind = rng.integers(0, 4)
        if kind == 0:
            n_owned = int(rng.integers(1, 1 << 16))
        elif kind == 1:
            n_owned = int(rng.integers(1 << 16, 1 << 24))
        elif kind == 2:
            n_owned = int(rng.integers(1 << 24, 1 << 29))
        else:
            n_owned = int(rng.integers(1, 64)) * 4096 + int(rng.integers(-17, 18))
        start = int(rng.integers(0, (N - n_owned) // 16 + 1)) * 16
        halo = int(rng.integers(0, HALO))
        n_avail = min(N - start, n_owned + halo)
        sl = scans & 1
        g.scan_async(n_owned, n_avail, d_input=int(buf.data_ptr()) + start, slot=sl)
        inflight.append((sl, start, n_owned, n_avail))
        if len(inflight) == 2:
            s_, st, no, na = inflight.pop(0)
            cnt, over = g.scan_finish(s_)
            want = expected(st, no, na)
            if cnt != want or over:
                raise SystemExit(f"MISMATCH scan {scans}: start {st} n_owned {no} n_avail {na}: got {cnt} want {want} overflow {over}")
            if no < (1 << 20):
                rec = g.records_to_host(cnt, slot=s_)
                opos, oids = o.scan_spec(tiled_bytes(na, para, phase=st % 402))
                keep = opos < no
                if not (np.array_equal(rec["pos"].astype(np.int64), opos[keep]) and np.array_equal(table.idmap[rec["state"]], oids[keep])):
                    raise SystemExit(f"RECORD MISMATCH scan {scans}: start {st} n_owned {no} n_avail {na}")
                checked += 1
        scans += 1
        if scans % 4096 == 0 and time.time() - last_note > 30:            # (a silent GPU run is taken to be hung)
            last_note = time.time()
            print(f"  ... {scans} scans, {time.time() - t0:.0f} s", flush=True)
    for s_, st, no, na in inflight:
        cnt, over = g.scan_finish(s_)
        assert cnt == expected(st, no, na) and not over
    print(f"soak ok: {scans} scans in {time.time() - t0:.1f} s, all counts exact, {checked} scans compared record for record")
o.close()
