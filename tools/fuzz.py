"""GPU differential fuzz: random pattern sets x random inputs, random kernel knobs, every record compared with the CPU oracle.
usage: fuzz.py [seconds] [seed]"""
import os, sys, tempfile, time
os.environ.setdefault("PFAC_ENABLE_KNOBS", "1")     # tuning / test knobs of libpfac_hip.so are opt-in
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from phfpfac_amd import GpuMatcher, PfacTable
from orc import Oracle

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
tmp = tempfile.mkdtemp()
KNOBS = [{}, {"PFAC_FORCE_L2": "1"}, {"PFAC_FORCE_L2": "1", "PFAC_DENSE": "1"}, {"PFAC_DENSE": "1"}, {"PFAC_LAG": "1"},
         {"PFAC_LAG": "2"}, {"PFAC_FORCE_L2": "1", "PFAC_NO_FUSE": "1"}, {"PFAC_FORCE_L2": "1", "PFAC_NO_D1": "1"},
         {"PFAC_REC_BYTES": "4"}, {"PFAC_WIDE": "1"}, {"PFAC_FORCE_L2": "1", "PFAC_NO_NW4": "1", "PFAC_DENSE": "1"},
         {"PFAC_L2F": "0"}, {"PFAC_L2F": "2"}, {"PFAC_L2F": "3"}, {"PFAC_L2F": "3", "PFAC_FORCE_L2": "1"}, {"PFAC_NO_SECF": "1", "PFAC_FORCE_L2": "1"}, {"PFAC_NWB": "4"},
         {"PFAC_FORCE_L2": "1", "PFAC_DENSE": "1", "PFAC_D2_LOGCAP": "64"}, {"PFAC_FORCE_L2": "1", "PFAC_DENSE": "1", "PFAC_NO_DENSE2": "1"},
         {"PFAC_FORCE_L2": "1", "PFAC_DENSE": "1", "PFAC_NWB": "5"}]
ALL = sorted({k for d in KNOBS for k in d})
t0 = t_last = time.time(); cases = 0; recs = 0
while time.time() - t0 < seconds:
    alpha = int(rng.choice([2, 3, 4, 8, 26, 60, 200]))
    symbols = rng.permutation(np.array([b for b in range(256) if b != 10], dtype=np.uint8))[:alpha]
    npat = int(rng.choice([1, 3, 20, 200, 1500]))
    maxlen = int(rng.choice([1, 2, 4, 8, 14, 40])) if rng.random() < 0.9 else int(rng.integers(100, 1000))
    pats = set()
    for _ in range(npat * 3):
        if len(pats) >= npat: break
        L = int(rng.integers(1, maxlen + 1))
        pats.add(bytes(symbols[rng.integers(0, alpha, L)]))
    pf = os.path.join(tmp, "p%d" % cases)
    open(pf, "wb").write(b"\n".join(sorted(pats, key=lambda x: rng.random())) + b"\n")
    width = int(rng.choice([64, 256, 256, 1024]))
    knobs = KNOBS[int(rng.integers(0, len(KNOBS)))]
    for k in ALL: os.environ.pop(k, None)
    os.environ.update(knobs)
    table = PfacTable.from_file(pf, width)
    n = int(rng.choice([1, 17, 4095, 4097, 70001, 300007, 300007, 2_000_003, 9_000_001], p=[.1, .1, .1, .1, .2, .2, .1, .07, .03]))
    data = symbols[rng.integers(0, alpha, n)]
    plist = sorted(pats)
    for at in rng.integers(0, max(n - 1, 1), max(n // 50, 1)):
        pt = np.frombuffer(plist[int(rng.integers(0, len(plist)))], dtype=np.uint8)
        m = min(len(pt), n - int(at)); data[int(at):int(at) + m] = pt[:m]
    n_owned = n if rng.random() < 0.7 else int(rng.integers(0, n + 1))
    with GpuMatcher(0, 1) as g:
        g.load_table(table)
        for rep in range(2):                               # twice: the staging layout may adapt after the first scan
            rec = g.scan_bytes(data, n_owned)
            o = Oracle(pf, 1, 1); pos, ids = o.scan_spec(data, None); o.close()
            own = pos < n_owned                            # (the rest of the buffer is halo: read, not scanned from)
            pos, ids = pos[own], ids[own]
            if not (rec.size == pos.size and np.array_equal(rec["pos"].astype(np.int64), pos) and np.array_equal(table.idmap[rec["state"]], ids)):
                raise SystemExit(f"MISMATCH case {cases} rep {rep}: seed {seed} alpha {alpha} npat {len(pats)} maxlen {maxlen} width {width} knobs {knobs} n {n} n_owned {n_owned}: got {rec.size} want {pos.size} (pattern file {pf})")
            recs += int(pos.size)
            if rep == 1 and pos.size < 400000:             # the GPU-side text emitter against lines formatted here
                base = int(rng.choice([0, 999_999_990, 3 << 32]))
                text = g.text_to_host(g.emit_text_device(base))
                want = "".join("At position %4d, match pattern %d\n" % (p + base, i) for p, i in zip(pos.tolist(), ids.tolist())).encode()
                if text != want:
                    raise SystemExit(f"TEXT MISMATCH case {cases}: seed {seed} knobs {knobs} n {n} n_owned {n_owned} base {base}: {len(text)} bytes, want {len(want)} (pattern file {pf})")
    os.remove(pf); cases += 1
    if time.time() - t_last > 30:
        t_last = time.time(); print(f"  ... {cases} cases, {recs} records compared, {t_last - t0:.0f} s", flush=True)
print(f"fuzz ok: {cases} cases in {time.time() - t0:.0f} s (seed {seed}), {recs} records compared")
