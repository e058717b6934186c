"""GPU diagnostic: per-launch kernel times of one scan workload (1 GiB of the reference text, no parity check --
usable with ablation builds whose counts are wrong).  usage: series.py <pattern fixture> [text|rand]   (PFAC_HIP_LIB selects the build)"""
import os, sys
os.environ.setdefault("PFAC_ENABLE_KNOBS", "1")     # tuning / test knobs of libpfac_hip.so are opt-in
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from phfpfac_amd import GpuMatcher, PfacTable
DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "data")
para = open(os.path.join(DATA, "paragraph402"), "rb").read()
N = 1 << 30
buf = torch.empty(N + 4096, dtype=torch.uint8, device="cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "experimentpattern"
path = os.path.join(DATA, name)
if "+" in name:
    import tempfile
    path = os.path.join(tempfile.mkdtemp(), "all.pat")
    open(path, "wb").write(b"".join(open(os.path.join(DATA, p), "rb").read() for p in name.split("+")))
if name.endswith(".gz"):
    import gzip, tempfile
    path = os.path.join(tempfile.mkdtemp(), name[:-3]); open(path, "wb").write(gzip.open(os.path.join(DATA, name), "rb").read())
t = PfacTable.from_bytes(b"\x01\x02\n", 256) if name == "nomatch1" else PfacTable.from_file(path, 256)
with GpuMatcher(0, 1) as g:
    g.load_table(t)
    if len(sys.argv) > 2 and sys.argv[2] == "rand": g.fill_random(buf, N, 0x5048465046414331)
    else: g.fill_tiled(buf, N, para)
    g.reserve(0, 0, N // 2 if "+" in name else N // 8)
    ms = []
    for _ in range(int(os.environ.get("N_LAUNCH", "40"))):
        g.scan_async(N, N, d_input=buf)
        try: g.scan_finish(0, allow_overflow=True)
        except Exception as e: print("finish:", e)
        ms.append(g.elapsed_ms(0))
half = np.array(ms[len(ms) // 2:])
tail = np.array(ms[-16:])
print(f"{os.path.basename(os.environ.get('PFAC_HIP_LIB', 'product'))} {name} {sys.argv[2] if len(sys.argv) > 2 else 'text'}: last16 mean {tail.mean():.4f} ms = {N / tail.mean() / 1e6:.0f} GB/s, min {tail.min():.4f} ms = {N / tail.min() / 1e6:.0f} GB/s; second half of {len(ms)} launches: mean {half.mean():.4f} ms = {N / half.mean() / 1e6:.0f} GB/s")
