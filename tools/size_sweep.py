"""GPU diagnostic: kernel time of one scan workload against the shard size -> the fixed cost of a launch (prologue,
first tiles, tail, exit protocol) and the marginal rate.  usage: size_sweep.py [pattern fixture]   (PFAC_HIP_LIB selects the build)"""
import os, sys
os.environ.setdefault("PFAC_ENABLE_KNOBS", "1")     # tuning / test knobs of libpfac_hip.so are opt-in
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from phfpfac_amd import GpuMatcher, PfacTable
DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "data")
para = open(os.path.join(DATA, "paragraph402"), "rb").read()
name = sys.argv[1] if len(sys.argv) > 1 else "experimentpattern"
t = PfacTable.from_bytes(b"\x01\x02\n", 256) if name == "nomatch1" else PfacTable.from_file(os.path.join(DATA, name), 256)
NMAX = 4 << 30
buf = torch.empty(NMAX + 4096, dtype=torch.uint8, device="cuda:0")
with GpuMatcher(0, 1) as g:
    g.load_table(t)
    g.fill_tiled(buf, NMAX, para)
    g.reserve(0, 0, NMAX // 8)
    for _ in range(150):                                   # clocks
        g.scan_async(1 << 30, 1 << 30, d_input=buf); g.scan_finish(0)
    xs, ys = [], []
    for n in (64 << 20, 128 << 20, 256 << 20, 512 << 20, 1 << 30, 2 << 30, 4 << 30):
        ms = []
        for _ in range(24):
            g.scan_async(n, n, d_input=buf); g.scan_finish(0); ms.append(g.elapsed_ms(0))
        m = float(np.mean(ms[8:]))
        xs.append(n / 2 ** 30); ys.append(m)
        print(f"{n >> 20:5d} MiB: {m * 1000:8.1f} us = {n / m / 1e6:6.0f} GB/s")
    a, b = np.polyfit(xs, ys, 1)
    print(f"{os.path.basename(os.environ.get('PFAC_HIP_LIB', 'product'))} {name}: fit {a * 1000:.1f} us per GiB ({2 ** 30 / a / 1e6:.0f} GB/s marginal) + {b * 1000:.1f} us per launch")
