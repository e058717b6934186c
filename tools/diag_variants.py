"""GPU diagnostic: kernel GB/s for small synthetic pattern sets that isolate root mode / halo / tables."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from phfpfac_amd import GpuMatcher, PfacTable
DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "data")
para = open(os.path.join(DATA, "paragraph402"), "rb").read()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 28
sets = {
    "exp(ROOT1,halo16)": b"aaaa\naa\na\naaa\n",
    "root1_long(ROOT1,halo32)": b"a\n" + b"a" * 22 + b"\n",
    "root2_short(ROOT0,halo16)": b"a\nzq\n",
    "root2_long(ROOT0,halo32)": b"a\n" + b"z" * 22 + b"\n",
    "nomatch_short(ROOT0,halo16)": b"\x01\x02\n\x03\x04\n",
    "nomatch_long(ROOT0,halo32)": b"\x01\x02\n" + b"\x03" * 22 + b"\n",
    "nomatch1(ROOT1,halo16)": b"\x01\x02\n",
}
for f in ("bytefile_10000byte",):
    sets[f] = open(os.path.join(DATA, f), "rb").read()
buf = torch.empty(N + 4096, dtype=torch.uint8, device="cuda:0")
for name, pats in sets.items():
    t = PfacTable.from_bytes(pats, 256)
    with GpuMatcher(0, 1) as g:
        g.load_table(t)
        g.fill_tiled(buf, N, para)
        g.reserve(0, 0, N // 8)
        n = g.scan_resident(N, N, d_input=buf)
        ms = []
        for _ in range(5):
            g.scan_async(N, N, d_input=buf); g.scan_finish(0); ms.append(g.elapsed_ms(0))
        print(f"{name:32s} matches {n:10d}  kernel {min(ms):8.3f} ms  {N/min(ms)/1e6:9.1f} GB/s  {g.info()}", flush=True)
