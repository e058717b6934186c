"""GPU diagnostic: per-launch kernel times of the headline scan (looks for alternating / drifting launch times)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from phfpfac_amd import GpuMatcher, PfacTable
DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "data")
para = open(os.path.join(DATA, "paragraph402"), "rb").read()
N = 1 << 30
buf = torch.empty(N + 4096, dtype=torch.uint8, device="cuda:0")
t = PfacTable.from_file(os.path.join(DATA, sys.argv[1] if len(sys.argv) > 1 else "experimentpattern"), 256)
with GpuMatcher(0, 1) as g:
    g.load_table(t); g.fill_tiled(buf, N, para); g.reserve(0, 0, N // 8)
    ms = []
    for _ in range(int(os.environ.get("N_LAUNCH", "24"))):
        g.scan_async(N, N, d_input=buf); g.scan_finish(0); ms.append(g.elapsed_ms(0))
print(" ".join("%.3f" % x for x in ms))
