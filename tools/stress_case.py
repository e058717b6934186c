"""GPU diagnostic: repeat one scan many times and count wrong results (intermittent-failure hunting)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from phfpfac_amd import GpuMatcher, PfacTable
DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "data")
n = 1 << 22
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
table = PfacTable.from_file(os.path.join(DATA, "bytefile_10000byte"), 256)
bad = 0
with GpuMatcher(0, 1) as g:
    g.load_table(table)
    buf = torch.empty(n, dtype=torch.uint8, device="cuda:0")
    g.fill_random(buf, n, 0x5048465046414331)
    g.reserve(0, 0, 1 << 16)
    for i in range(reps):
        cnt = g.scan_resident(n, n, d_input=buf)
        if cnt != 1:
            bad += 1
            if bad <= 3: print("iteration", i, "count", cnt, flush=True)
print(os.environ.get("TAG", ""), "wrong", bad, "of", reps, g.info() if False else "")
