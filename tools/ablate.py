"""GPU diagnostic: kernel time with parts of the per-tile work skipped (PFAC_ABLATE bits; results are wrong)."""
import os, sys, subprocess
here = os.path.dirname(os.path.abspath(__file__))
code = r'''
import os, sys
sys.path.insert(0, os.path.dirname(%r))
import torch
from phfpfac_amd import GpuMatcher, PfacTable
DATA = os.path.join(os.path.dirname(%r), "tests", "golden", "data")
para = open(os.path.join(DATA, "paragraph402"), "rb").read()
N = 1 << 30
buf = torch.empty(N + 4096, dtype=torch.uint8, device="cuda:0")
t = PfacTable.from_file(os.path.join(DATA, sys.argv[1]), 256)
with GpuMatcher(0, 1) as g:
    g.load_table(t); g.fill_tiled(buf, N, para); g.reserve(0, 0, N // 8)
    ms = []
    for _ in range(6):
        g.scan_async(N, N, d_input=buf)
        try: g.scan_finish(0, allow_overflow=True)
        except Exception as e: pass
        ms.append(g.elapsed_ms(0))
    print("%%-20s ablate=%%-3s kernel %%.3f ms  %%.0f GB/s" %% (sys.argv[1], os.environ.get("PFAC_ABLATE", "0"), min(ms[1:]), N/min(ms[1:])/1e6), flush=True)
''' % (here, here)
for ab in ("0", "4", "2", "6", "1", "9"):
    e = dict(os.environ); e["PFAC_ABLATE"] = ab
    subprocess.run([sys.executable, "-c", code, "experimentpattern"], env=e)
