import os, sys, subprocess
here = os.path.dirname(os.path.abspath(__file__))
code = r'''
import os, sys
sys.path.insert(0, os.path.dirname(%r))
import torch
from phfpfac_amd import GpuMatcher, PfacTable
DATA = os.path.join(os.path.dirname(%r), "tests", "golden", "data")
para = open(os.path.join(DATA, "paragraph402"), "rb").read()
N = 1 << 28
buf = torch.empty(N + 4096, dtype=torch.uint8, device="cuda:0")
t = PfacTable.from_file(os.path.join(DATA, sys.argv[1]), 256)
with GpuMatcher(0, 1) as g:
    g.load_table(t); g.fill_tiled(buf, N, para); g.reserve(0, 0, N // 8)
    n = g.scan_resident(N, N, d_input=buf)
    ms = []
    for _ in range(5):
        g.scan_async(N, N, d_input=buf); g.scan_finish(0); ms.append(g.elapsed_ms(0))
    print(sys.argv[1], os.environ.get("PFAC_FORCE_L2"), os.environ.get("PFAC_NWB"), "matches", n, "GB/s %%.1f" %% (N/min(ms)/1e6), g.info(), flush=True)
''' % (here, here)
for pat in ("bytefile_10000byte",):
    for env in ({}, {"PFAC_FORCE_L2": "1"}, {"PFAC_NWB": "8"}, {"PFAC_NWB": "12"}, {"PFAC_NWB": "13"}, {"PFAC_FORCE_L2": "1", "PFAC_NWB": "14"}, {"PFAC_FORCE_L2": "1", "PFAC_NWB": "8"}):
        e = dict(os.environ); e.update(env)
        subprocess.run([sys.executable, "-c", code, pat], env=e)
