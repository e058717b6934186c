"""GPU diagnostic: kernel time of several prebuilt libraries on the headline input WITHOUT result checks
(ablation builds compute wrong answers on purpose).  usage: time_libs.py lib1.so lib2.so ..."""
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
t = PfacTable.from_file(os.path.join(DATA, "experimentpattern"), 256)
with GpuMatcher(0, 1) as g:
    g.load_table(t); g.fill_tiled(buf, N, para); g.reserve(0, 0, N // 8)
    ms = []
    for _ in range(10):
        g.scan_async(N, N, d_input=buf); g.scan_finish(0, allow_overflow=True); ms.append(g.elapsed_ms(0))
    print(os.path.basename(os.environ.get("PFAC_HIP_LIB", "default")), "kernel min %%.4f ms avg %%.4f" %% (min(ms[2:]), sum(ms[2:])/len(ms[2:])), flush=True)
''' % (here, here)
for i in range(2):
    for lib in sys.argv[1:]:
        e = dict(os.environ)
        e["PFAC_HIP_LIB"] = os.path.abspath(lib)
        subprocess.run([sys.executable, "-c", code], env=e)
