"""GPU diagnostic (needs the trips-instrumented library as PFAC_HIP_LIB): look-back loop iterations per resolved round."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "trace.bin")
os.environ["PFAC_TRACE"] = out
import numpy as np, torch
from phfpfac_amd import GpuMatcher, PfacTable
DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "data")
para = open(os.path.join(DATA, "paragraph402"), "rb").read()
N = 1 << 30
buf = torch.empty(N + 4096, dtype=torch.uint8, device="cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "experimentpattern"
t = PfacTable.from_file(os.path.join(DATA, name), 256)
with GpuMatcher(0, 1) as g:
    g.load_table(t); g.fill_tiled(buf, N, para); g.reserve(0, 0, N // 8)
    for _ in range(3):
        g.scan_async(N, N, d_input=buf); g.scan_finish(0)
    print(name, "kernel ms", g.elapsed_ms(0))
d = np.fromfile(out, dtype=np.uint64).reshape(8, 64, 32).astype(np.int64)
tr = d[:, 4:60, 11]
print("look-back trips per round: mean %.2f" % tr.mean(), "histogram", np.bincount(tr.ravel())[:10])
