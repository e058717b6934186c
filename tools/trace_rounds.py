"""GPU diagnostic: per-round timestamps of the coordinator and of compute wave 0 (PFAC_TRACE)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "trace.bin")
os.environ["PFAC_TRACE"] = out
os.environ["PFAC_HIP_LIB"] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ab", "libpfac_hip_trace.so")   # make -C phfpfac_amd/csrc trace
import numpy as np, torch
from phfpfac_amd import GpuMatcher, PfacTable
DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "data")
para = open(os.path.join(DATA, "paragraph402"), "rb").read()
N = 1 << 30
pats = {"exp": open(os.path.join(DATA, "experimentpattern"), "rb").read(), "nomatch": b"\x01\x02\n"}[sys.argv[1] if len(sys.argv) > 1 else "exp"]
buf = torch.empty(N + 4096, dtype=torch.uint8, device="cuda:0")
t = PfacTable.from_bytes(pats, 256)
with GpuMatcher(0, 1) as g:
    g.load_table(t); g.fill_tiled(buf, N, para); g.reserve(0, 0, N // 8)
    for _ in range(3):
        g.scan_async(N, N, d_input=buf); g.scan_finish(0)
    print("kernel ms", g.elapsed_ms(0))
d = np.fromfile(out, dtype=np.uint64).reshape(8, 64, 32).astype(np.int64)
for b in (0, 3):
    x = d[b]
    t0 = x[0, 4] if x[0, 4] else x[0, 0]
    print(f"block {b}: times in us relative to first stamp; columns:")
    print("  r | coord: start arrivals_done lookback_done batch | compute w0: top lds_written masks_done pass_done end base_ready")
    for r in range(0, 40):
        if x[r, 0] == 0 and x[r, 4] == 0: break
        f = lambda v: f"{(v - t0) / 100:8.2f}" if v else "    -   "
        print(f"{r:3d} | {f(x[r,0])} {f(x[r,1])} {f(x[r,2])} {x[r,3]:7d} | {f(x[r,4])} {f(x[r,5])} {f(x[r,6])} {f(x[r,7])} {f(x[r,8])} {f(x[r,9])}")

x = d[0]
t0 = x[0, 4]
print("block 0: per-wave pass_done (us) per round")
for r in range(0, 16):
    if x[r, 16] == 0: break
    print(f"{r:3d} | " + " ".join(f"{(x[r,16+w]-t0)/100:7.2f}" if x[r,16+w] else "   -   " for w in range(15)))
