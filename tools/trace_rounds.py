"""GPU diagnostic: per-round timestamps of the coordinator and of compute wave 0 (trace build: make -C phfpfac_amd/csrc trace).
usage: trace_rounds.py <pattern fixture | nomatch1> [text|rand]"""
import os, sys
os.environ.setdefault("PFAC_ENABLE_KNOBS", "1")     # tuning / test knobs of libpfac_hip.so are opt-in
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out = os.path.join(ROOT, "gpurun_out", "trace.bin")
os.environ["PFAC_TRACE"] = out
os.environ["PFAC_HIP_LIB"] = os.path.join(ROOT, "abx", "libpfac_hip_trace.so")
import numpy as np, torch
from phfpfac_amd import GpuMatcher, PfacTable
DATA = os.path.join(ROOT, "tests", "golden", "data")
para = open(os.path.join(DATA, "paragraph402"), "rb").read()
N = 1 << 30
name = sys.argv[1] if len(sys.argv) > 1 else "experimentpattern"
kind = sys.argv[2] if len(sys.argv) > 2 else "text"
path = os.path.join(DATA, name)
if "+" in name:
    import tempfile
    path = os.path.join(tempfile.mkdtemp(), "all.pat")
    open(path, "wb").write(b"".join(open(os.path.join(DATA, p), "rb").read() for p in name.split("+")))
if name.endswith(".gz"):
    import gzip, tempfile
    path = os.path.join(tempfile.mkdtemp(), name[:-3]); open(path, "wb").write(gzip.open(os.path.join(DATA, name), "rb").read())
t = PfacTable.from_bytes(b"\x01\x02\n", 256) if name == "nomatch1" else PfacTable.from_file(path, 256)
buf = torch.empty(N + 4096, dtype=torch.uint8, device="cuda:0")
with GpuMatcher(0, 1) as g:
    g.load_table(t)
    if kind == "rand": g.fill_random(buf, N, 0x5048465046414331)
    else: g.fill_tiled(buf, N, para)
    g.reserve(0, 0, N // 8)
    g.scan_resident(N, N, d_input=buf)         # sizes the record heap (and lets the staging mode adapt)
    for _ in range(3):
        g.scan_async(N, N, d_input=buf); n, _ = g.scan_finish(0)
    print(name, kind, "kernel ms", g.elapsed_ms(0), "matches", n, g.info())
d = np.fromfile(out, dtype=np.uint64).reshape(8, 64, 32).astype(np.int64)
for b in (0, 3):
    x = d[b]
    t0 = x[0, 4] if x[0, 4] else x[0, 0]
    print(f"block {b}: times in us relative to first stamp; columns:")
    print("  r | coord: start arrivals_done placed batch | compute w0: top lds_written masks_done pass_done end base_ready")
    for r in range(0, 24):
        if x[r, 0] == 0 and x[r, 4] == 0: break
        f = lambda v: f"{(v - t0) / 100:8.2f}" if v else "    -   "
        print(f"{r:3d} | {f(x[r,0])} {f(x[r,1])} {f(x[r,2])} {x[r,3]:7d} | {f(x[r,4])} {f(x[r,5])} {f(x[r,6])} {f(x[r,7])} {f(x[r,8])} {f(x[r,9])}")
    # mean phase durations of compute wave 0 over rounds 4..40
    rows = [x[r] for r in range(4, 60) if x[r, 4] and x[r + 1, 4]]
    if rows:
        ph = {"load wait + lds write": np.mean([r_[5] - r_[4] for r_ in rows]), "epoch + root/level-2": np.mean([r_[6] - r_[5] for r_ in rows]),
              "compact + walk + stage": np.mean([r_[7] - r_[6] for r_ in rows]), "post + emit": np.mean([r_[8] - r_[7] for r_ in rows])}
        if rows[0][10]:
            fine = {"lds write->loads issued": np.mean([r_[10] - r_[5] for r_ in rows]), "emit r-2": np.mean([r_[11] - r_[10] for r_ in rows]),
                    "root half0": np.mean([r_[12] - r_[11] for r_ in rows]), "level2 half0": np.mean([r_[13] - r_[12] for r_ in rows]),
                    "root half1": np.mean([r_[14] - r_[13] for r_ in rows]), "level2 half1": np.mean([r_[6] - r_[14] for r_ in rows])}
            print("  classification, fine (us):", {k: round(v / 100, 2) for k, v in fine.items()})
        per = np.mean(np.diff([x[r, 4] for r in range(4, 60) if x[r, 4]]))
        print("  wave 0 mean phases (us):", {k: round(v / 100, 2) for k, v in ph.items()}, "round", round(per / 100, 2))

# launch skeleton of the 8 traced workgroups: entry -> tables staged -> first tile in LDS ... last wave out
ent, stg, out_ = d[:, 0, 15], d[:, 1, 15], d[:, 2, 15]
if ent.any():
    e0 = ent[ent > 0].min()
    first_lds = d[:, 0, 5]
    last_round_end = np.array([max(d[b, r, 8] for r in range(64)) for b in range(8)])
    n_rounds = [int((d[b, :, 4] > 0).sum()) for b in range(8)]
    print("launch skeleton (us after the earliest traced entry), workgroups 0..7:")
    for name, v in (("entry", ent), ("tables staged (after the barrier)", stg), ("first tile of wave 0 in LDS", first_lds),
                    ("wave 0 done with its last traced round", last_round_end), ("last wave of the workgroup out", out_)):
        print(f"  {name:42s}", " ".join(f"{(x_ - e0) / 100:8.2f}" if x_ else "    -   " for x_ in v))
    print("  (rounds traced per workgroup:", n_rounds, "; at most 64)")

acc = d[:, 0, 10:15].sum(axis=0).astype(float)
rows2 = [d[0][r] for r in range(4, 60) if d[0][r, 4] and d[0][r, 12] and d[0][r, 13] and d[0][r, 14]]
if rows2:
    # dense mode's second form: the accumulators and the fine stamps mean something else there
    print("dense mode, second form -- wave 0 of workgroup 0, mean per tile (us):",
          {"load wait + lds write": round(float(np.mean([r_[5] - r_[4] for r_ in rows2])) / 100, 2),
           "probe + next loads": round(float(np.mean([r_[10] - r_[5] for r_ in rows2])) / 100, 2),
           "dense2_tile": round(float(np.mean([r_[12] - r_[10] for r_ in rows2])) / 100, 2),
           "count posted + heap atomic back": round(float(np.mean([r_[13] - r_[12] for r_ in rows2])) / 100, 2),
           "scatter": round(float(np.mean([r_[14] - r_[13] for r_ in rows2])) / 100, 2),
           "tile to tile": round(float(np.mean(np.diff([d[0][r, 4] for r in range(4, 60) if d[0][r, 4]]))) / 100, 2)})
    print(f"  inside dense2_tile (8 workgroups, whole launch): {acc[3]:.0f} tiles, {acc[2] / acc[3] / 100:.2f} us per tile, of it front end "
          f"{acc[0] / acc[3] / 100:.2f} us; {acc[1] / acc[3]:.1f} walk iterations per tile at {(acc[2] - acc[0]) / max(acc[1], 1) / 100:.2f} us each; {acc[4] / acc[3]:.0f} records per tile")
elif acc[3]:
    print(f"tile_pass of compute wave 0 (8 workgroups, whole launch): {acc[3]:.0f} tiles with FIFO work, {acc[1] / acc[3]:.2f} rounds per tile, "
          f"{acc[0] / max(acc[1], 1) / 100:.2f} us per round, {acc[2] / acc[3] / 100:.2f} us per tile_pass (rounds {acc[0] / acc[3] / 100:.2f} us), last-round entries {acc[4] / acc[3]:.1f}")
