#!/usr/bin/env python3
"""Copy the judged rocprofv3 summaries of one tools/prof.sh run from gpurun_out/ (scratch) into profiles/.

usage: python tools/summarize_prof.py <tag> <round> [suffix]      e.g.  head r2      or   snortrand r2 rand1g_snort75k
writes profiles/<round>_kernel_stats[_suffix].csv      rocprofv3 --kernel-trace --stats summary of the bench command
       profiles/<round>_pmc_per_launch[_suffix].json   PMC counters of the full-size pfac_scan_kernel launches (mean per
                                                       launch), one rocprofv3 --pmc pass per counter group, plus derived
                                                       HBM bytes with the gfx950 corrections of MI355X_MICROARCH.md
                                                       (FETCH_SIZE x2 for wide streaming reads; FETCH/WRITE_SIZE in KiB),
                                                       per-input-byte rates, and the sha256 of the kernel source profiled
"""
import collections, csv, glob, hashlib, json, os, shutil, sys
tag, rnd = sys.argv[1], sys.argv[2]
suffix = "_" + sys.argv[3] if len(sys.argv) > 3 else ""
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_" + tag)
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)
def newest(pattern):
    """gpurun_out/ keeps the files of earlier runs of the same tag: only the most recent one counts."""
    files = sorted(glob.glob(pattern), key=os.path.getmtime)
    return files[-1] if files else None


ks = newest(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))
shutil.copyfile(ks, os.path.join(dst, f"{rnd}_kernel_stats{suffix}.csv"))
out = {"kernel_source_sha256": hashlib.sha256(open(os.path.join(root, "phfpfac_amd", "csrc", "pfac_hip.hip"), "rb").read()).hexdigest()}
try:
    line = [l for l in open(os.path.join(src, "trace.log")) if l.startswith("{")][-1]
    b = json.loads(line)
    out["bench_under_trace"] = {"workload": b["config"]["workload"], "kernel_ms_avg": b["roofline"]["kernel_ms_avg"],
                                "achieved_gbs": b["roofline"]["achieved"], "matches_per_step": b["config"]["matches_per_step"],
                                "bytes_per_gpu": b["config"]["bytes_per_gpu"], "steps": b["steps"]}
    n_bytes = b["config"]["bytes_per_gpu"]
except (IndexError, OSError, ValueError, KeyError):
    n_bytes = 1 << 30
for f in sorted(filter(None, (newest(os.path.join(d, "*", "*_counter_collection.csv")) for d in glob.glob(os.path.join(src, "pmc_*")) if os.path.isdir(d)))):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if "pfac_scan_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]][r["Kernel_Name"]].append(float(r["Counter_Value"]))
    for k, by_kernel in agg.items():
        # the kernel of the timed loop = the instantiation with the most launches (a workload that adapts its staging
        # mode runs ONE launch of another instantiation first: four times the cost, and not what is being measured)
        name, v = max(by_kernel.items(), key=lambda kv: len(kv[1]))
        v = sorted(v)[-3:]          # the full-size launches of the timed loop
        out[k] = sum(v) / len(v)
        out.setdefault("pmc_kernel", name)
if "FETCH_SIZE" in out and "WRITE_SIZE" in out:
    out["derived_hbm_read_bytes"] = out["FETCH_SIZE"] * 1024 * 2     # gfx950: FETCH_SIZE counts 1/2 of wide reads
    out["derived_hbm_write_bytes"] = out["WRITE_SIZE"] * 1024
    out["derived_hbm_bytes"] = out["derived_hbm_read_bytes"] + out["derived_hbm_write_bytes"]
if "SQ_INSTS_VALU" in out:
    tiles = n_bytes / 4096
    out["per_tile"] = {k: round(out[k] / tiles, 1) for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR") if k in out}
    out["vmem_rd_wave_instr_per_input_byte"] = out.get("SQ_INSTS_VMEM_RD", 0) / n_bytes
if "TCC_HIT_sum" in out:
    out["tcc_hit_rate"] = out["TCC_HIT_sum"] / (out["TCC_HIT_sum"] + out["TCC_MISS_sum"])
if "SQ_WAVE_CYCLES" in out:
    for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS"):
        if k in out:
            out["frac_" + k] = round(out[k] / out["SQ_WAVE_CYCLES"], 4)
if "SQ_LDS_IDX_ACTIVE" in out and out["SQ_LDS_IDX_ACTIVE"]:
    out["lds_bank_conflict_frac"] = round(out.get("SQ_LDS_BANK_CONFLICT", 0) / out["SQ_LDS_IDX_ACTIVE"], 4)
if out.get("GRBM_GUI_ACTIVE") and out.get("SQ_INSTS_VALU"):
    # busy fractions of the execution units over the launch: GRBM_GUI_ACTIVE is summed over the 8 XCDs; a wave64 VALU
    # instruction occupies its SIMD (16 lanes wide, 4 per CU) for 4 cycles; the scalar unit and the LDS are one per CU
    cyc = out["GRBM_GUI_ACTIVE"] / 8.0
    out["util"] = {"valu_pipe": round(out["SQ_INSTS_VALU"] * 4 / (1024 * cyc), 3),
                   "salu": round(out.get("SQ_INSTS_SALU", 0) / (256 * cyc), 3),
                   "lds": round(out.get("SQ_LDS_IDX_ACTIVE", 0) / (256 * cyc), 3),
                   "note": "VALU: wave-instructions x 4 cycles / (1024 SIMDs x kernel cycles); SALU, LDS: per CU"}
for row in csv.DictReader(open(ks)):
    if "pfac_scan_kernel" in row["Name"]:
        out.setdefault("kernel_stats", []).append({"name": row["Name"][:120], "calls": int(row["Calls"]),
                                                   "avg_ns": float(row["AverageNs"]), "max_ns": float(row["MaxNs"])})
# per-launch durations from the kernel trace: the --stats average covers EVERY launch of the command (setup, clock
# settling, warm-up, timed steps); bench.py's roofline uses the timed steps only = the last `steps` launches
kt = newest(os.path.join(src, "trace", "*", "*_kernel_trace.csv"))
if kt:
    rows = [r for r in csv.DictReader(open(kt)) if "pfac_scan_kernel" in r.get("Kernel_Name", "")]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
    if dur:
        k = int(os.environ.get("TIMED_STEPS", str(out.get("bench_under_trace", {}).get("steps", 20))))
        out["kernel_trace_launches"] = len(dur)
        out["kernel_trace_last_steps"] = k
        out["kernel_trace_last_steps_avg_ns"] = sum(dur[-k:]) / len(dur[-k:])
        out["kernel_trace_first_launches_ns"] = dur[:12]
        if rows and "VGPR_Count" in rows[-1]:
            out["vgpr_count"] = rows[-1].get("VGPR_Count"); out["sgpr_count"] = rows[-1].get("SGPR_Count")
            out["lds_block_size"] = rows[-1].get("LDS_Block_Size")
json.dump(out, open(os.path.join(dst, f"{rnd}_pmc_per_launch{suffix}.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
