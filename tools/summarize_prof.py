#!/usr/bin/env python3
"""Copy the judged rocprofv3 summaries of one tools_prof.sh run from gpurun_out/ (scratch) into profiles/.

usage: python tools/summarize_prof.py <tag> <round-name>      e.g.  r1b r1
writes profiles/<round>_kernel_stats.csv        rocprofv3 --kernel-trace --stats summary of `bench.py --steps 5`
       profiles/<round>_pmc_per_launch.json     PMC counters of the 1 GiB pfac_scan_kernel launches (mean per launch),
                                                one rocprofv3 --pmc pass per counter group, plus derived HBM bytes with
                                                the gfx950 corrections of MI355X_MICROARCH.md (FETCH_SIZE x2 for wide
                                                streaming reads; FETCH/WRITE_SIZE are in KiB)
"""
import collections, csv, glob, json, os, shutil, sys
tag, rnd = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_" + tag)
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)
ks = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))[0]
shutil.copyfile(ks, os.path.join(dst, f"{rnd}_kernel_stats.csv"))
out = {}
for f in sorted(glob.glob(os.path.join(src, "pmc_*", "*", "*_counter_collection.csv"))):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "pfac_scan_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        v = sorted(v)[-3:]          # the full-size (1 GiB) launches of the timed loop
        out[k] = sum(v) / len(v)
if "FETCH_SIZE" in out and "WRITE_SIZE" in out:
    out["derived_hbm_read_bytes"] = out["FETCH_SIZE"] * 1024 * 2     # gfx950: FETCH_SIZE counts 1/2 of wide reads
    out["derived_hbm_write_bytes"] = out["WRITE_SIZE"] * 1024
    out["derived_hbm_bytes"] = out["derived_hbm_read_bytes"] + out["derived_hbm_write_bytes"]
for row in csv.DictReader(open(ks)):
    if "pfac_scan_kernel" in row["Name"]:
        out["kernel_stats_avg_ns"] = float(row["AverageNs"]); out["kernel_stats_max_ns"] = float(row["MaxNs"])
        out["kernel_stats_calls"] = int(row["Calls"])
# per-launch durations from the kernel trace: the --stats average covers EVERY launch of the command (setup, clock
# settling, warm-up, timed steps); bench.py's roofline uses the timed steps only = the last `steps` launches
kt = glob.glob(os.path.join(src, "trace", "*", "*_kernel_trace.csv"))
if kt:
    rows = [r for r in csv.DictReader(open(kt[0])) if "pfac_scan_kernel" in r.get("Kernel_Name", "")]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
    if dur:
        k = int(os.environ.get("TIMED_STEPS", "20"))
        out["kernel_trace_launches"] = len(dur)
        out["kernel_trace_last_steps"] = k
        out["kernel_trace_last_steps_avg_ns"] = sum(dur[-k:]) / len(dur[-k:])
        out["kernel_trace_first_launches_ns"] = dur[:12]
json.dump(out, open(os.path.join(dst, f"{rnd}_pmc_per_launch.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
