#!/bin/bash
# GPU box: instruction-cache counters for the headline launch.
set -o pipefail
export TMPDIR=/tmp
out=$PWD/gpurun_out/pmc_icache
mkdir -p $out
i=0
for pass in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" \
            "SQC_ICACHE_BUSY_CYCLES SQC_ICACHE_INPUT_VALID_READYB SQC_TC_INST_REQ SQC_TC_STALL" \
            "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_BUSY_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $pass --output-format csv -d $out/p$i -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra --no-end-to-end --sustain-seconds 0 "$@" > $out/p$i.log 2>&1 || echo "pass failed: $pass" >> $out/errors.log
done
python3 - $out <<'PY'
import sys, glob, csv, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "pfac_scan_kernel" in r.get("Kernel_Name", ""):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]
    print(f"{k:32s} per-launch mean {sum(v)/len(v):16.1f}  (n={len(v)})")
PY
