#!/bin/bash
# GPU box: extra SQ counter passes for the headline launch (instruction fetch, LDS/VMEM latency levels, scalar pipe).
set -o pipefail
export TMPDIR=/tmp
out=$PWD/gpurun_out/pmc_extra
mkdir -p $out
i=0
for pass in "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_CYCLES SQ_BUSY_CU_CYCLES" \
            "SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM" \
            "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_THREAD_CYCLES_VALU" \
            "SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL"; do
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
