#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/prof.sh <tag> [bench args...]
# 1) kernel trace + stats of `bench.py --no-cpu-baseline --no-extra <args>`  2) PMC passes of the same command with
# 3 steps (separate runs: counters are never collected together with trace domains).  Summaries -> tools/summarize_prof.py
set -o pipefail
tag=$1; shift
export TMPDIR=/tmp
out=$PWD/gpurun_out/prof_$tag
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --no-cpu-baseline --no-extra --no-end-to-end --sustain-seconds 0 "$@" > $out/trace.log 2>&1
for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
            "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $pass --output-format csv -d $out/pmc_$name -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra --no-end-to-end --sustain-seconds 0 "$@" > $out/pmc_$name.log 2>&1 || echo "pmc pass failed: $pass" >> $out/errors.log
done
find $out -name "*.csv" | head -50 > $out/files.txt
tail -1 $out/trace.log
