#!/bin/bash
# usage (GPU box): bash tools/pmc_diag.sh <workload>      extra PMC passes (texture path, L1, instruction cache) of bench.py on
# one workload, each under its own timeout; raw CSVs under gpurun_out/pmc_diag/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
W=${1:-text1g_dictionary}
i=0
for pass in "TA_TA_BUSY_sum GRBM_GUI_ACTIVE" "TD_TD_BUSY_sum TCP_PENDING_STALL_CYCLES_sum" "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_WAVE_CYCLES" "SQC_ICACHE_MISSES SQC_ICACHE_REQ SQ_IFETCH_LEVEL"; do
  i=$((i+1))
  echo "pass $i: $pass"
  timeout -k 10 100 rocprofv3 --pmc $pass --output-format csv -d gpurun_out/pmc_diag/p$i -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra --no-end-to-end --sustain-seconds 0 --workload $W > gpurun_out/pmc_diag_p$i.log 2>&1 || { echo "pass $i failed"; exit 1; }
done
echo done
