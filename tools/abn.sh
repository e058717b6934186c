#!/bin/bash
# A/B/n of several builds of the HIP library in ONE gpurun call (same device, interleaved repetitions).
#   tools/abn.sh build NAME "-DPFAC_X=1 ..."     (here, CPU) compiles the working tree's pfac_hip.hip into ab/lib_NAME.so
#   tools/abn.sh run "NAME1 NAME2 ..." "workload1 workload2 ..." [reps]      (GPU box; NAME "cur" = the product library)
# Ablation builds (wrong results on purpose; time them with tools/series.py, which does not check parity):
#   -DPFAC_ABL_NOROOT (no root test)  -DPFAC_ABL_NOCLASS (no level-2 lookups)  -DPFAC_ABL_NOKEEP (survivors classified, then dropped)
#   -DPFAC_ABL_NOWALK (deep survivors treated as shallow)  -DPFAC_ABL_NOSTAGE (records counted, never staged)
#   -DPFAC_ABL_NOEMIT (records never leave LDS)  -DPFAC_ABL_NOLDSCOPY  -DPFAC_ABL_NOCOORD  -DPFAC_ABL_STATIC
set -e
cd "$(dirname "$0")/.."
case "$1" in
build)
  mkdir -p ab
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Iinclude $3 -o ab/lib_$2.so phfpfac_amd/csrc/pfac_hip.hip
  echo "built ab/lib_$2.so ($3)";;
run)
  for i in $(seq 1 ${4:-2}); do
    for W in $3; do
      for v in $2; do
        if [ $v = cur ]; then unset PFAC_HIP_LIB; else export PFAC_HIP_LIB=$PWD/ab/lib_$v.so; fi
        python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra --workload $W 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$W', '$v', 'kernel GB/s', d['roofline']['achieved'], 'min ms', d['roofline']['kernel_ms_min'], 'value', d['value'])"
      done
    done
  done;;
esac
