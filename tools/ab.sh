#!/bin/bash
# A/B two builds of the HIP library in ONE gpurun call (same device, interleaved rounds).
#   tools/ab.sh build <git-rev>     (here, CPU) compiles that revision's pfac_hip.hip into gpurun_out/../ab/libB.so
#   tools/ab.sh run [workload]      (GPU box) interleaves 3 rounds of each
set -e
cd "$(dirname "$0")/.."
case "$1" in
build)
  mkdir -p ab
  git show "$2":phfpfac_amd/csrc/pfac_hip.hip > ab/pfac_hip_B.hip
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Iinclude -o ab/libB.so ab/pfac_hip_B.hip
  echo "built ab/libB.so from $2";;
run)
  W=${2:-text1g_experimentpattern}
  for i in 1 2 3; do
    for v in A B; do
      if [ $v = B ]; then export PFAC_HIP_LIB=$PWD/ab/libB.so; else unset PFAC_HIP_LIB; fi
      python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --workload $W 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$v', 'kernel', d['roofline']['achieved'], d['roofline']['kernel_ms_min'], 'value', d['value'], d['ms_per_step'])"
    done
  done;;
esac
