#!/bin/bash
# A/B/n of several builds of the HIP library in ONE gpurun call (same device, interleaved repetitions).
#   tools/abn.sh build NAME "-DPFAC_X=1 ..."     (here, CPU) compiles the working tree's pfac_hip.hip into abx/lib_NAME.so
#   tools/abn.sh ablations [NAME ...]            (here, CPU) builds the named ablation libraries below (default: all), 4 at a time
#   tools/abn.sh run "NAME1 NAME2 ..." "workload1 workload2 ..." [reps]      (GPU box; NAME "cur" = the product library)
#   tools/abn.sh series "NAME1 NAME2 ..." "PATTERNFIXTURE text|rand" [...]   (GPU box) tools/series.py per library (no parity check:
#                                                                            the way to time ablation builds, whose records are wrong)
# abx/ is scratch: git-ignored, NOT gpurun-ignored (the libraries must travel to the GPU box) -- empty it when done.
# (ab/, the round-2 scratch directory, is gpurun-ignored.)
#
# The ablation libraries the profiles/r*_ablation_*.log files name, and the exact flags each is built with
# (wrong results on purpose -- each removes one stage of the kernel so its cost can be read off the difference):
declare -A ABL=(
  [cur0]=""                          # the product source, unmodified (the baseline of every ablation log)
  [noroot]="-DPFAC_ABL_NOROOT"       # no root test (nothing survives): streaming + LDS mirror only
  [noclass]="-DPFAC_ABL_NOCLASS"     # root test, but no level-2 (byte pair) lookups
  [nokeep]="-DPFAC_ABL_NOKEEP"       # survivors classified, then dropped: no compaction, no walks, no records
  [nowalk]="-DPFAC_ABL_NOWALK"       # deep survivors treated as shallow: compaction + staging + emission, no walks
  [nostage]="-DPFAC_ABL_NOSTAGE"     # records counted and placed, never staged
  [noemit]="-DPFAC_ABL_NOEMIT"       # records staged, never written to memory
  [nolds]="-DPFAC_ABL_NOLDSCOPY"     # the tile never reaches LDS
  [nocoord]="-DPFAC_ABL_NOCOORD"     # static tiles, no coordinator wave, counts dropped
  [static]="-DPFAC_ABL_STATIC"       # batches dealt round-robin instead of by ticket
  [d2nowalk]="-DPFAC_ABL_D2NOWALK"   # dense mode, second form: nothing goes beyond its second byte (front end + scatter only)
  [d2noscat]="-DPFAC_ABL_D2NOSCATTER" # dense mode, second form: the records never leave the log
)
set -e
export PFAC_ENABLE_KNOBS=1
cd "$(dirname "$0")/.."
HIPCC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Iinclude"
case "$1" in
build)
  mkdir -p abx
  $HIPCC $3 -o abx/lib_$2.so phfpfac_amd/csrc/pfac_hip.hip
  echo "built abx/lib_$2.so ($3)";;
ablations)
  mkdir -p abx
  shift
  names="${*:-${!ABL[@]}}"
  n=0
  for v in $names; do
    ( $HIPCC ${ABL[$v]} -o abx/lib_$v.so phfpfac_amd/csrc/pfac_hip.hip && echo "built abx/lib_$v.so (${ABL[$v]:-no flags})" ) &
    n=$((n + 1)); if [ $((n % 4)) = 0 ]; then wait; fi
  done
  wait;;
run)
  for i in $(seq 1 ${4:-2}); do
    for W in $3; do
      for v in $2; do
        if [ $v = cur ]; then unset PFAC_HIP_LIB; else export PFAC_HIP_LIB=$PWD/abx/lib_$v.so; fi
        python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra --no-end-to-end --sustain-seconds 0 --workload $W 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$W', '$v', 'kernel GB/s', d['roofline']['achieved'], 'min ms', d['roofline']['kernel_ms_min'], 'value', d['value'])"
      done
    done
  done;;
series)
  libs="$2"; shift 2
  for W in "$@"; do
    for v in $libs; do
      if [ $v = cur ]; then unset PFAC_HIP_LIB; else export PFAC_HIP_LIB=$PWD/abx/lib_$v.so; fi
      echo -n "[$W] lib_$v: "; python3 tools/series.py $W 2>&1 | tail -1
    done
  done;;
*) sed -n 2,12p "$0";;
esac
