#!/bin/bash
# A/B of the product kernels against an OLDER COMMIT's kernel source on one GPU box -- how a regression of a few per cent in a
# kernel that a change was not even meant to touch is caught (profiles/r3_ab_vs_round2_kernel.log).
#   tools/ab_kernel.sh build <commit> <tag>      (here, CPU) compiles <commit>'s pfac_hip.hip against ITS pfac.h into
#                                                abx/lib_<tag>.so, with stubs for entry points added since (so that today's
#                                                ctypes mirror loads it)
#   tools/ab_kernel.sh run "<tag> ... cur"       (GPU box) tools/series.py per library and workload, 800 launches each,
#                                                mean of the second half; "cur" = the product library
set -e
export PFAC_ENABLE_KNOBS=1
cd "$(dirname "$0")/.."
case "$1" in
build)
  c=$2; tag=$3; d=$(mktemp -d); mkdir -p $d/include abx
  git show $c:phfpfac_amd/csrc/pfac_hip.hip > $d/pfac_hip.hip
  git show $c:include/pfac.h > $d/include/pfac.h
  echo 'extern "C" {' > $d/stubs.cc
  for s in $(python3 -c "import sys; sys.path.insert(0, '.'); from phfpfac_amd import _ffi; print(' '.join(_ffi.HIP_SYMBOLS))"); do
    grep -q "$s" $d/pfac_hip.hip || echo "int $s() { return -7; }" >> $d/stubs.cc
  done
  echo '}' >> $d/stubs.cc
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I$d/include -o abx/lib_$tag.so $d/pfac_hip.hip $d/stubs.cc
  echo "built abx/lib_$tag.so from $c";;
run)
  for v in $2; do
    if [ $v = cur ]; then unset PFAC_HIP_LIB; else export PFAC_HIP_LIB=$PWD/abx/lib_$v.so; fi
    for w in "experimentpattern text" "experimentpattern rand" "bytefile_10000byte text" "bytefile_1000000byte.gz text" "bytefile_1000000byte.gz rand"; do
      echo -n "lib=$v [$w]: "; N_LAUNCH=${N_LAUNCH:-800} python3 tools/series.py $w 2>&1 | tail -1 | sed "s/.*second half/second half/"
    done
    echo -n "lib=$v [dictionary]: "; N_LAUNCH=24 python3 tools/series.py xaa+xab+xac+xad text 2>&1 | tail -1 | sed "s/.*second half/second half/"
  done;;
*) sed -n 2,10p "$0";;
esac
