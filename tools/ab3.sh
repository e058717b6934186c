#!/bin/bash
# usage: tools/ab3.sh workload lib1 lib2 ...   (GPU box) -- interleaved rounds of bench.py over prebuilt libraries
cd "$(dirname "$0")/.."
W=$1; shift
for i in 1 2 3; do
  for L in "$@"; do
    PFAC_HIP_LIB=$PWD/$L python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --workload $W 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$L', 'kernel', d['roofline']['achieved'], d['roofline']['kernel_ms_min'], 'value', d['value'])"
  done
done
