#!/bin/bash
# usage: tools/ab_long.sh workload libA libB   (GPU box) -- interleaved, 60 steps each (averages over the clock drift)
cd "$(dirname "$0")/.."
W=$1; shift
for i in 1 2 3; do
  for L in "$@"; do
    PFAC_HIP_LIB=$PWD/$L python3 bench.py --steps 60 --warmup 3 --no-cpu-baseline --workload $W 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$L', 'kernel', d['roofline']['achieved'], d['roofline']['kernel_ms_min'], 'value', d['value'])"
  done
done
