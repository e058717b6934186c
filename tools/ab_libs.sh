#!/bin/bash
# usage: ab3.sh libA libB workload
cd "$(dirname "$0")/.."
W=${3:-text1g_experimentpattern}
for i in 1 2 3; do
  for v in A B; do
    if [ $v = A ]; then export PFAC_HIP_LIB=$PWD/$1; else export PFAC_HIP_LIB=$PWD/$2; fi
    python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --workload $W 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$v', 'kernel', d['roofline']['achieved'], d['roofline']['kernel_ms_min'], 'value', d['value'], d['ms_per_step'])"
  done
done
