#!/bin/bash
cd "$(dirname "$0")/.."
for w in ${WL:-text1g_experimentpattern text1g_bytefile10000 text1g_snort75k rand1g_experimentpattern}; do
  for i in 1 2; do
    for k in ${KS:-1 2 4}; do
      PFAC_TICKET_WAYS=$k python3 bench.py --steps 12 --warmup 2 --no-cpu-baseline --workload $w 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$w WAYS=$k', 'kernel', d['roofline']['achieved'], d['roofline']['kernel_ms_min'], 'value', d['value'])"
    done
  done
done
