#!/bin/bash
# GPU box: sweep PFAC_LBD (look-back window delay, 10 ns ticks) over workloads
cd "$(dirname "$0")/.."
for w in ${WL:-text1g_experimentpattern rand1g_experimentpattern text1g_snort75k}; do
  for i in 1 2; do
    for k in ${KS:-150 220 300 400}; do
      PFAC_LBD=$k python3 bench.py --steps 12 --warmup 2 --no-cpu-baseline --workload $w 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$w LBD=$k', 'kernel', d['roofline']['achieved'], d['roofline']['kernel_ms_min'], 'value', d['value'])"
    done
  done
done
