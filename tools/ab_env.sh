#!/bin/bash
# usage: tools/ab_env.sh "<env for A>" "<env for B>" [workload]   (GPU box) -- interleaved rounds of bench.py
cd "$(dirname "$0")/.."
W=${3:-text1g_experimentpattern}
for i in 1 2 3; do
  for v in A B; do
    if [ $v = A ]; then E="$1"; else E="$2"; fi
    env $E python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --workload $W 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$v', 'kernel', d['roofline']['achieved'], d['roofline']['kernel_ms_min'], 'value', d['value'], 'grid', d['config']['grid_blocks'], 'lds', d['config']['lds_bytes'])"
  done
done
