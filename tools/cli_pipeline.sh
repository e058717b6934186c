#!/bin/bash
# GPU box: end-to-end gphf runs (ingest + H2D + scan + text back + write) on a generated text file: the PCIe-inclusive rate.
#   SIZE=bytes (default 1 GiB)  STREAMS="1 4"  PATS="bytefile_10000byte experimentpattern"  EMIT="device host"  INGEST="mmap pread"
set -e
cd "$(dirname "$0")/.."
ROOT=$PWD
D=tests/golden/data
W=$(mktemp -d)
python3 - "$W" "${SIZE:-1073741824}" <<'PY'
import sys, os
para = open("tests/golden/data/paragraph402","rb").read()
n = int(sys.argv[2])
with open(os.path.join(sys.argv[1], "text"), "wb") as f:
    blk = (para * (1 + (1 << 24) // 402))
    # keep the 402-byte phase continuous across blocks
    off = 0
    while off < n:
        k = min(1 << 24, n - off)
        ph = off % 402
        f.write((para[ph:] + blk)[:k])
        off += k
    f.write(b"\n")
PY
for i in ${INGEST:-mmap}; do
for e in ${EMIT:-device}; do
for s in ${STREAMS:-1 4}; do
  for p in ${PATS:-bytefile_10000byte experimentpattern}; do
    (
      R=$(mktemp -d -p $W); cd $R
      T0=$(date +%s.%N)
      PFAC_INGEST=$i PFAC_EMIT=$e $ROOT/phfpfac_amd/bin/gphf $ROOT/$D/$p $s 256 $W/text | grep -E "^0\.|^2\.|^3\.|^4\.|^5\.|^!!"
      T1=$(date +%s.%N)
      python3 -c "print('wall %.2f s (whole process incl. exec, HIP start-up, table build, ingest, emit)' % ($T1 - $T0))"
      ls -la GPU_match_result.txt | awk '{print "output bytes", $5}'
    ) 2>&1 | sed "s/^/[$p streams=$s ingest=$i emit=$e] /"
  done
done
done
done
rm -rf $W
