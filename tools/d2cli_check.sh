set -e
cd $GRAFT_REPO_ROOT
D=tests/golden/data
W=/tmp/d2cli; mkdir -p $W
cat $D/xaa $D/xab $D/xac $D/xad > $W/all.pat
python3 - <<'PY'
import os
para=open('tests/golden/data/paragraph402','rb').read()
n=256<<20
buf=(para*(n//402+2))[:n]
open('/tmp/d2cli/big.txt','wb').write(buf)
PY
cd $W
for mode in new old; do
  if [ $mode = old ]; then export PFAC_ENABLE_KNOBS=1 PFAC_NO_DENSE2=1; fi
  $GRAFT_REPO_ROOT/phfpfac_amd/bin/gphf all.pat 4 256 big.txt > log_$mode.txt 2>&1 || { tail -5 log_$mode.txt; exit 1; }
  tail -2 log_$mode.txt
  md5sum GPU_match_result.txt | tee md5_$mode.txt; wc -l GPU_match_result.txt
done
