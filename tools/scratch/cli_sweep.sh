set -e
cd /root/repo
D=tests/golden/data
W=$(mktemp -d)
python3 - "$W" "17179869184" <<'PY'
import sys, os
para = open("tests/golden/data/paragraph402","rb").read()
n = int(sys.argv[2])
with open(os.path.join(sys.argv[1], "text"), "wb") as f:
    blk = (para * (1 + (1 << 24) // 402))
    off = 0
    while off < n:
        k = min(1 << 24, n - off)
        ph = off % 402
        f.write((para[ph:] + blk)[:k])
        off += k
    f.write(b"\n")
PY
cd $W
run() { echo "== S=$S $*"; env "$@" /root/repo/phfpfac_amd/bin/gphf /root/repo/$D/bytefile_10000byte $S 256 $W/text 2>&1 | grep -E "^\[|^2\.|^5\.|inside|rror" | sed 's/(1 worker.*each)//'; }
for S in 1 2 4; do run PFAC_READ_THREADS=12; done
S=4; run PFAC_READ_THREADS=4
S=4; run PFAC_READ_THREADS=12 PFAC_INGEST=pread
S=4; run PFAC_READ_THREADS=12 PFAC_CHUNK_MB=64
S=4; run PFAC_READ_THREADS=12 PFAC_TIMELINE=1
head -c 1073741825 $W/text > $W/text1g
for S in 1 4; do echo "== 1 GiB S=$S"; /root/repo/phfpfac_amd/bin/gphf /root/repo/$D/bytefile_10000byte $S 256 $W/text1g | grep -E "^0\.|^2\.|^5\."; done
echo "== 1 GiB experimentpattern S=1"; PFAC_TIMELINE=1 /root/repo/phfpfac_amd/bin/gphf /root/repo/$D/experimentpattern 1 256 $W/text1g 2>&1 | grep -E "^\[|^0\.|^2\.|^4\.|^5\."
rm -rf $W
