set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r3s
SOAK_BIG=1 timeout 500 python scripts/soak.py 420 > gpurun_out/r3s/soak_big.log 2>&1
tail -n 3 gpurun_out/r3s/soak_big.log
