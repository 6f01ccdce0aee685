set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r3q
timeout 400 python scripts/soak.py 240 > gpurun_out/r3q/soak.log 2>&1
tail -n 6 gpurun_out/r3q/soak.log
SOAK_BIG=1 timeout 200 python scripts/soak.py 100 > gpurun_out/r3q/soak_big.log 2>&1
tail -n 4 gpurun_out/r3q/soak_big.log
