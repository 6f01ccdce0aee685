set -u
export TMPDIR=/tmp
for cfg in "SRHIP_CHAIN=4" "SRHIP_CHAIN=240"; do
echo "== $cfg f32 1080p"
env $cfg SRHIP_TRACE=2 timeout 60 python scripts/run_once.py f32 1080x1920 2 2>&1 | grep "stage 4" | tail -n 3
echo "== $cfg split 1080p"
env $cfg SRHIP_TRACE=2 timeout 60 python scripts/run_once.py split_f16 1080x1920 2 2>&1 | grep "stage 4" | tail -n 3
done
