set -u
export TMPDIR=/tmp
timeout 200 python scripts/experiments/bands_exp.py 2>/dev/null
