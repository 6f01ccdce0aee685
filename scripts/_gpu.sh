set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r3zz
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r3zz/pytest.log 2>&1
echo "pytest rc=$?"
tail -n 4 gpurun_out/r3zz/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 1
