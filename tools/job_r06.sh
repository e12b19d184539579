export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $O/r06_smoke.txt
timeout 2400 python -W ignore -m pytest tests/ -m gpu -q --timeout 900 2>&1 | grep -v "not found in latency" | tail -8 | cut -c1-400 | tee $O/r06_gpu_tests.log
