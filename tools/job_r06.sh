export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out; mkdir -p $O
( time timeout 1500 python bench.py --detail $O/r06_bench_default_detail.json ) > $O/r06_bench_default.json 2> $O/r06_bench_default.err
tail -4 $O/r06_bench_default.err | cut -c1-300
wc -c $O/r06_bench_default.json
python tools/extract_bench.py $O/r06_bench_default.json 2>&1 | head -30
timeout 2400 python -W ignore -m pytest tests/ -m gpu -q --timeout 900 2>&1 | grep -v "not found in latency" | tail -12 | cut -c1-400 | tee $O/r06_gpu_tests.log
