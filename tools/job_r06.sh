# The round's scratch GPU job (rewritten per gpurun call during round 6; this is its last form: the final check).
#   gpurun --timeout 4500 -- 'bash tools/job_r06.sh'
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $O/r06_smoke.txt
( time timeout 1500 python bench.py --detail $O/r06_bench_default_detail.json ) > $O/r06_bench_default.json 2> $O/r06_bench_default.err
tail -4 $O/r06_bench_default.err | cut -c1-300
python tools/extract_bench.py $O/r06_bench_default.json 2>&1 | head -5
timeout 2400 python -W ignore -m pytest tests/ -m gpu -q --timeout 900 2>&1 | grep -v "not found in latency" | tail -8 | cut -c1-400 | tee $O/r06_gpu_tests.log
