export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out; mkdir -p $O
( time timeout 1500 python bench.py --detail $O/r06q_bench_default_detail.json ) > $O/r06q_bench_default.json 2> $O/r06q_bench_default.err
tail -4 $O/r06q_bench_default.err | cut -c1-300
cat $O/r06q_bench_default.json | cut -c1-3000
timeout 1200 python -W ignore -m pytest tests/test_parallel_gpu.py -m gpu -q --timeout 900 2>&1 | tail -5 | cut -c1-400 | tee $O/r06q_gpu_tests.txt
