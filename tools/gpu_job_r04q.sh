#!/bin/bash
# round 4: full GPU suite + default bench on the current tree
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
timeout 1500 python -W ignore -m pytest tests -m gpu -q --timeout 900 > $O/r04q_gpu_tests.log 2>&1; tail -8 $O/r04q_gpu_tests.log | cut -c1-250
timeout 900 python bench.py --steps 20 --warmup 5 --detail $O/r04q_bench_detail.json > $O/r04q_bench.json 2> $O/r04q_bench.err; tail -c 3500 $O/r04q_bench.json
