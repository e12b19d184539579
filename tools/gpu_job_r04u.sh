#!/bin/bash
# round 4, final evidence: full GPU suite, default bench, rocprofv3 kernel tables + PMC passes
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" | head -4
timeout 1500 python -W ignore -m pytest tests -m gpu -q --timeout 900 > $O/r04u_gpu_tests.log 2>&1; tail -4 $O/r04u_gpu_tests.log | cut -c1-250
timeout 900 python bench.py --steps 20 --warmup 5 --detail $O/r04u_bench_detail.json > $O/r04u_bench.json 2> $O/r04u_bench.err; tail -c 600 $O/r04u_bench.json; echo
bash tools/prof_r04.sh steps 2>&1 | tail -8
bash tools/prof_r04.sh c2 2>&1 | tail -6
