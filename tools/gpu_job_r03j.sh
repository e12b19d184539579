#!/bin/bash
# which part of bench.py's process state slows the C3 step (8-10 %)?  the supernet before / after the C2 phase, and with a small CPU thread pool
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
F="--workloads c3 --no-cpu-baseline --no-class-map --no-fp32-leg --no-roofline --steps 300 --warmup 20"
FS_BENCH_ORDER=c3,c2 timeout 300 python bench.py $F > $O/r03j_c3first.json 2>/dev/null; python tools/extract_bench.py $O/r03j_c3first.json
FS_BENCH_ORDER=c2,c3 timeout 300 python bench.py $F > $O/r03j_c2first.json 2>/dev/null; python tools/extract_bench.py $O/r03j_c2first.json
OMP_NUM_THREADS=16 FS_BENCH_ORDER=c2,c3 timeout 300 python bench.py $F > $O/r03j_c2first_omp16.json 2>/dev/null; python tools/extract_bench.py $O/r03j_c2first_omp16.json
