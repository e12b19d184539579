#!/bin/bash
# round 3, sixth GPU call: why is the C3 step slower inside bench.py than alone?  + the round's profiles
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
mkdir -p $O
timeout 300 python tools/step_time.py c3 10 2>&1 | grep STEP_TIME
timeout 600 python bench.py --workloads c3 --steps 200 --warmup 20 --no-cpu-baseline --no-class-map --no-fp32-leg --no-roofline > $O/r03f_bench_c3only.json 2>/dev/null; python tools/extract_bench.py $O/r03f_bench_c3only.json

bash tools/prof_r03.sh all 2>&1 | tail -60
