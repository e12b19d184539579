#!/bin/bash
# round 4: (1) where C4's igemm family lost time against round 3, (2) census step vs rocprofv3 table of the timed steps
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
FS_IGEMM2=0 timeout 300 python tools/census_shapes.py c4 --json $O/r04v_c4_shapes_old.json 2>&1 | grep -vi warn | head -32
timeout 300 python tools/census_shapes.py c4 --json $O/r04v_c4_shapes_new.json 2>&1 | grep -vi warn | head -32
FS_SWEEP_CFGS2=100,101,102,103,105,106 timeout 300 python tools/conv_sweep.py --dtype bf16 --set fwd,dgrad --quick --from-census $O/r04v_c4_shapes_new.json --out $O/r04v_c4_sweep_bf16.json 2>&1 | grep -E "^fwd|^dgrad"
timeout 600 python bench.py --workloads c3 --steps 20 --warmup 5 --no-cpu-baseline --detail $O/r04v_bench_c3_detail.json > $O/r04v_bench_c3.json 2>/dev/null; tail -c 700 $O/r04v_bench_c3.json; echo
bash tools/prof_step.sh c3 3 r04v_c3_bf16 2>&1 | head -12
