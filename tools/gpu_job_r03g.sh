#!/bin/bash
# round 3, seventh GPU call: the PMC passes that timed out (one eager step per pass), whole GPU suite, default bench line
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp
pmc() {  # name counters... -- command...
  name=$1; shift; ctr=""
  while [ "$1" != "--" ]; do ctr="$ctr $1"; shift; done; shift
  rm -rf /tmp/pmc_$name
  timeout 280 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/pmc_$name -o run -- "$@" > $O/r03_pmc_$name.log 2>&1
  find /tmp/pmc_$name -name "*counter_collection.csv" | head -1
}
export FS_ENGINE_PLAN=$O/r03_c2_plan_choices.json
M=$(pmc c2m SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- python $R/tools/profile_frame.py 20)
[ -n "$M" ] && python $R/tools/pmc_table.py $O/r03_c2_pmc_mfma.json m=$M | head -8
unset FS_ENGINE_PLAN
export FS_SUPERNET_GRAPHS=0 FS_PROFILE_WARMUP=0 FS_PREWARM_PROGRAMS=0
for wl in c4 c3; do
  F=$(pmc ${wl}f FETCH_SIZE -- python $R/tools/profile_step.py $wl 1)
  W=$(pmc ${wl}w WRITE_SIZE -- python $R/tools/profile_step.py $wl 1)
  M=$(pmc ${wl}m SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- python $R/tools/profile_step.py $wl 1)
  args=""; [ -n "$F" ] && args="$args f=$F"; [ -n "$W" ] && args="$args w=$W"; [ -n "$M" ] && args="$args m=$M"
  [ -n "$args" ] && python $R/tools/pmc_table.py $O/r03_${wl}_pmc.json $args | head -12
done
unset FS_SUPERNET_GRAPHS FS_PROFILE_WARMUP FS_PREWARM_PROGRAMS
cd $R
rm -f $O/parity_metrics.json
timeout 1500 python -m pytest tests -m gpu -q > $O/r03g_gpu_tests.log 2>&1; tail -4 $O/r03g_gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/r03g_smoke.log 2>&1; tail -1 $O/r03g_smoke.log
timeout 1200 python bench.py > $O/r03_bench_default.json 2> $O/r03_bench_default.err; tail -c 300 $O/r03_bench_default.err
python tools/extract_bench.py $O/r03_bench_default.json
