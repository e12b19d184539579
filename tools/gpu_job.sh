#!/bin/bash
# One parameterised GPU job script (replaces the one-shot tools/gpu_job_r0*.sh of rounds 3-4).  Run through gpurun from the repo root:
#   gpurun --timeout 900 -- 'bash tools/gpu_job.sh <job> [args]'
# Everything a job writes goes to gpurun_out/ (merged back by gpurun); summaries worth keeping are copied to profiles/ by hand.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
mkdir -p $O
job=${1:-help}; shift
case $job in
  probe)      # grid-barrier probe + supernet step time against the number of hardware queues the runtime may use
    timeout 300 tools/probes/grid_barrier.bin > $O/r05_grid_barrier.txt 2>&1; tail -5 $O/r05_grid_barrier.txt
    for q in default 2 8; do
      if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
      timeout 300 python -W ignore tools/step_time.py c3 20 2>&1 | grep STEP_TIME | sed "s/^/hwq=$q /" | tee -a $O/r05_hwq_c3.txt
    done
    unset GPU_MAX_HW_QUEUES
    ;;
  sweep)      # bash tools/gpu_job.sh sweep <c3|c5> "ENV=.. ENV=.." "ENV=.." ...: host_vs_device of one workload under each environment, one box
    wl=$1; shift
    for env in "$@"; do
      env $env timeout 300 python -W ignore tools/host_vs_device.py $wl 10 2>&1 | grep -a -A3 "HOST_VS" | grep -v synchronised | sed "s/^/$env /" | tee -a $O/r05_sweep_$wl.txt
    done
    ;;
  final)      # the round's evidence on ONE box: whole GPU suite, the default bench line, rocprofv3 kernel tables of the timed steps
    tag=${1:-r05}
    timeout 1500 python -W ignore -m pytest tests -m gpu -q --timeout 900 --durations=10 > $O/${tag}_gpu_tests.log 2>&1; echo "pytest rc=$?"; tail -4 $O/${tag}_gpu_tests.log | cut -c1-200
    timeout 900 python -W ignore bench.py --steps 20 --warmup 5 --detail $O/${tag}_bench_default_detail.json > $O/${tag}_bench_default.json 2> $O/${tag}_bench_default.err; echo "bench rc=$?"
    tail -c 3200 $O/${tag}_bench_default.json; tail -2 $O/${tag}_bench_default.err
    bash tools/prof_round.sh steps $tag 2>&1 | tail -6
    ;;
  tests)      # bash tools/gpu_job.sh tests <pytest args...>
    timeout ${FS_JOB_TIMEOUT:-600} python -W ignore -m pytest "$@" -m gpu -q -x --timeout 300 --durations=8 2>&1 | tail -25 | cut -c1-240
    ;;
  step)       # bash tools/gpu_job.sh step c3|c4|c5 [steps] [fp32]
    timeout 400 python -W ignore tools/step_time.py "$@" 2>&1 | tail -3
    ;;
  bench)      # bash tools/gpu_job.sh bench <name> [bench.py args...]  -> gpurun_out/<name>.json + <name>_detail.json
    name=$1; shift
    timeout 900 python -W ignore bench.py --detail $O/${name}_detail.json "$@" > $O/$name.json 2> $O/$name.err; echo rc=$?; tail -c 1500 $O/$name.json; tail -3 $O/$name.err
    ;;
  prof)       # bash tools/gpu_job.sh prof c3|c4|c5 <steps> <name>: rocprofv3 kernel table of the timed steps only
    bash tools/prof_step.sh "$@" 2>&1 | head -40
    ;;
  *)
    echo "jobs: probe | tests <pytest args> | step <c3|c4|c5> [steps] [fp32] | bench <name> [args] | prof <wl> <steps> <name>"
    ;;
esac
