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
  call2)      # round 5, second call: barrier probe v2, ATen call sites of a C5 step, the tests that touch this round's host-side changes,
              # C3 under fewer hardware queues, and the in-flight histogram of a C3 step
    timeout 200 tools/probes/grid_barrier.bin > $O/r05_grid_barrier_v2.txt 2>&1; tail -3 $O/r05_grid_barrier_v2.txt
    timeout 400 python -W ignore tools/aten_sites.py c5 $O/r05_c5_aten_sites.json > $O/r05_c5_aten_sites.txt 2>&1; head -45 $O/r05_c5_aten_sites.txt
    timeout 600 python -W ignore -m pytest tests/test_program_group_gpu.py tests/test_supernet.py tests/test_train_steps_gpu.py tests/test_train_parity_gpu.py \
        "tests/test_kernels_gpu.py::test_conv2d_fwd" -m gpu -q -x --timeout 300 --durations=5 2>&1 | tail -15 | cut -c1-220
    for q in 2 1; do
      GPU_MAX_HW_QUEUES=$q timeout 300 python -W ignore tools/step_time.py c3 20 2>&1 | grep -a "STEP_TIME\|Error\|error" | head -3 | sed "s/^/hwq=$q /" | tee -a $O/r05_hwq_c3.txt
    done
    bash tools/prof_step.sh c3 3 r05_c3_probe 2>&1 | head -12
    python tools/trace_concurrency.py $(find /tmp/prof_step -name "*kernel_trace.csv" | head -1) $O/prof_step_r05_c3_probe.log 3 $O/r05_c3_concurrency.json
    ;;
  call3)      # round 5, third call: the tests that touch the BN kernels / zero arenas / DP changes, step times, and the C2 part of the bench line
    timeout 900 python -W ignore -m pytest tests/test_bn_group_gpu.py tests/test_conv_unit_gpu.py tests/test_train_steps_gpu.py tests/test_parallel_gpu.py \
        tests/test_engine_gpu.py tests/test_ops_gpu.py tests/test_supernet.py tests/test_program_group_gpu.py \
        "tests/test_kernels_gpu.py::test_batchnorm_train_fwd_bwd" "tests/test_kernels_gpu.py::test_bn_finalize_counter_and_fused_param_grad_accumulation" -m gpu -q -x --timeout 600 --durations=6 -s 2>&1 | grep -v "^$" | tail -30 | cut -c1-260
    for wl in c3 c5; do timeout 400 python -W ignore tools/step_time.py $wl 20 2>&1 | grep -a STEP_TIME | tee -a $O/r05_step_times.txt; done
    timeout 600 python -W ignore bench.py --workloads c2 --steps 20 --warmup 5 --detail $O/r05_bench_c2_detail.json > $O/r05_bench_c2.json 2> $O/r05_bench_c2.err; echo rc=$?
    tail -c 2500 $O/r05_bench_c2.json; tail -3 $O/r05_bench_c2.err
    ;;
  call4)      # round 5: what bounds the C3 step - host issue or device execution - and what the launch-grouping switches do on ONE box
    timeout 300 python -W ignore tools/host_vs_device.py c3 10 2>&1 | grep -a -A4 HOST_VS | tee $O/r05_host_vs_device.txt
    for env in "FS_NONE=1" "FS_GROUP_CAPTURE=2" "FS_GROUP_PROGRAMS=0" "FS_EAGER_LANES=1" "FS_JOIN_FR=0" "FS_LAYER_LANES=1"; do
      env $env timeout 300 python -W ignore tools/step_time.py c3 20 2>&1 | grep -a STEP_TIME | sed "s/^/$env /" | tee -a $O/r05_switches_c3.txt
    done
    ;;
  call5)      # round 5: the host side of the C3 step (it is host-bound: call4) - cProfile, and the capture without forks
    timeout 300 python -W ignore tools/host_profile.py c3 5 > $O/r05_host_profile_c3.txt 2>&1; grep -a -A40 "sorted by tottime" $O/r05_host_profile_c3.txt | cut -c1-180
    for env in "FS_LAYER_LANES=1 FS_BRANCH_LANES=1" "FS_LAYER_LANES=1 FS_BRANCH_LANES=1 FS_EAGER_LANES=1"; do
      env $env timeout 300 python -W ignore tools/host_vs_device.py c3 10 2>&1 | grep -a -A3 HOST_VS | sed "s/^/$env /" | tee -a $O/r05_host_vs_device.txt
    done
    ;;
  call6)      # round 5: after the beta-table read left the eager forwards - the capture layouts against each other (host issue vs device)
    for env in "FS_NONE=1" "FS_LAYER_LANES=2" "FS_LAYER_LANES=2 FS_GROUP_CAPTURE=2" "FS_LAYER_LANES=1 FS_BRANCH_LANES=1" "FS_LAYER_LANES=2 FS_EAGER_LANES=1"; do
      env $env timeout 300 python -W ignore tools/host_vs_device.py c3 10 2>&1 | grep -a -A3 HOST_VS | grep -v synchronised | sed "s/^/$env /" | tee -a $O/r05_host_vs_device_2.txt
    done
    ;;
  sweep)      # bash tools/gpu_job.sh sweep <c3|c5> "ENV=.. ENV=.." "ENV=.." ...: host_vs_device of one workload under each environment, one box
    wl=$1; shift
    for env in "$@"; do
      env $env timeout 300 python -W ignore tools/host_vs_device.py $wl 10 2>&1 | grep -a -A3 "HOST_VS" | grep -v synchronised | sed "s/^/$env /" | tee -a $O/r05_sweep_$wl.txt
    done
    ;;
  call9)      # round 5: supernet tests after the host-side changes, then step times of C3 / C5 and the C2 frame's host share
    timeout 900 python -W ignore -m pytest tests/test_train_steps_gpu.py tests/test_program_group_gpu.py tests/test_supernet.py tests/test_train_parity_gpu.py \
        -m gpu -q -x --timeout 600 --durations=4 2>&1 | grep -v "^$" | tail -12 | cut -c1-250
    for wl in c3 c5 c2; do timeout 400 python -W ignore tools/host_vs_device.py $wl 10 2>&1 | grep -a -A3 "HOST_VS" | grep -v synchronised | tee -a $O/r05_host_vs_device_3.txt; done
    for wl in c3 c5; do timeout 400 python -W ignore tools/step_time.py $wl 20 2>&1 | grep -a STEP_TIME | tee -a $O/r05_step_times.txt; done
    ;;
  call10)     # round 5: the search step after the side-stream width read and the fast phase flips
    timeout 900 python -W ignore -m pytest tests/test_train_steps_gpu.py tests/test_parallel_gpu.py tests/test_train_parity_gpu.py tests/test_supernet.py \
        -m gpu -q -x --timeout 600 -k "search or arch or latency or supernet" --durations=4 2>&1 | grep -v "^$" | tail -12 | cut -c1-250
    timeout 400 python -W ignore tools/host_vs_device.py c5 10 2>&1 | grep -a -A3 "HOST_VS" | grep -v synchronised | tee -a $O/r05_host_vs_device_3.txt
    timeout 400 python -W ignore tools/host_profile.py c5 5 > $O/r05_host_profile_c5.txt 2>&1; grep -a -A28 "sorted by tottime" $O/r05_host_profile_c5.txt | cut -c1-160
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
