export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out; mkdir -p $O
timeout 1500 python -W ignore -m pytest tests/test_kernels_gpu.py tests/test_bn_group_gpu.py tests/test_ops_gpu.py tests/test_losses_gpu.py tests/test_engine_gpu.py tests/test_program_group_gpu.py tests/test_eval_path.py -m gpu -q -x --timeout 600 2>&1 | tail -6 | cut -c1-400 | tee $O/r06u_tests.txt
out=$O/r06u_times.txt; : > $out
t() { echo "=== $WL $DT $*" | tee -a $out; env "$@" timeout 400 python -W ignore tools/step_time.py $WL 15 $DT 2>&1 | grep -a -E "STEP_TIME|Error|error" | tee -a $out; }
WL=c3; DT=fp32; t FS_X=1
WL=c3; DT=; t FS_X=1
WL=c5; DT=fp32; t FS_X=1
WL=c5; DT=; t FS_X=1
WL=c4; DT=fp32; t FS_X=1
WL=c4; DT=; t FS_X=1
timeout 400 python bench.py --workloads c2 --no-cpu-baseline 2>/dev/null | python tools/extract_bench.py /dev/stdin 2>&1 | head -3 | tee -a $out
