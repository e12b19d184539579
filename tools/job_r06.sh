export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out; mkdir -p $O
timeout 1500 python -W ignore -m pytest tests/test_program_group_gpu.py tests/test_train_steps_gpu.py tests/test_train_parity_gpu.py -m gpu -q -x --timeout 600 2>&1 | grep -v "not found in lat" | tail -5 | cut -c1-400 | tee $O/r06w_tests.txt
out=$O/r06w_times.txt; : > $out
t() { echo "=== $WL $DT $*" | tee -a $out; env "$@" timeout 400 python -W ignore tools/step_time.py $WL 15 $DT 2>&1 | grep -a -E "STEP_TIME|Error|error" | tee -a $out; }
WL=c3; DT=fp32; t FS_X=1
WL=c3; DT=fp32; t FS_WGRAD_DEFER=0
WL=c3; DT=; t FS_X=1
WL=c3; DT=; t FS_WGRAD_DEFER=0
WL=c5; DT=fp32; t FS_X=1
WL=c5; DT=fp32; t FS_WGRAD_DEFER=0
