export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out; mkdir -p $O
out=$O/r06i_times.txt; : > $out
t() { echo "=== $WL $DT $*" | tee -a $out; env "$@" timeout 400 python -W ignore tools/step_time.py $WL 15 $DT 2>&1 | grep -a -E "STEP_TIME|Error|error" | tee -a $out; }
WL=c3; DT=fp32; t FS_FP32_X3=1
WL=c3; DT=fp32; t FS_FP32_X3=0
WL=c5; DT=fp32; t FS_FP32_X3=1
WL=c5; DT=fp32; t FS_FP32_X3=0
WL=c4; DT=fp32; t FS_FP32_X3=1
WL=c4; DT=fp32; t FS_FP32_X3=0
timeout 1500 python -W ignore -m pytest tests/test_train_steps_gpu.py tests/test_train_parity_gpu.py tests/test_kernels_gpu.py tests/test_conv_unit_gpu.py -m gpu -q --timeout 600 2>&1 | tail -25 | cut -c1-600 | grep -v "not found in latency" | tee $O/r06i_tests.txt
