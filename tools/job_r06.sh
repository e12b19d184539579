export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out; mkdir -p $O
timeout 1500 python -W ignore -m pytest tests/test_kernels_gpu.py tests/test_ops_gpu.py tests/test_program_group_gpu.py tests/test_conv_unit_gpu.py tests/test_train_steps_gpu.py tests/test_train_parity_gpu.py tests/test_engine_gpu.py -m gpu -q --timeout 600 2>&1 | tail -25 | cut -c1-400 | tee $O/r06f_tests.txt
out=$O/r06f_times.txt; : > $out
t() { echo "=== $WL $DT $*" | tee -a $out; env "$@" timeout 400 python -W ignore tools/step_time.py $WL 15 $DT 2>&1 | grep -a -E "STEP_TIME|Error|error" | tee -a $out; }
WL=c3; DT=fp32; t FS_FP32_X3=1
WL=c3; DT=fp32; t FS_FP32_X3=0
WL=c3; DT=fp32; t FS_FP32_X3=1 FS_PAIR_DIRECT=0 FS_MERGE_GROUP=0
WL=c3; DT=; t FS_X=1
WL=c3; DT=; t FS_PAIR_DIRECT=0 FS_MERGE_GROUP=0
WL=c5; DT=fp32; t FS_FP32_X3=1
WL=c5; DT=fp32; t FS_FP32_X3=0
WL=c4; DT=fp32; t FS_FP32_X3=1
WL=c4; DT=fp32; t FS_FP32_X3=0
