export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out; mkdir -p $O
timeout 900 python -W ignore -m pytest tests/test_kernels_gpu.py tests/test_program_group_gpu.py tests/test_conv_unit_gpu.py -m gpu -q -x --timeout 600 -k "wgrad or group or unit" 2>&1 | tail -4 | cut -c1-300 | tee $O/r06d_tests.txt
bash tools/prof_step.sh c3 3 r06d_c3_bf16 2>&1 | head -70 | tee $O/r06d_c3_bf16_top.txt
out=$O/r06d_times.txt; : > $out
t() { echo "=== $*" | tee -a $out; env "$@" timeout 400 python -W ignore tools/step_time.py $WL 15 $DT 2>&1 | grep -a -E "STEP_TIME|Error|error" | tee -a $out; }
WL=c3; DT=; t FS_X=1
WL=c3; DT=fp32; t FS_X=1
WL=c5; DT=; t FS_X=1
WL=c5; DT=fp32; t FS_X=1
WL=c4; DT=; t FS_X=1
WL=c4; DT=fp32; t FS_X=1
