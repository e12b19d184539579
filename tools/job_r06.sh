export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out; mkdir -p $O
timeout 900 python -W ignore -m pytest tests/test_kernels_gpu.py -m gpu -q -x --timeout 600 -k "matrix_cores or conv2d" 2>&1 | tail -4 | cut -c1-400 | tee $O/r06o_tests.txt
out=$O/r06o_times.txt; : > $out
t() { echo "=== $WL $DT $*" | tee -a $out; env "$@" timeout 400 python -W ignore tools/step_time.py $WL 15 $DT 2>&1 | grep -a -E "STEP_TIME|Error|error" | tee -a $out; }
WL=c3; DT=fp32; t FS_X=1
WL=c3; DT=fp32; t FS_FP32_X3=0
WL=c5; DT=fp32; t FS_X=1
WL=c4; DT=fp32; t FS_X=1
FS_DTYPE=fp32 bash tools/prof_step.sh c3 3 r06o_c3_fp32 2>&1 | head -12 | tee $O/r06o_c3_fp32_top.txt
