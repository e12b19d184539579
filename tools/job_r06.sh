export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out; mkdir -p $O
timeout 900 python -W ignore -m pytest tests/test_kernels_gpu.py -m gpu -q -x --timeout 600 -k "wgrad or matrix_cores" 2>&1 | tail -4 | cut -c1-400 | tee $O/r06m_tests.txt
python tools/wgrad_micro.py 2>&1 | grep -a "M=\|weighted" | cut -c1-70 | tee $O/r06m_wgrad_micro.txt
out=$O/r06m_times.txt; : > $out
t() { echo "=== $WL $DT $*" | tee -a $out; env "$@" timeout 400 python -W ignore tools/step_time.py $WL 15 $DT 2>&1 | grep -a -E "STEP_TIME|Error|error" | tee -a $out; }
WL=c3; DT=fp32; t FS_X=1
WL=c3; DT=; t FS_X=1
WL=c5; DT=fp32; t FS_X=1
WL=c4; DT=fp32; t FS_X=1
WL=c4; DT=; t FS_X=1
