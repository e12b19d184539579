export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out; mkdir -p $O
timeout 1500 python -W ignore -m pytest tests/test_train_steps_gpu.py tests/test_train_parity_gpu.py -m gpu -q --timeout 600 -k "dp_overlap or joint or l16 or graphed" 2>&1 | grep -v "not found in latency" | tail -25 | cut -c1-500 | tee $O/r06t_tests.txt
out=$O/r06t_times.txt; : > $out
t() { echo "=== $WL $DT $*" | tee -a $out; env "$@" timeout 400 python -W ignore tools/step_time.py $WL 15 $DT 2>&1 | grep -a -E "STEP_TIME|Error|error" | tee -a $out; }
WL=c3; DT=fp32; t FS_X=1
WL=c3; DT=fp32; t FS_TAIL_BATCH=0 FS_STEM_SHARE=0
WL=c3; DT=fp32; t FS_STEM_SHARE=0
WL=c3; DT=; t FS_X=1
WL=c3; DT=; t FS_TAIL_BATCH=0 FS_STEM_SHARE=0
WL=c5; DT=fp32; t FS_X=1
WL=c5; DT=fp32; t FS_TAIL_BATCH=0 FS_STEM_SHARE=0
WL=c5; DT=; t FS_X=1
