export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out; mkdir -p $O
out=$O/r06x_kernarg.txt; : > $out
t() { echo "=== $WL $DT $*" | tee -a $out; env "$@" timeout 400 python -W ignore tools/step_time.py $WL 15 $DT 2>&1 | grep -a -E "STEP_TIME|Error|error" | tee -a $out; }
WL=c3; DT=fp32; t FS_X=1
WL=c3; DT=fp32; t HIP_FORCE_DEV_KERNARG=1
WL=c3; DT=fp32; t HIP_FORCE_DEV_KERNARG=0
WL=c3; DT=; t HIP_FORCE_DEV_KERNARG=1
WL=c3; DT=; t HIP_FORCE_DEV_KERNARG=0
WL=c5; DT=fp32; t HIP_FORCE_DEV_KERNARG=1
WL=c5; DT=fp32; t HIP_FORCE_DEV_KERNARG=0
