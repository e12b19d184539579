export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out; mkdir -p $O
out=$O/r06r_sweeps.txt; : > $out
t() { echo "=== $WL $DT $*" | tee -a $out; env "$@" timeout 400 python -W ignore tools/step_time.py $WL 12 $DT 2>&1 | grep -a -E "STEP_TIME|Error|error" | tee -a $out; }
WL=c3; DT=fp32
t FS_X=1
t FS_IGEMM2_GROUP_CFG=0
t FS_IGEMM2_GROUP_CFG=4
t FS_IGEMM2_GROUP_CFG=5
t FS_IGEMM2_GROUP_CFG=6
t FS_WGRAD_GROUP_BLOCKS=256
t FS_WGRAD_GROUP_BLOCKS=384
t FS_WGRAD_GROUP_BLOCKS=768
t FS_WGRAD_GROUP_BLOCKS=1024
t FS_IGEMM2_GROUP_FIXED=3000
t FS_IGEMM2_GROUP_FIXED=6000
