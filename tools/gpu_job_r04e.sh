#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
FS_SWEEP_CFGS2=100,103,104,105,106 timeout 600 python tools/conv_sweep.py --dtype fp32 --set dgrad --out $O/r04e_sweep_fp32_dgrad.json > $O/r04e_sweep_fp32.log 2>&1; tail -2 $O/r04e_sweep_fp32.log | cut -c1-200
