#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
true
true
timeout 900 python -W ignore -m pytest tests/test_train_parity_gpu.py -q --timeout 600 -k "trajectory" -s > $O/r04t_traj.log 2>&1; grep -E "^trajectory |passed|failed|Error|assert" $O/r04t_traj.log | head -12 | cut -c1-250
