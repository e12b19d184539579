#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
timeout 900 python -W ignore -m pytest tests/test_train_parity_gpu.py -q --timeout 600 -k "supernet" > $O/r04n_parity.log 2>&1; tail -30 $O/r04n_parity.log | cut -c1-250
