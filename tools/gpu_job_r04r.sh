#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
t() { echo "== $*"; env "$@" timeout 900 python -W ignore -m pytest tests/test_eval_path.py tests/test_train_steps_gpu.py -q --timeout 600 -m gpu -x 2>&1 | grep -E "passed|failed|Segmentation" | head -2; }
t FS_IGEMM2=0 FS_GROUP_PROGRAMS=0
t FS_GROUP_CAPTURE=2
echo "== engine class map test only + train_steps"; timeout 900 python -W ignore -m pytest tests/test_eval_path.py tests/test_train_steps_gpu.py -q --timeout 600 -m gpu -x -k "seg_evaluator or graphed_supernet" 2>&1 | grep -E "passed|failed|Segmentation" | head -2
echo "== hist + train_steps"; timeout 900 python -W ignore -m pytest tests/test_eval_path.py tests/test_train_steps_gpu.py -q --timeout 600 -m gpu -x -k "hist_info or graphed_supernet" 2>&1 | grep -E "passed|failed|Segmentation" | head -2
