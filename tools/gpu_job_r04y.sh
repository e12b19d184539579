#!/bin/bash
# round 4: the GPU tests that touch what changed after the full-suite run (r04u): statistics switch, big-map routing, census mimic, LUT journal
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 235 python -W ignore -m pytest tests/test_latency_lut.py tests/test_train_steps_gpu.py tests/test_parallel_gpu.py tests/test_ops_gpu.py tests/test_losses_gpu.py tests/test_supernet.py -m gpu -q -x --timeout 200 --durations=8 2>&1 | tail -16 | cut -c1-220
