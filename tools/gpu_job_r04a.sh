#!/bin/bash
# round 4, call a: conv_igemm2 (LDS-DMA implicit GEMM) - kernel tests, configuration sweep, C3 step with and without it
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" | head -4
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -rf --timeout 120 -k "conv2d_fwd or dgrad or two_segment or wider_pack or split_k or into_channel" > $O/r04a_kernels.log 2>&1
tail -25 $O/r04a_kernels.log | cut -c1-300
timeout 420 python tools/conv_sweep.py --dtype bf16 --out $O/r04a_sweep_bf16.json > $O/r04a_sweep_bf16.log 2>&1; tail -5 $O/r04a_sweep_bf16.log | cut -c1-250
timeout 240 python tools/conv_sweep.py --dtype fp32 --quick --set fwd,s2 --out $O/r04a_sweep_fp32.json > $O/r04a_sweep_fp32.log 2>&1; tail -3 $O/r04a_sweep_fp32.log | cut -c1-250
FS_IGEMM2=0 timeout 200 python tools/step_time.py c3 10 2>&1 | grep STEP_TIME
FS_IGEMM2=1 timeout 200 python tools/step_time.py c3 10 2>&1 | grep STEP_TIME
FS_IGEMM2=1 timeout 200 python tools/step_time.py c3 10 fp32 2>&1 | grep STEP_TIME
