#!/bin/bash
# round 4, call b: conv_igemm2 with precomputed DMA offsets + register double-buffered fragments; forced-slice codes now live
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -rf --timeout 120 -k "conv2d_fwd or dgrad or two_segment or wider_pack or split_k or into_channel" > $O/r04b_kernels.log 2>&1
tail -12 $O/r04b_kernels.log | cut -c1-250
timeout 500 python tools/conv_sweep.py --dtype bf16 --set fwd,dgrad,s2 --out $O/r04b_sweep_bf16.json > $O/r04b_sweep_bf16.log 2>&1; tail -2 $O/r04b_sweep_bf16.log | cut -c1-250
timeout 300 python tools/conv_sweep.py --dtype bf16 --quick --set unit --out $O/r04b_sweep_unit_bf16.json > $O/r04b_sweep_unit_bf16.log 2>&1; tail -2 $O/r04b_sweep_unit_bf16.log | cut -c1-250
timeout 300 python tools/conv_sweep.py --dtype fp32 --quick --set fwd,dgrad,s2 --out $O/r04b_sweep_fp32.json > $O/r04b_sweep_fp32.log 2>&1; tail -2 $O/r04b_sweep_fp32.log | cut -c1-250
