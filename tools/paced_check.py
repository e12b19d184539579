"""conv_igemm2 measurement variants (codes 130-133 = ABL 4 of csrc/conv_igemm2.hip) against the production
kernel (100, 104, 105, 103): same bits, and the time per launch.  python tools/paced_check.py [fp32]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fasterseg_amd import _lib, kernels as K
from tools.conv_sweep import bench
lib = _lib.lib()
dtype = torch.float32 if "fp32" in sys.argv[1:] else torch.bfloat16
SHAPES = [(6, 384, 384, 8, 16), (6, 192, 192, 16, 32), (6, 96, 96, 32, 64), (6, 192, 384, 16, 32), (3, 96, 192, 32, 64), (12, 64, 64, 64, 128), (6, 384, 768, 8, 16)]
PAIRS = [(100, 130), (104, 131), (105, 132), (103, 133)]
for (N, cin, cout, H, W) in SHAPES:
    x = K.to_nhwc(torch.randn(N, cin, H, W, device="cuda"), dtype)
    w = K.pack_weight(torch.randn(cout, cin, 3, 3, device="cuda") * 0.05, dtype)
    out = K.empty_nhwc(N, cout, H, W, dtype, "cuda")
    line = "N%d %d->%d @%dx%d:" % (N, cin, cout, H, W)
    for a, b in PAIRS:
        ys = []
        for code in (a, b):
            lib.fs_debug_force_conv_cfg(code)
            out.zero_()
            K.conv2d(x, w, cout, 3, 3, 1, 1, out=out)
            torch.cuda.synchronize()
            ys.append(out.clone())
        same = torch.equal(ys[0], ys[1])
        ts = []
        for code in (a, b):
            lib.fs_debug_force_conv_cfg(code)
            ts.append(bench(lambda: K.conv2d(x, w, cout, 3, 3, 1, 1, out=out), iters=60))
        line += "  %d/%d %s %.1f->%.1f" % (a, b, "same" if same else "DIFF", ts[0], ts[1])
    lib.fs_debug_force_conv_cfg(-1)
    print(line, flush=True)
