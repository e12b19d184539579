"""Times fs_conv2d_fwd with every tile configuration forced (and the heuristic) on the supernet's conv shapes, fp32 and bf16;
the data behind the tile-choice and split-K rules in csrc/conv_igemm.hip.  Run on an MI355X:  python tools/conv_sweep.py"""
import os
import sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fasterseg_amd import kernels as K, _lib
lib = _lib.lib()
shapes = [(3, 96, 96, 32, 64), (3, 192, 192, 16, 32), (3, 384, 384, 8, 16), (3, 96, 96, 16, 32), (3, 192, 192, 8, 16), (3, 384, 384, 4, 8),
          (3, 64, 80, 32, 64), (3, 32, 32, 32, 64), (2, 96, 96, 28, 56), (2, 192, 192, 14, 28), (2, 384, 384, 7, 14), (3, 192, 96, 32, 64),
          # round 2: the from-down / from-keep pair of a cell evaluated as one batch (N = 6 pretrain, 4 search)
          (6, 192, 192, 16, 32), (6, 384, 384, 8, 16), (6, 192, 192, 8, 16), (6, 384, 384, 4, 8), (4, 192, 192, 14, 28), (4, 384, 384, 7, 14),
          (4, 192, 192, 7, 14), (4, 160, 128, 14, 28),
          # round 3: the fused first convs of a MixedOp pair (output channels doubled) and their data gradients (input channels doubled)
          (6, 384, 768, 8, 16), (6, 192, 384, 16, 32), (3, 96, 192, 32, 64), (6, 192, 384, 8, 16), (6, 384, 768, 4, 8), (3, 192, 384, 16, 32),
          (6, 768, 384, 8, 16), (6, 384, 192, 16, 32), (3, 192, 96, 32, 64), (6, 768, 384, 4, 8)]
if os.environ.get("FS_SWEEP_ONLY_R3"):
    shapes = shapes[-10:] + [(6, 384, 384, 8, 16), (6, 192, 192, 16, 32), (3, 96, 96, 32, 64)]
CFGS = tuple(int(c) for c in os.environ["FS_SWEEP_CFGS"].split(",")) if os.environ.get("FS_SWEEP_CFGS") else (-1, 1, 2, 3, 4, 5, 6, 7)
DTYPES = [torch.bfloat16] if os.environ.get("FS_SWEEP_DTYPE") == "bf16" else [torch.float32, torch.bfloat16]
def bench(fn, iters=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(20): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters // 20): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for dtype in DTYPES:
    print("====", dtype)
    for (N, cin, cout, H, W) in shapes:
        x = K.to_nhwc(torch.randn(N, cin, H, W, device="cuda"), dtype)
        w = K.pack_weight(torch.randn(cout, cin, 3, 3, device="cuda") * 0.05, dtype)
        out = K.empty_nhwc(N, cout, H, W, dtype, "cuda")
        stats = torch.zeros(2 * cout, device="cuda")
        res = []
        for cfg in CFGS:
            lib.fs_debug_force_conv_cfg(cfg)
            try:
                t = bench(lambda: K.conv2d(x, w, cout, 3, 3, 1, 1, out=out, stats=stats))
            except Exception as e:
                t = float('nan')
            res.append(t)
        lib.fs_debug_force_conv_cfg(-1)
        gf = 2 * N * H * W * cout * cin * 9 / 1e9
        print("N%d %3d->%3d @%2dx%2d  M=%5d K=%4d %.2fGF | " % (N, cin, cout, H, W, N * H * W, cin * 9, gf) +
              " ".join("%s %.1f" % ("heur" if c < 0 else "cfg%d" % c, t) for c, t in zip(CFGS, res)) + " us")
