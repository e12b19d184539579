"""One conv geometry / configuration, launched many times back to back: the target of rocprofv3 --pmc passes and kernel traces.

    python tools/conv_one.py N cin cout H W code [dtype] [mode fwd|dgrad] [iters]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fasterseg_amd import _lib, kernels as K  # noqa: E402

N, cin, cout, H, W, code = (int(v) for v in sys.argv[1:7])
dtype = torch.float32 if (len(sys.argv) > 7 and sys.argv[7] == "fp32") else torch.bfloat16
mode = sys.argv[8] if len(sys.argv) > 8 else "dgrad"
iters = int(sys.argv[9]) if len(sys.argv) > 9 else 50
lib = _lib.lib()
x = K.to_nhwc(torch.randn(N, cin, H, W, device="cuda"), dtype)
if mode == "fwd":
    w = K.pack_weight(torch.randn(cout, cin, 3, 3, device="cuda") * 0.05, dtype)
    out = K.empty_nhwc(N, cout, H, W, dtype, "cuda")
    stats = torch.zeros(2 * cout, device="cuda")
    fn = lambda: K.conv2d(x, w, cout, 3, 3, 1, 1, out=out, stats=stats)
else:       # plain conv without the statistics epilogue (what a data gradient launches)
    w = K.pack_weight(torch.randn(cout, cin, 3, 3, device="cuda") * 0.05, dtype)
    out = K.empty_nhwc(N, cout, H, W, dtype, "cuda")
    fn = lambda: K.conv2d(x, w, cout, 3, 3, 1, 1, out=out)
lib.fs_debug_force_conv_cfg(code)
for _ in range(iters):
    fn()
torch.cuda.synchronize()
print("done", N, cin, cout, H, W, code)
