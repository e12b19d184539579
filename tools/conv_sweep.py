"""Times fs_conv2d_fwd on the supernet's / student's small-map conv shapes with every tile configuration forced - the data behind the
tile-choice and split-K rules in csrc/conv_igemm.hip and csrc/conv_igemm2.hip.  Run on an MI355X:

    python tools/conv_sweep.py [--dtype bf16|fp32|both] [--set fwd,unit,dgrad,s2] [--out file.json]

  fwd    raw conv with the BN-statistics epilogue (split-K configurations pay their reduce launch here)
  unit   the train-mode conv -> BN -> ReLU unit (fs_conv_bn_act_train_fwd): what a supernet MixedOp launches; small maps sum the
         split-K slabs inside the BatchNorm kernel
  dgrad  stride-1 data gradients (the same kernel, channels swapped, two-segment contraction not modelled)
  s2     data gradients of stride-2 convs: zero-insertion (cfg -2) vs parity classes (conv_igemm2)
Codes: -2 = round-3 heuristic (conv_igemm.hip only), -1 = production heuristic, 100 + c = conv_igemm2 configuration c, 1000 * s + 100 + c =
with s K slices (1: unsplit)."""
import argparse
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fasterseg_amd import _lib, kernels as K  # noqa: E402

lib = _lib.lib()

# (N, cin, cout, H, W): supernet C3 (batch 3, pairs 6) at 1/8, 1/16, 1/32 and their zoomed halves; fused pairs double cout
FWD = [(6, 384, 768, 8, 16), (6, 384, 384, 8, 16), (6, 384, 768, 4, 8), (6, 384, 384, 4, 8), (3, 384, 384, 8, 16), (3, 384, 768, 4, 8),
       (6, 192, 384, 16, 32), (6, 192, 192, 16, 32), (6, 192, 384, 8, 16), (6, 192, 192, 8, 16), (3, 192, 384, 16, 32),
       (6, 96, 192, 32, 64), (6, 96, 96, 32, 64), (6, 96, 192, 16, 32), (6, 96, 96, 16, 32), (3, 96, 192, 32, 64),
       # min-width pass (1/3 of the channels) and a random-width one
       (6, 128, 256, 8, 16), (6, 64, 128, 16, 32), (6, 32, 64, 32, 64), (6, 32, 32, 16, 32), (6, 160, 256, 8, 16), (6, 80, 128, 32, 64),
       # C5 (batch 2, pairs 4, 224x448)
       (4, 384, 768, 7, 14), (4, 192, 384, 14, 28), (4, 96, 192, 28, 56),
       # student C2 small maps (batch 1)
       (1, 256, 256, 16, 32), (1, 128, 128, 16, 32), (1, 128, 256, 16, 32), (1, 192, 192, 32, 64), (1, 128, 128, 32, 64), (1, 64, 64, 64, 128),
       (1, 192, 128, 64, 128)]
# stride-2 forward geometries (N, cin, cout, H, W of the INPUT): their data gradients are the s2 set
S2 = [(3, 96, 384, 32, 64), (3, 192, 768, 16, 32), (6, 96, 192, 32, 64), (3, 32, 128, 32, 64)]
CFGS2 = [int(c) for c in os.environ.get("FS_SWEEP_CFGS2", "100,101,102,103,104,105,106").split(",")]
SLICES = [1, 2, 4, 8]


def bench(fn, iters=100):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(20):
                fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters // 20):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def codes(full):
    out = [-2, -1]
    for c in CFGS2:
        out.append(c)
        if full:
            out.extend(1000 * s + c for s in SLICES)
    return out


def run_codes(fn, code_list):
    res = {}
    for code in code_list:
        lib.fs_debug_force_conv_cfg(code)
        try:
            res[code] = round(bench(fn), 2)
        except Exception as e:                                        # noqa: BLE001
            res[code] = None
            print("   code", code, "failed:", repr(e)[:100])
    lib.fs_debug_force_conv_cfg(-1)
    return res


def fmt(res):
    best = min((v, k) for k, v in res.items() if v is not None and k >= 100)
    return "old %.1f  heur %.1f  best %d: %.1f | " % (res[-2], res[-1], best[1], best[0]) + " ".join(
        "%d:%.1f" % (k, v) for k, v in res.items() if k >= 100 and v is not None and v <= 1.15 * best[0])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="both")
    ap.add_argument("--set", default="fwd,unit,dgrad,s2")
    ap.add_argument("--out", default=None)
    ap.add_argument("--quick", action="store_true", help="no forced slice counts")
    ap.add_argument("--from-census", default=None, help="geometries = the 3x3 stride-1 launches of a tools/census_shapes.py --json table (top 14 by time)")
    args = ap.parse_args()
    global FWD, S2
    if args.from_census:
        import re
        table = json.load(open(args.from_census))
        geo = []
        for key, r in sorted(table.items(), key=lambda kv: -kv[1]["ms"]):
            m = re.match(r"N(\d+) (\d+)x(\d+) (\d+)->(\d+) k(\d+) s(\d+) fl([0-9a-f]+)", key)
            n, h, w, cin, cout, k, st, fl = [int(v, 16) if i == 7 else int(v) for i, v in enumerate(m.groups())]
            if k == 3 and st == 1 and not fl & 0x2 and cin % 8 == 0 and (n, cin, cout, h, w) not in geo:
                geo.append((n, cin, cout, h, w))
        FWD, S2 = geo[:14], []
    dtypes = {"bf16": [torch.bfloat16], "fp32": [torch.float32], "both": [torch.bfloat16, torch.float32]}[args.dtype]
    sets = args.set.split(",")
    results = {}
    for dtype in dtypes:
        dn = "bf16" if dtype == torch.bfloat16 else "fp32"
        print("====", dn)
        for (N, cin, cout, H, W) in FWD:
            gf = 2 * N * H * W * cout * cin * 9 / 1e9
            tag = "N%d %3d->%3d @%2dx%2d M=%5d K=%4d %.2fGF" % (N, cin, cout, H, W, N * H * W, cin * 9, gf)
            x = K.to_nhwc(torch.randn(N, cin, H, W, device="cuda"), dtype)
            w = K.pack_weight(torch.randn(cout, cin, 3, 3, device="cuda") * 0.05, dtype)
            out = K.empty_nhwc(N, cout, H, W, dtype, "cuda")
            stats = torch.zeros(2 * cout, device="cuda")
            if "fwd" in sets:
                res = run_codes(lambda: K.conv2d(x, w, cout, 3, 3, 1, 1, out=out, stats=stats), codes(not args.quick))
                results["%s fwd %s" % (dn, tag)] = res
                print("fwd  ", tag, "|", fmt(res))
            if "unit" in sets:
                gamma, beta = torch.ones(cout, device="cuda"), torch.zeros(cout, device="cuda")
                rm, rv = torch.zeros(cout, device="cuda"), torch.ones(cout, device="cuda")
                groups = 2 if N % 2 == 0 and N > 1 else 1
                d = K.ConvDesc(N, H, W, cin, cout, 3, 3, 1, 1, H, W, x.stride(3), cout, K.dtype_code(dtype), K.FS_CONV_RELU, 0, 0, 0, 0, 0, groups)
                z = K.empty_nhwc(N, cout, H, W, dtype, "cuda")
                y = K.empty_nhwc(N, cout, H, W, dtype, "cuda")
                saved = torch.empty(groups * 4 * cout, device="cuda")
                st2 = torch.zeros(groups * 2 * cout, device="cuda")
                ws, wsb = K.stream_workspace("cuda")

                def unit():
                    K.call("fs_conv_bn_act_train_fwd", K._stream(), ctypes.byref(d), x.data_ptr(), w.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                           rm.data_ptr(), rv.data_ptr(), None, 1e-5, 0.1, st2.data_ptr(), saved.data_ptr(), z.data_ptr(), y.data_ptr(), ws, wsb)
                res = run_codes(unit, codes(not args.quick))
                results["%s unit %s" % (dn, tag)] = res
                print("unit ", tag, "|", fmt(res))
            if "dgrad" in sets:
                dz = K.to_nhwc(torch.randn(N, cout, H, W, device="cuda"), dtype)
                wf = K.pack_weight(torch.randn(cout, cin, 3, 3, device="cuda") * 0.05, dtype, flip=True)
                dx = K.empty_nhwc(N, cin, H, W, dtype, "cuda")
                res = run_codes(lambda: K.conv2d(dz, wf, cin, 3, 3, 1, 1, out=dx), codes(not args.quick))
                results["%s dgrad %s" % (dn, tag)] = res
                print("dgrad", tag, "|", fmt(res))
        if "s2" in sets:
            for (N, cin, cout, H, W) in S2:
                ho, wo = H // 2, W // 2
                dz = K.to_nhwc(torch.randn(N, cout, ho, wo, device="cuda"), dtype)
                wf = K.pack_weight(torch.randn(cout, cin, 3, 3, device="cuda") * 0.05, dtype, flip=True)
                dx = K.empty_nhwc(N, cin, H, W, dtype, "cuda")
                res = run_codes(lambda: K.conv2d(dz, wf, cin, 3, 3, 1, 1, transposed=True, out_hw=(H, W), out=dx), [-2, -1] + CFGS2)
                tag = "N%d dz %d@%dx%d -> dx %d@%dx%d" % (N, cout, ho, wo, cin, H, W)
                results["%s s2 %s" % (dn, tag)] = res
                print("s2   ", tag, "|", fmt(res))
    if args.out:
        with open(args.out, "w") as f:
            json.dump({k: {str(c): v for c, v in r.items()} for k, r in results.items()}, f, indent=1)


if __name__ == "__main__":
    main()
