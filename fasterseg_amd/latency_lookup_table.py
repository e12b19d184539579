"""Regenerates the per-operator latency lookup table from hipEvent timings of the HIP kernels on MI355X.

Function of the reference's latency/latency_lookup_table.py (:18-113): sweep every (operator x scale x w_in x w_out) the
search can instantiate at 1024x2048 input, plus the stem / refine ConvNorms, FeatureFusion and Head entries, under the
SAME key strings, and store a pickled dict {key: ms} in latency_lookup_table.npy (the format operations.py loads at
import).  The TensorRT/ONNX/PyCUDA mechanism is replaced by fasterseg_amd.latency.compute_latency_ms_hip.

    python -m fasterseg_amd.latency_lookup_table [--out latency_lookup_table.npy] [--quick] [--dtype bf16|fp32]

Differences by design: the operators timed are the real bilinear / two-conv-skip ones (the reference's latency/ variant
swaps in nearest-neighbour resampling and a single-conv FactorizedReduce to please TensorRT), and keys use
the reference's own `int(C*ratio)` channel naming.
"""
import argparse
import json
import os

import numpy as np
import torch

from . import functional as FN
from .operations import BasicResidual1x, BasicResidual2x, BasicResidual_downup_1x, BasicResidual_downup_2x, ConvNorm, FactorizedReduce
from .seg_oprs import FeatureFusion, Head

H, W = 1024, 2048
SCALES = [8, 16, 32]
WIDTHS = [4. / 12, 6. / 12, 8. / 12, 10. / 12, 1.]


def entries(Fch_cells=(12,), Fch_heads=(8, 12), Fch_max=12):
    """Yields (key, thunk) in the reference's sweep order."""
    res = [(BasicResidual1x, "BasicResidual1x"), (BasicResidual_downup_1x, "BasicResidual_downup_1x"),
           (BasicResidual2x, "BasicResidual2x"), (BasicResidual_downup_2x, "BasicResidual_downup_2x")]
    for Fch in Fch_cells:
        for scale in SCALES:
            h, w = H // scale, W // scale
            for w_in in WIDTHS:
                for w_out in WIDTHS:
                    for stride in ((1, 2) if scale < 32 else (1,)):
                        cin, cout = int(Fch * scale * w_in), int(Fch * scale * stride * w_out)
                        for cls, name in res:
                            yield ("%s_H%d_W%d_Cin%d_Cout%d_stride%d_dilation%d" % (name, h, w, cin, cout, stride, 1),
                                   lambda cls=cls, a=(h, w, cin, cout, 3, stride, 1, 1): cls._latency(*a))
                        yield ("FactorizedReduce_H%d_W%d_Cin%d_Cout%d_stride%d" % (h, w, cin, cout, stride),
                               lambda a=(h, w, cin, cout, stride): FactorizedReduce._latency(*a))
    for Fch in Fch_heads:
        yield ("ConvNorm_H%d_W%d_Cin%d_Cout%d_kernel%d_stride%d" % (H, W, 3, 2 * Fch * 2, 3, 2),
               lambda F=Fch: ConvNorm._latency(H, W, 3, 2 * F * 2, kernel_size=3, stride=2, padding=1))
        yield ("ConvNorm_H%d_W%d_Cin%d_Cout%d_kernel%d_stride%d" % (H // 2, W // 2, 2 * Fch * 2, 4 * Fch * 2, 3, 2),
               lambda F=Fch: ConvNorm._latency(H // 2, W // 2, 2 * F * 2, 4 * F * 2, kernel_size=3, stride=2, padding=1))
        yield ("BasicResidual2x_H%d_W%d_Cin%d_Cout%d_stride%d_dilation%d" % (H // 2, W // 2, 2 * Fch * 2, 4 * Fch * 2, 2, 1),
               lambda F=Fch: BasicResidual2x._latency(H // 2, W // 2, 2 * F * 2, 4 * F * 2, 3, 2, 1, 1))
        yield ("BasicResidual2x_H%d_W%d_Cin%d_Cout%d_stride%d_dilation%d" % (H // 4, W // 4, 4 * Fch * 2, 8 * Fch, 2, 1),
               lambda F=Fch: BasicResidual2x._latency(H // 4, W // 4, 4 * F * 2, 8 * F, 3, 2, 1, 1))
    for Fch in Fch_heads:
        yield ("ConvNorm_H%d_W%d_Cin%d_Cout%d_kernel%d_stride%d" % (H // 32, W // 32, 32 * Fch, 16 * Fch, 1, 1),
               lambda F=Fch: ConvNorm._latency(H // 32, W // 32, 32 * F, 16 * F, kernel_size=1, stride=1))
        yield ("ConvNorm_H%d_W%d_Cin%d_Cout%d_kernel%d_stride%d" % (H // 16, W // 16, 16 * Fch, 8 * Fch, 1, 1),
               lambda F=Fch: ConvNorm._latency(H // 16, W // 16, 16 * F, 8 * F, kernel_size=1, stride=1))
        for w_in in WIDTHS:
            c16, c8 = int(16 * Fch + 16 * Fch_max * w_in), int(8 * Fch + 8 * Fch_max * w_in)
            yield ("ConvNorm_H%d_W%d_Cin%d_Cout%d_kernel%d_stride%d" % (H // 16, W // 16, c16, 16 * Fch, 3, 1),
                   lambda F=Fch, c=c16: ConvNorm._latency(H // 16, W // 16, c, 16 * F, kernel_size=3, stride=1, padding=1))
            yield ("ConvNorm_H%d_W%d_Cin%d_Cout%d_kernel%d_stride%d" % (H // 8, W // 8, c8, 8 * Fch, 3, 1),
                   lambda F=Fch, c=c8: ConvNorm._latency(H // 8, W // 8, c, 8 * F, kernel_size=3, stride=1, padding=1))
        for branch in range(1, 4):
            yield ("ff_H%d_W%d_C%d" % (H // 8, W // 8, 8 * Fch * branch),
                   lambda F=Fch, b=branch: FeatureFusion._latency(H // 8, W // 8, 8 * F * b, 8 * F * b))
            yield ("head_H%d_W%d_Cin%d_Cout%d" % (H // 8, W // 8, 8 * Fch * branch, 19),
                   lambda F=Fch, b=branch: Head._latency(H // 8, W // 8, 8 * F * b, 19))


SHIPPED = {"bf16": os.path.join(os.path.dirname(os.path.abspath(__file__)), "fasterseg", "latency_lookup_table_mi355x_bf16.json")}


def load_shipped(dtype="bf16"):
    """The table this module generated on an MI355X (hipEvent timings of the HIP kernels, 667 keys = the reference's key set):
    what `operations.latency_lookup_table` should hold for a search on this hardware (search/operations.py:33-36)."""
    with open(SHIPPED[dtype]) as f:
        return json.load(f)


def generate(out="latency_lookup_table.npy", quick=False, dtype=torch.bfloat16, verbose=True):
    from . import latency
    FN.set_compute_dtype(dtype)
    budget = dict(min_calib_ms=5.0, budget_ms=20.0) if quick else dict(min_calib_ms=50.0, budget_ms=150.0)
    real = latency.compute_latency_ms_hip
    table = {}
    import fasterseg_amd.operations as ops
    import fasterseg_amd.seg_oprs as sops
    timed = lambda model, size: real(model, size, **budget)
    ops.compute_latency = sops.compute_latency = timed
    try:
        for i, (key, thunk) in enumerate(entries()):
            if key in table:
                continue
            table[key] = float(thunk())
            if verbose and i % 50 == 0:
                print(i, key, "%.4f ms" % table[key], flush=True)
    finally:
        ops.compute_latency = sops.compute_latency = real
        FN.set_compute_dtype(torch.float32)
    np.save(out, table)
    with open(os.path.splitext(out)[0] + ".json", "w") as f:
        json.dump(table, f, indent=0, sort_keys=True)
    return table


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="latency_lookup_table.npy")
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    a = ap.parse_args()
    t = generate(a.out, a.quick, torch.bfloat16 if a.dtype == "bf16" else torch.float32)
    print("entries:", len(t))
