"""TEST INFRASTRUCTURE — deterministic, construction-order-independent parameter fill.

Trained FasterSeg weights are not available offline (README.md:122 links Google Drive), so parity
uses seeded random weights.  Each tensor is generated from its *state_dict key* and shape only, so
the unmodified reference network (oracle/make_golden.py), the oracle (oracle/ref_ops.py) and the HIP
product path receive bit-identical parameters as long as key names and shapes agree — which is itself
part of the drop-in contract.

Conv weights follow init_weight's kaiming_normal_(fan_in, relu) scale (tools/utils/init_func.py:5-14,
train/train.py:122); BN affine and running statistics are made non-trivial so eval-mode BN is exercised.
"""
import zlib

import torch


def _gen(key, seed):
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(key.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    return g


def fill_like(key, ref, seed=0):
    """Tensor for state_dict entry ``key`` with the shape/dtype of ``ref``."""
    shape = tuple(ref.shape)
    g = _gen(key, seed)
    leaf = key.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return torch.zeros(shape, dtype=ref.dtype)
    if len(shape) == 4:                                  # conv weight, OIHW
        fan_in = shape[1] * shape[2] * shape[3]
        return torch.randn(shape, generator=g) * (2.0 / fan_in) ** 0.5
    if leaf == "running_var":
        return torch.rand(shape, generator=g) + 0.5
    if leaf == "running_mean":
        return torch.randn(shape, generator=g) * 0.1
    if leaf == "weight":                                 # BN gamma
        return torch.rand(shape, generator=g) + 0.5
    if leaf == "bias":                                   # BN beta / classifier bias
        return torch.randn(shape, generator=g) * 0.1
    return torch.randn(shape, generator=g) * 0.1


def seeded_state(template, seed=0):
    """{key: tensor} for every entry of ``template`` (a state_dict or {key: tensor/shape-holder})."""
    return {k: fill_like(k, v, seed) for k, v in template.items()}


def seeded_input(shape, seed=0):
    g = torch.Generator(device="cpu")
    g.manual_seed(1000003 + seed)
    return torch.randn(shape, generator=g)


def resolve_aliases(params, meta):
    """Shared cells appear under several state_dict keys (train/model_seg.py:293-294, e.g. cells.0-0.* and
    cells.0-1.* are ONE module).  nn.Module.load_state_dict writes every key, so the last alias wins; give
    all aliases that value so key-addressed consumers (the oracle) see what the module holds."""
    out = dict(params)
    for layer, groups in enumerate(meta["branch_groups"]):
        for group in groups:
            if len(group) < 2:
                continue
            src = "cells.%d-%d." % (layer, group[-1])
            for b in group[:-1]:
                dst = "cells.%d-%d." % (layer, b)
                for k in list(out):
                    if k.startswith(src):
                        out[dst + k[len(src):]] = out[k]
    return out
