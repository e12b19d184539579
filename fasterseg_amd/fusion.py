"""Horizontal fusion inside the search MixedOp (reference search/model_search.py:64-78).

All five primitives of a MixedOp read the same input, and two pairs of them start with the SAME convolution geometry:

    'conv'        (BasicResidual1x,        operations.py:131-200)  conv1 3x3 stride s  C_in -> C_out   on x
    'conv_2x'     (BasicResidual2x,        :280-359)               conv1 3x3 stride s  C_in -> C_out   on x
    'conv_downup' (BasicResidual_downup_1x, :203-277)              conv1 3x3 stride 1  C_in -> C_out   on bilinear(x, 1/2)
    'conv_2x_downup' (BasicResidual_downup_2x, :362-446)           conv1 3x3 stride 1  C_in -> C_out   on bilinear(x, 1/2)

so each pair is ONE GEMM over the concatenated output channels, one BatchNorm over 2 C_out channels, one weight-gradient and
one data-gradient launch, and the two zoomed primitives share one down-sample (fasterseg_amd/program.py lowers it).  The
kernels address the two filter banks / rotated packs / gradient slices as two segments of one array (fs_conv_desc.n_seg,
n_jump, k_jump, g_jump), which needs the pair's storage to be ADJACENT.  This module arranges that:

  * colocate(model)        re-homes the BatchNorm parameters and running statistics of every fusable pair (per width of the
                           USBatchNorm2d banks, slimmable_ops.py:51-70) into one tensor [A | B], the two `num_batches_tracked`
                           counters into one int64[2].
                           state_dict keys, shapes and values are unchanged; call after .to(device), before the optimizer is built.
  * flat_order(model, params)  orders the parameters so that FlatGradientSync lays the pair's gradients out as [A | B] (it assigns
                           offsets in reverse registration order); optim.FlatSGD lays its resident packs out in flat-offset order.

Nothing here changes arithmetic: a model that was not colocated (or whose storage was moved afterwards) simply lowers to the
unfused launch programs - program.py checks adjacency before it fuses.
"""
import torch

PAIRS = ((1, 3), (2, 4))        # indices into MixedOp._ops (genotypes.PRIMITIVES order): (conv, conv_2x), (conv_downup, conv_2x_downup)


def mixed_ops(model):
    from .model_search import MixedOp
    return [m for m in model.modules() if isinstance(m, MixedOp)]


def pair_modules(mixed):
    """[(opA, opB)] of a search MixedOp whose first conv -> BN units can be fused."""
    from .operations import _Residual
    out = []
    for a, b in PAIRS:
        opa, opb = mixed._ops[a], mixed._ops[b]
        if (isinstance(opa, _Residual) and isinstance(opb, _Residual) and opa.slimmable and opb.slimmable and opa.ZOOM == opb.ZOOM
                and opa.stride == opb.stride and tuple(opa.conv1.weight.shape) == tuple(opb.conv1.weight.shape)):
            out.append((opa, opb))
    return out


def colocate(model):
    """Adjacent storage for the BatchNorm state of every fusable pair.  Returns the number of (pair, width) banks re-homed."""
    n = 0
    for mixed in mixed_ops(model):
        for opa, opb in pair_modules(mixed):
            for bna, bnb in zip(opa.bn1.bn, opb.bn1.bn):
                c = bna.num_features
                if bnb.num_features != c or bna.weight.device != bnb.weight.device:
                    continue
                arena = torch.empty(4, 2 * c, dtype=bna.weight.dtype, device=bna.weight.device)
                with torch.no_grad():
                    for row, name in enumerate(("weight", "bias")):
                        arena[row, :c].copy_(getattr(bna, name))
                        arena[row, c:].copy_(getattr(bnb, name))
                        getattr(bna, name).data = arena[row, :c]
                        getattr(bnb, name).data = arena[row, c:]
                    for row, name in ((2, "running_mean"), (3, "running_var")):
                        arena[row, :c].copy_(getattr(bna, name))
                        arena[row, c:].copy_(getattr(bnb, name))
                        bna._buffers[name] = arena[row, :c]
                        bnb._buffers[name] = arena[row, c:]
                    counters = torch.stack([bna.num_batches_tracked, bnb.num_batches_tracked])
                    bna._buffers["num_batches_tracked"] = counters[0]
                    bnb._buffers["num_batches_tracked"] = counters[1]
                n += 1
    return n


def flat_order(model, params):
    """`params` re-ordered for parallel.FlatGradientSync (offsets in REVERSE list order) so that the gradient slices of a pair are
    laid out [A | B]: conv1 filters, and per width the BatchNorm weights and biases."""
    params = list(params)
    index = {id(p): i for i, p in enumerate(params)}
    partner = {}                                    # id(A) -> B
    for mixed in mixed_ops(model):
        for opa, opb in pair_modules(mixed):
            couples = [(opa.conv1.weight, opb.conv1.weight)]
            for bna, bnb in zip(opa.bn1.bn, opb.bn1.bn):
                couples += [(bna.weight, bnb.weight), (bna.bias, bnb.bias)]
            for a, b in couples:
                if id(a) in index and id(b) in index:
                    partner[id(a)] = b
    taken = {id(b) for b in partner.values()}
    out = []
    for p in params:
        if id(p) in taken:
            continue                                 # emitted right before its A (so the reversed layout reads A, B)
        if id(p) in partner:
            out.append(partner[id(p)])
        out.append(p)
    assert len(out) == len(params)
    return out


def adjacent(a, b, nbytes=None):
    """b's storage starts right where a's ends (a, b: tensors; nbytes: size of a's block, default its own bytes)."""
    if a is None or b is None:
        return False
    nbytes = a.numel() * a.element_size() if nbytes is None else nbytes
    return a.device == b.device and b.data_ptr() == a.data_ptr() + nbytes
