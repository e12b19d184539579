"""TEST INFRASTRUCTURE — CPU oracle for the supernet (search/model_search.py), torch fp32/fp64 on the CPU.

Functional restatement of Network_Multi_Path.forward / sample_prun_ratio / _loss over a flat ``params`` dict keyed like
the reference state_dict (the five primitives themselves are oracle/ref_ops.Ctx.primitive).  Pinned by
tests/test_oracle_supernet.py against tests/golden/supernet.npz, which oracle/make_golden.py produced from the
UNMODIFIED reference modules: eval logits of both architectures, the pretrain and search losses and sampled gradients
(same RNG call order as the reference: np.random.choice for "random" widths, torch.rand for the Gumbel noise).

Used by tests/ (parity of the HIP supernet step at sizes the fixtures do not cover) and by bench.py's cpu_baseline leg
(the reference's train step cannot travel to the GPU box).  Never imported by fasterseg_amd/.
"""
import numpy as np
import torch
import torch.nn.functional as F

from .ref_ops import PRIMITIVES, Ctx, resize


def gumbel_softmax_hard(logits):
    """search/model_search.py:14-44 (straight-through one-hot); the reference's .cuda() is a no-op here."""
    u = torch.rand(logits.size())
    y = F.softmax(logits + (-torch.log(-torch.log(u + 1e-20) + 1e-20)).to(logits.dtype), dim=-1)
    ind = y.max(dim=-1)[1]
    y_hard = torch.zeros_like(y).view(-1, y.shape[-1])
    y_hard.scatter_(1, ind.view(-1, 1), 1)
    y_hard = y_hard.view(*y.shape)
    return (y_hard - y).detach() + y


def sample_prun_ratio(params, cfg, arch_idx, mode):
    """Network_Multi_Path.sample_prun_ratio (:209-260): three lists (scales 0,1,2) of widths or one-hot score tensors."""
    layers, wml = cfg["layers"], cfg["width_mult_list"]
    counts = (layers - 1, layers - 1, layers - 2)
    if mode == "arch_ratio":
        out = []
        for s, n in enumerate(counts):
            r = params["ratio_%d_%d" % (arch_idx, s)]
            out.append([gumbel_softmax_hard(F.log_softmax(r[layer], dim=-1)) for layer in range(n)])
        return out
    if mode == "min":
        return [[wml[0]] * n for n in counts]
    if mode == "max":
        return [[wml[-1]] * n for n in counts]
    assert mode == "random"
    return [[np.random.choice(wml) for _ in range(n)] for n in counts]


def mixed_op(c, x, prefix, stride, weights, ratios):
    """MixedOp.forward (:64-78): sum_k op_k(x) * alpha_k * r_score0 * r_score1 with the widths the ratios select."""
    rr, scores = [], []
    for r in ratios:
        if torch.is_tensor(r):
            i = int(r.argmax())
            rr.append(c.wml[i]); scores.append(r[i])
        else:
            rr.append(r); scores.append(1.0)
    result = 0
    for k, kind in enumerate(PRIMITIVES):
        result = result + c.primitive(kind, x, "%s._ops.%d" % (prefix, k), stride, (rr[0], rr[1])) * weights[k] * scores[0] * scores[1]
    return result


def cell(c, x, prefix, down, alphas, ratios):
    """Cell.forward (:123-128)."""
    out = mixed_op(c, x, prefix + "._op", 1, alphas, (ratios[0], ratios[1]))
    dn = mixed_op(c, x, prefix + ".downsample", 2, alphas, (ratios[0], ratios[2])) if down else None
    return out, dn


def forward(params, cfg, x, arch_idx, prun_mode=None, training=True):
    """Network_Multi_Path.forward (:263-358) -> (pred0, pred1, pred2, pred02, pred12)."""
    layers, shw = cfg["layers"], cfg["stem_head_width"]
    c = Ctx(params, training, cfg["width_mult_list"])
    a = arch_idx
    alphas = [F.softmax(params["alpha_%d_%d" % (a, s)], dim=-1) for s in range(3)]
    betas = [None] + [F.softmax(params["beta_%d_%d" % (a, s)], dim=-1) for s in (1, 2)]
    ratios = sample_prun_ratio(params, cfg, a, prun_mode if prun_mode is not None else cfg["prun_modes"][a])
    y = c.conv_norm(x, "stem.%d.0" % a, 3, 2, 1)
    y = c.primitive("conv_2x", y, "stem.%d.1" % a, 2)
    y = c.primitive("conv_2x", y, "stem.%d.2" % a, 2)
    out_prev = [[y, None]]
    for i in range(layers):
        n_scales = 1 if i == 0 else 2 if i == 1 else 3
        out = []
        for j in range(n_scales):
            has_down = (i < layers - 1) and j < 2                                         # :153-170
            alpha = alphas[j][i - j]
            if i == 0 and j == 0:
                ratio = (shw[a][0], ratios[j][i - j], ratios[j + 1][i - j])
            elif i == layers - 1:
                ratio = (ratios[j][i - j - 1] if j == 0 else ratios[j][i - j], shw[a][1], None)
            elif j == 2:
                ratio = (ratios[j][i - j], ratios[j][i - j + 1], None)
            elif j == 0:
                ratio = (ratios[j][i - j - 1], ratios[j][i - j], ratios[j + 1][i - j])
            else:
                ratio = (ratios[j][i - j], ratios[j][i - j + 1], ratios[j + 1][i - j])
            prefix = "cells.%d.%d" % (i, j)
            if j == 0:
                out.append(cell(c, out_prev[0][0], prefix, has_down, alpha, ratio))
            elif i == j:
                out.append(cell(c, out_prev[j - 1][1], prefix, has_down, alpha, ratio))
            else:
                b = betas[j][i - j - 1]
                o0 = d0 = o1 = d1 = None
                if b[0] > 0:
                    o0, d0 = cell(c, out_prev[j - 1][1], prefix, has_down, alpha, ratio)
                if b[1] > 0:
                    o1, d1 = cell(c, out_prev[j][0], prefix, has_down, alpha, ratio)
                out.append((sum(w * o for w, o in zip(b, [o0, o1])),
                            sum(w * d if d is not None else 0 for w, d in zip(b, [d0, d1]))))
        out_prev = out
    up2 = lambda t: resize(t, (t.shape[2] * 2, t.shape[3] * 2))
    out0 = out[0][0]
    out1 = up2(c.conv_norm(out[1][0], "refine16.%d.0" % a, 1, 1, 0))
    out1 = c.conv_norm(torch.cat([out1, out[0][0]], 1), "refine16.%d.1" % a, 3, 1, 1)
    out2 = up2(c.conv_norm(out[2][0], "refine32.%d.0" % a, 1, 1, 0))
    out2 = c.conv_norm(torch.cat([out2, out[1][0]], 1), "refine32.%d.1" % a, 3, 1, 1)
    out2 = up2(c.conv_norm(out2, "refine32.%d.2" % a, 1, 1, 0))
    out2 = c.conv_norm(torch.cat([out2, out[0][0]], 1), "refine32.%d.3" % a, 3, 1, 1)
    preds = [c.head(out0, "head0.%d" % a), c.head(out1, "head1.%d" % a), c.head(out2, "head2.%d" % a),
             c.head(torch.cat([out0, out2], 1), "head02.%d" % a), c.head(torch.cat([out1, out2], 1), "head12.%d" % a)]
    if not training:
        preds = [resize(p, (p.shape[2] * 8, p.shape[3] * 8)) for p in preds]
    return preds


def loss(params, cfg, x, target, pretrain, criterion=None):
    """Network_Multi_Path._loss (:478-505) for width_mult_list of length > 1; arch_idx follows the reference (the search
    passes leave it at the last architecture; pretrain never switches it: SURVEY.md A.3 quirk 5)."""
    crit = criterion or torch.nn.CrossEntropyLoss(ignore_index=255)
    total, arch_idx = 0, 0
    if not pretrain:
        for idx in range(len(cfg["prun_modes"])):
            arch_idx = idx
            total = total + sum(crit(p, target) for p in forward(params, cfg, x, idx, None))
    for mode in (("max", "min", "random", "random") if pretrain else ("max", "min")):
        total = total + sum(crit(p, target) for p in forward(params, cfg, x, arch_idx, mode))
    return total
