"""Derived (discrete) multi-branch network built from searched architecture parameters.

Drop-in for the reference's train/model_seg.py: `Network_Multi_Path_Infer(alphas, betas, ratios, num_classes, layers,
criterion, Fch, width_mult_list, stem_head_width, ignore_skip)`, `.build_structure(lasts)`, `.forward(input)`,
`.forward_latency(size)`, and the decode helpers (`network_metas`, `alphas2ops_path_width`, `betas2path`,
`path2widths`, `path2downs`, `downs2path`).  Module names match, so state_dict keys are identical and
`train/fasterseg/arch_{0,1}.pt` build the same networks (tests/test_arch_decode.py checks against fixtures produced by
the reference).  The decode is re-implemented (numpy on host), *including* the reference's in-place side effects on
its arguments: network_metas replaces betas[1], betas[2] by their softmax on every call and alphas2ops_path_width
writes -inf into alpha rows, and __init__ calls it three times on the same objects (model_seg.py:198-200).
"""
import numpy as np
import torch
import torch.nn as nn
from torch.nn import functional as F

from . import functional as FN
from .genotypes import PRIMITIVES
from .nn import BatchNorm2d
from .operations import *            # noqa: F401,F403  (reference modules do the same: names re-exported)
from .operations import OPS, BasicResidual2x, ConvNorm
from .seg_oprs import FeatureFusion, Head


def softmax(x):
    return np.exp(x) / (np.exp(x).sum() + np.spacing(1))


def path2downs(path):
    '''
    0 same 1 down
    '''
    steps = [b - a for a, b in zip(path[:-1], path[1:])]
    assert all(s in (0, 1) for s in steps)
    return steps + [0]


def downs2path(downs):
    path = [0]
    for d in downs[:-1]:
        path.append(path[-1] + (1 if d == 1 else 0))
    return path


def _row(alphas, path, i):
    scale = path[i]
    return alphas[scale][i - scale]


def _skip_score(row):
    return float(F.softmax(row, dim=-1)[0])


def alphas2ops_path_width(alphas, path, widths, ignore_skip=False):
    '''
    alphas: [alphas0, ..., alphas3]  (rows are modified in place, as in the reference model_seg.py:40-96)
    '''
    n = len(path)
    assert n == len(widths) + 1, "len(path) %d, len(widths) %d" % (n, len(widths))
    min_len = int(np.round(n / 3.)) + path[-1] * 2
    NEG = -float('inf')

    # 1) candidate skips: argmax is 'skip' and the layer does not change scale
    candidates = []                      # (position, softmax prob of skip), in position order
    for i in range(n):
        row = _row(alphas, path, i)
        if ignore_skip:
            row[0] = NEG
        if int(row.argmax()) == 0 and (i == n - 1 or path[i] == path[i + 1]):
            candidates.append((i, _skip_score(row)))

    # 2) between two consecutive down-samples (and from the last one to the end) not every layer may be a skip:
    #    forbid skip on the weakest one (last one on ties)
    cand_pos = [p for p, _ in candidates]
    down_pos = [p for p in range(n - 1) if path[p] < path[p + 1]]
    if down_pos:
        for lo, hi in zip(down_pos, down_pos[1:] + [n]):
            first, last = lo + 1, hi - 1
            if first in cand_pos and last in cand_pos and cand_pos.index(last) - cand_pos.index(first) == last - first:
                best_score, best_pos = 1, -1
                for j in range(first, hi):
                    score = _skip_score(_row(alphas, path, j))
                    if score <= best_score:
                        best_score, best_pos = score, j
                _row(alphas, path, best_pos)[0] = NEG

    # 3) at most n - min_len layers may be dropped: keep the most confident skips
    budget = n - min_len
    if len(candidates) > budget:
        candidates = sorted(candidates, key=lambda c: c[1], reverse=True)[:budget]
    drop = set(p for p, _ in candidates)

    ops, path_compact, widths_compact = [], [], []
    for i in range(n):
        row = _row(alphas, path, i)
        op = int(row.argmax())
        if op == 0:
            if i in drop:
                if i == n - 1:           # dropping the last layer also drops the width that fed it
                    widths_compact = widths_compact[:-1]
                continue
            row[0] = NEG                  # skip not allowed here: take the runner-up
            op = int(row.argmax())
        path_compact.append(path[i])
        if i < len(widths):
            widths_compact.append(widths[i])
        ops.append(op)
    assert len(path_compact) >= min_len
    return ops, path_compact, widths_compact


def betas2path(betas, last, layers):
    downs = [0] * layers
    # betas1 is of length layers-2; beta2: layers-3
    if last == 1:
        b1 = betas[1].detach().cpu().numpy()
        downs[int(np.argmax(b1[1:-1, 0])) + 1] = 1
    elif last == 2:
        b1 = betas[1].detach().cpu().numpy()
        b2 = betas[2].detach().cpu().numpy()
        best, best_ij = 0, (0, 1)
        for j in range(layers - 4):
            for i in range(1, j - 1):
                prob = b1[i, 0] * b2[j, 0]
                if prob > best:
                    best, best_ij = prob, (i, j)
        downs[best_ij[0] + 1] = 1
        downs[best_ij[1] + 2] = 1
    path = downs2path(downs)
    assert path[-1] == last
    return path


def path2widths(path, ratios, width_mult_list):
    widths = []
    for layer in range(1, len(path)):
        scale = path[layer]
        row = ratios[0][layer - 1] if scale == 0 else ratios[scale][layer - scale]
        widths.append(width_mult_list[int(row.argmax())])
    return widths


def network_metas(alphas, betas, ratios, width_mult_list, layers, last, ignore_skip=False):
    betas[1] = F.softmax(betas[1], dim=-1)      # list entries are replaced: repeated calls re-softmax (reference :128-129)
    betas[2] = F.softmax(betas[2], dim=-1)
    path = betas2path(betas, last, layers)
    widths = path2widths(path, ratios, width_mult_list)
    ops, path, widths = alphas2ops_path_width(alphas, path, widths, ignore_skip=ignore_skip)
    assert len(ops) == len(path) and len(path) == len(widths) + 1, "op %d, path %d, width%d" % (len(ops), len(path), len(widths))
    downs = path2downs(path)  # 0 same 1 down
    return ops, path, downs, widths


class MixedOp(nn.Module):
    def __init__(self, C_in, C_out, op_idx, stride=1):
        super(MixedOp, self).__init__()
        self._op = OPS[PRIMITIVES[op_idx]](C_in, C_out, stride, slimmable=False, width_mult_list=[1.])

    def forward(self, x):
        return self._op(x)

    def forward_latency(self, size):
        latency, size_out = self._op.forward_latency(size)
        return latency, size_out


class Cell(nn.Module):
    def __init__(self, op_idx, C_in, C_out, down):
        super(Cell, self).__init__()
        self._C_in = C_in
        self._C_out = C_out
        self._down = down
        self._op = MixedOp(C_in, C_out, op_idx, stride=2 if down else 1)

    def forward(self, input):
        return self._op(input)

    def forward_latency(self, size):
        return self._op.forward_latency(size)


class Network_Multi_Path_Infer(nn.Module):
    def __init__(self, alphas, betas, ratios, num_classes=19, layers=9, criterion=nn.CrossEntropyLoss(ignore_index=-1), Fch=12,
                 width_mult_list=[1., ], stem_head_width=(1., 1.), ignore_skip=False):
        super(Network_Multi_Path_Infer, self).__init__()
        self._num_classes = num_classes
        assert layers >= 2
        self._layers = layers
        self._criterion = criterion
        self._Fch = Fch
        if ratios[0].size(1) == 1:
            self._width_mult_list = [1., ] if ignore_skip else [4. / 12, ]
        else:
            self._width_mult_list = width_mult_list
        self._stem_head_width = stem_head_width
        self.latency = 0

        w0 = stem_head_width[0]
        self.stem = nn.Sequential(
            ConvNorm(3, self.num_filters(2, w0) * 2, kernel_size=3, stride=2, padding=1, bias=False, groups=1, slimmable=False),
            BasicResidual2x(self.num_filters(2, w0) * 2, self.num_filters(4, w0) * 2, kernel_size=3, stride=2, groups=1, slimmable=False),
            BasicResidual2x(self.num_filters(4, w0) * 2, self.num_filters(8, w0), kernel_size=3, stride=2, groups=1, slimmable=False)
        )
        for last in (0, 1, 2):       # order matters: the calls share (and mutate) alphas / betas
            ops, path, downs, widths = network_metas(alphas, betas, ratios, self._width_mult_list, layers, last,
                                                     ignore_skip=ignore_skip)
            setattr(self, "ops%d" % last, ops)
            setattr(self, "path%d" % last, path)
            setattr(self, "downs%d" % last, downs)
            setattr(self, "widths%d" % last, widths)

    def num_filters(self, scale, width=1.0):
        return int(np.round(scale * self._Fch * width))

    def build_structure(self, lasts):
        self._branch = len(lasts)
        self.lasts = lasts
        self.ops = [getattr(self, "ops%d" % last) for last in lasts]
        self.paths = [getattr(self, "path%d" % last) for last in lasts]
        self.downs = [getattr(self, "downs%d" % last) for last in lasts]
        self.widths = [getattr(self, "widths%d" % last) for last in lasts]
        self.branch_groups, self.cells = self.get_branch_groups_cells(self.ops, self.paths, self.downs, self.widths, self.lasts)
        self.build_arm_ffm_head()

    def build_arm_ffm_head(self):
        hw = self._stem_head_width[1]
        nf = self.num_filters
        if self.training:        # auxiliary heads exist only when built in train mode (reference :216-224)
            if 2 in self.lasts:
                self.heads32 = Head(nf(32, hw), self._num_classes, True, norm_layer=BatchNorm2d)
                if 1 in self.lasts:
                    self.heads16 = Head(nf(16, hw) + self.ch_16, self._num_classes, True, norm_layer=BatchNorm2d)
                else:
                    self.heads16 = Head(self.ch_16, self._num_classes, True, norm_layer=BatchNorm2d)
            else:
                self.heads16 = Head(nf(16, hw), self._num_classes, True, norm_layer=BatchNorm2d)
        self.heads8 = Head(nf(8, hw) * self._branch, self._num_classes, Fch=self._Fch, scale=4, branch=self._branch,
                           is_aux=False, norm_layer=BatchNorm2d)
        if 2 in self.lasts:
            self.arms32 = nn.ModuleList([
                ConvNorm(nf(32, hw), nf(16, hw), 1, 1, 0, slimmable=False),
                ConvNorm(nf(16, hw), nf(8, hw), 1, 1, 0, slimmable=False),
            ])
            self.refines32 = nn.ModuleList([
                ConvNorm(nf(16, hw) + self.ch_16, nf(16, hw), 3, 1, 1, slimmable=False),
                ConvNorm(nf(8, hw) + self.ch_8_2, nf(8, hw), 3, 1, 1, slimmable=False),
            ])
        if 1 in self.lasts:
            self.arms16 = ConvNorm(nf(16, hw), nf(8, hw), 1, 1, 0, slimmable=False)
            self.refines16 = ConvNorm(nf(8, hw) + self.ch_8_1, nf(8, hw), 3, 1, 1, slimmable=False)
        self.ffm = FeatureFusion(nf(8, hw) * self._branch, nf(8, hw) * self._branch, reduction=1, Fch=self._Fch, scale=8,
                                 branch=self._branch, norm_layer=BatchNorm2d)

    def get_branch_groups_cells(self, ops, paths, downs, widths, lasts):
        """Branches that share scale, op, width and next scale in every layer so far are merged into one Cell that is
        registered under each branch's key (`cells.<layer>-<branch>`), reference :241-296."""
        num_branch = len(ops)
        layers = max(len(path) for path in paths)
        groups_all = []
        self.ch_16 = 0
        self.ch_8_2 = 0
        self.ch_8_1 = 0
        cells = nn.ModuleDict()  # layer-branch: op
        same_so_far = np.ones((num_branch, num_branch))

        def differs(i, j, l):
            return (len(paths[i]) <= l + 1 or len(paths[j]) <= l + 1 or paths[i][l + 1] != paths[j][l + 1]
                    or ops[i][l] != ops[j][l] or widths[i][l] != widths[j][l])

        for l in range(layers):
            for i in range(num_branch):
                for j in range(i + 1, num_branch):
                    if differs(i, j, l):       # the last layer of a branch never merges
                        same_so_far[i, j] = same_so_far[j, i] = 0
            branch_groups = []
            for branch in range(num_branch):
                if len(paths[branch]) < l + 1:
                    continue
                homes = [g for g in branch_groups if same_so_far[g[0], branch] == 1]
                for g in homes:
                    g.append(branch)
                if not homes:
                    branch_groups.append([branch])
            for group in branch_groups:
                lead = group[0]
                for other in group[1:]:       # members of a group must agree on op / next scale / down / width
                    assert ops[lead][l] == ops[other][l] and paths[lead][l + 1] == paths[other][l + 1] \
                        and downs[lead][l] == downs[other][l] and widths[lead][l] == widths[other][l]
                op = ops[lead][l]
                scale = 2 ** (paths[lead][l] + 3)
                down = downs[lead][l]
                depth = len(paths[lead])
                if l < depth - 1:
                    assert down == paths[lead][l + 1] - paths[lead][l]
                assert down in [0, 1]
                if l == 0:
                    c_in = self.num_filters(scale, self._stem_head_width[0])
                    c_out = self.num_filters(scale * (down + 1), widths[lead][l])
                elif l == depth - 1:      # last cell of this branch feeds the head
                    assert down == 0
                    c_in = self.num_filters(scale, widths[lead][l - 1])
                    c_out = self.num_filters(scale, self._stem_head_width[1])
                else:
                    c_in = self.num_filters(scale, widths[lead][l - 1])
                    c_out = self.num_filters(scale * (down + 1), widths[lead][l])
                cell = Cell(op, c_in, c_out, down)
                # feature fusion needs the channel count of the last 1/16 and 1/8 maps of the 1/32 branch and the
                # last 1/8 map of the 1/16 branch (the inputs of their down-sampling cells)
                if 2 in self.lasts and self.lasts.index(2) in group and down and scale == 16:
                    self.ch_16 = cell._C_in
                if 2 in self.lasts and self.lasts.index(2) in group and down and scale == 8:
                    self.ch_8_2 = cell._C_in
                if 1 in self.lasts and self.lasts.index(1) in group and down and scale == 8:
                    self.ch_8_1 = cell._C_in
                for branch in group:
                    cells[str(l) + "-" + str(branch)] = cell
            groups_all.append(branch_groups)
        return groups_all, cells

    def _arm_up_refine(self, x, arm, target, refine):
        """1x1 arm -> bilinear up to `target`'s size -> channel concat -> 3x3 refine (reference :304-312)."""
        out = arm(x)
        out = FN.interpolate(out, size=(target.size(2), target.size(3)))
        return refine(FN.cat([out, target]))

    def agg_ffm(self, outputs8, outputs16, outputs32):
        pred32 = []; pred16 = []; pred8 = []  # order of predictions is not important
        for branch in range(self._branch):
            last = self.lasts[branch]
            if last == 2:
                if self.training: pred32.append(outputs32[branch])
                out = self._arm_up_refine(outputs32[branch], self.arms32[0], outputs16[branch], self.refines32[0])
                if self.training: pred16.append(outputs16[branch])
                out = self._arm_up_refine(out, self.arms32[1], outputs8[branch], self.refines32[1])
                pred8.append(out)
            elif last == 1:
                if self.training: pred16.append(outputs16[branch])
                out = self._arm_up_refine(outputs16[branch], self.arms16, outputs8[branch], self.refines16)
                pred8.append(out)
            elif last == 0:
                pred8.append(outputs8[branch])
        join = lambda ts: ts[0] if len(ts) == 1 else FN.cat(ts)
        pred32 = self.heads32(join(pred32)) if len(pred32) > 0 else None
        pred16 = self.heads16(join(pred16)) if len(pred16) > 0 else None
        pred8 = self.heads8(self.ffm(join(pred8)))
        if self.training:
            return pred8, pred16, pred32
        return pred8

    def forward_lowres(self, input):
        """forward() without its last line (model_seg.py:357-365): the heads' logits at 1/8, 1/16, 1/32 resolution as NHWC
        views (channel stride 32).  For losses that evaluate the bilinear up-sample themselves (losses.ohem_ce_lowres,
        distill_kl_lowres) instead of reading (B, 19, H, W) fp32 tensors."""
        return self.forward(input, lowres=True)

    def forward(self, input, lowres=False):
        _, _, H, W = input.size()
        stem = self.stem(input)

        # store the last feature map w. corresponding scale of each branch
        outputs8 = [stem] * self._branch
        outputs16 = [stem] * self._branch
        outputs32 = [stem] * self._branch
        outputs = [stem] * self._branch

        for layer in range(len(self.branch_groups)):
            for group in self.branch_groups[layer]:
                output = self.cells[str(layer) + "-" + str(group[0])](outputs[group[0]])
                scale = int(H // output.size(2))
                for branch in group:
                    outputs[branch] = output
                    if scale == 8: outputs8[branch] = output
                    elif scale == 16: outputs16[branch] = output
                    elif scale == 32: outputs32[branch] = output

        up = lambda t, f: None if t is None else FN.interpolate(t, size=(int(t.size(2)) * f, int(t.size(3)) * f), out_nchw=1)
        if self.training:
            pred8, pred16, pred32 = self.agg_ffm(outputs8, outputs16, outputs32)
            if lowres:
                return pred8, pred16, pred32
            return up(pred8, 8), up(pred16, 16), up(pred32, 32)     # contiguous NCHW fp32 logits
        pred8 = self.agg_ffm(outputs8, outputs16, outputs32)
        return pred8 if lowres else up(pred8, 8)

    def forward_latency(self, size):
        _, H, W = size
        latency_total = 0
        for i in range(3):
            latency, size = self.stem[i].forward_latency(size); latency_total += latency

        # store the last feature map w. corresponding scale of each branch
        outputs8 = [size] * self._branch
        outputs16 = [size] * self._branch
        outputs32 = [size] * self._branch
        outputs = [size] * self._branch

        for layer in range(len(self.branch_groups)):
            for group in self.branch_groups[layer]:
                latency, size = self.cells[str(layer) + "-" + str(group[0])].forward_latency(outputs[group[0]])
                latency_total += latency
                scale = int(H // size[1])
                for branch in group:
                    outputs[branch] = size
                    # the reference (:388) tests `scale == 4` and writes an undefined `outputs4`; no shipped or
                    # searchable architecture has a 1/4 map after the stem, so only 1/16 and 1/32 are tracked and
                    # outputs8 keeps the stem size (equal to the true 1/8 size) exactly as there.
                    if scale == 16: outputs16[branch] = size
                    elif scale == 32: outputs32[branch] = size

        for branch in range(self._branch):
            last = self.lasts[branch]
            if last == 2:
                latency, size = self.arms32[0].forward_latency(outputs32[branch]); latency_total += latency
                latency, size = self.refines32[0].forward_latency((size[0] + self.ch_16, size[1] * 2, size[2] * 2)); latency_total += latency
                latency, size = self.arms32[1].forward_latency(size); latency_total += latency
                latency, size = self.refines32[1].forward_latency((size[0] + self.ch_8_2, size[1] * 2, size[2] * 2)); latency_total += latency
                out_size = size
            elif last == 1:
                latency, size = self.arms16.forward_latency(outputs16[branch]); latency_total += latency
                latency, size = self.refines16.forward_latency((size[0] + self.ch_8_1, size[1] * 2, size[2] * 2)); latency_total += latency
                out_size = size
            elif last == 0:
                out_size = outputs8[branch]
        latency, size = self.ffm.forward_latency((out_size[0] * self._branch, out_size[1], out_size[2])); latency_total += latency
        latency, size = self.heads8.forward_latency(size); latency_total += latency
        return latency_total, size
