"""Supernet for architecture search (drop-in for the reference's search/model_search.py).

`Network_Multi_Path(num_classes, layers, criterion, Fch, width_mult_list, prun_modes, stem_head_width)` keeps the
reference's module tree (`stem.<i>`, `cells.<l>.<s>.{_op,downsample}._ops.<k>`, `refine32/refine16`, `head*`), its
architecture parameters (`alpha_<i>_<s>`, `beta_<i>_<s>`, `ratio_<i>_<s>`, `_arch_parameters`, `_arch_names`) and the
methods the search driver calls (`forward`, `_loss`, `forward_latency`, `sample_prun_ratio`, `arch_idx`, `prun_mode`).

What differs underneath: every primitive is a chain of fused HIP kernels (operations.py); the MixedOp sum
`result + op(x) * w * r0 * r1` (model_search.py:76-78) and the beta-weighted merges (:330-333) are one
fs_axpy_channels pass per term with the scalar coefficient left on the device (its gradient is a fused full-tensor dot,
functional.scale_accumulate), and the `betas[..] > 0` tests that make the reference synchronise the GPU once per cell
are evaluated once per forward.
"""
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import functional as FN
from . import kernels as K
from .genotypes import PRIMITIVES
from .operations import *            # noqa: F401,F403
from .operations import OPS, BasicResidual2x, ConvNorm
from .seg_oprs import Head
from .latency_model import LatencyModelMixin


# https://github.com/YongfeiYan/Gumbel_Softmax_VAE (as cited by the reference)
def sample_gumbel(shape, eps=1e-20, device=None):
    U = torch.rand(shape)                       # host RNG like the reference (model_search.py:15-17): ranks stay in sync
    if device is not None:
        if torch.device(device).type == "cuda":
            # a blocking `.to(device)` of pageable memory ends in a synchronisation of the launch stream - a device drain per call, 28 ms
            # of a host-bound C5 iteration (profiles/r05_host_profile_c5.txt).  Pinned + non_blocking: the copy is just another enqueue
            # (the caching host allocator keeps the staging block alive until the stream has consumed it).
            U = U.pin_memory().to(device, non_blocking=True)
        else:
            U = U.to(device)
    return -torch.log(-torch.log(U + eps) + eps)


def gumbel_softmax_sample(logits, temperature=1):
    y = logits + sample_gumbel(logits.size(), device=logits.device)
    return F.softmax(y / temperature, dim=-1)


def gumbel_softmax(logits, temperature=1, hard=False):
    """
    ST-gumple-softmax
    input: [*, n_class]
    return: flatten --> [*, n_class] an one-hot vector
    """
    y = gumbel_softmax_sample(logits, temperature)
    if not hard:
        return y
    shape = y.size()
    _, ind = y.max(dim=-1)
    y_hard = torch.zeros_like(y).view(-1, shape[-1])
    y_hard.scatter_(1, ind.view(-1, 1), 1)
    y_hard = y_hard.view(*shape)
    # straight-through: forward one-hot, gradient of the soft sample
    out = (y_hard - y).detach() + y
    out._fs_index_t = ind.reshape(-1)          # device-side arg-max, read back in one batch by sample_prun_ratio
    return out


class _SampledRatios(list):
    """sample_prun_ratio("arch_ratio") result: the reference's nested [scale][layer] list of straight-through one-hots, plus the batched
    tensors they are rows of (`stacked` [slots, widths], `index` [slots] = device-side arg-max) for the consumers that want all slots."""
    stacked = None
    index = None


def gumbel_softmax_rows(logits, temperature=1):
    """gumbel_softmax(hard=True) of every ROW of `logits` [slots, widths] in one batch of ~14 launches instead of ~14 per slot (44 slots
    per call, three calls per architecture step: ~1.8 k of a C5 iteration's ATen launches).  The host RNG is consumed exactly like the
    reference's slot-by-slot `torch.rand(widths)` calls (one serial uniform stream: torch.rand(n, w) == n x torch.rand(w), checked in
    tests/test_supernet.py), so seeded runs draw the same sub-networks.  Returns (straight-through one-hots [slots, widths], arg-max)."""
    y = logits + sample_gumbel(logits.size(), device=logits.device)
    if temperature != 1:                      # (x / 1 is the identity: no launch for the reference's default)
        y = y / temperature
    y = F.softmax(y, dim=-1)
    _, ind = y.max(dim=-1)
    y_hard = torch.zeros_like(y).scatter_(1, ind.view(-1, 1), 1)
    return (y_hard - y).detach() + y, ind


def _width_and_score(ratio, width_mult_list):
    """int: force #channel; tensor: arch_ratio; float(<=1): force width (reference comment, model_search.py:61)."""
    if isinstance(ratio, torch.Tensor):
        k = getattr(ratio, "_fs_index", None)
        if k is None:
            k = int(ratio.argmax())            # one host sync; sample_prun_ratio pre-reads all of a forward's indices at once
        return width_mult_list[k], ratio[k]
    return ratio, 1.


# Inside a hipGraph capture the five primitives of a MixedOp (independent given x) are issued on separate HIP streams, so
# the captured graph has five parallel chains per MixedOp instead of one: the supernet's kernels are a few microseconds on
# a fraction of the CUs, and a replayed pass is bound by their serial latency, not by throughput.  autograd runs each
# node's backward on the stream of its forward, so the backward graph forks and joins the same way.  Eager passes are
# host-bound and stay on one stream.  FS_BRANCH_LANES=1 disables the fork.
_BRANCH_LANES = int(os.environ.get("FS_BRANCH_LANES", "5"))
# Eager training passes replay each MixedOp from pre-built launch programs (fasterseg_amd/program.py); FS_MIXEDOP_PROGRAMS=0
# keeps the per-module autograd path.
_PROGRAMS = bool(int(os.environ.get("FS_MIXEDOP_PROGRAMS", "1")))
# ... also inside a capture (one program per MixedOp on its own lane instead of five per-module autograd chains): fewer nodes,
# and the five input gradients of a MixedOp are summed by one kernel instead of four autograd adds.  FS_CAPTURE_PROGRAMS=0
# captures the per-module path.
_CAPTURE_PROGRAMS = bool(int(os.environ.get("FS_CAPTURE_PROGRAMS", "1")))
# One level up, all MixedOps of a layer (up to 3 scales x {from-down, from-keep} x {_op, downsample} = 12) only depend on the
# previous layer: under capture the primitives of ALL of them fork from the capturing stream at once (one flat fork / join
# per layer over a pool of 20 streams - 60 measured slower; nested forks - a lane forking its own lanes - crash hipStreamEndCapture on ROCm 7.2),
# so a layer's critical path is one primitive chain instead of twelve MixedOps in a row.  FS_LAYER_LANES=1 disables it.
_LAYER_LANES = int(os.environ.get("FS_LAYER_LANES", "20"))
# A cell that is fed both from the scale above (down) and from its own scale (keep) is evaluated ONCE on the two inputs
# concatenated along the batch, its BatchNorms normalising the two halves independently (functional.bn_groups): the arithmetic
# of the reference's two evaluations (model_search.py:322-329) at half the launches - the supernet step is launch-bound.
# FS_PAIR_BATCH=0 evaluates the two inputs one after the other.
_PAIR_BATCH = bool(int(os.environ.get("FS_PAIR_BATCH", "1")))
# Round 6: the producers of a pair's two inputs write straight into the halves of the joint buffer (FS_PAIR_DIRECT=0: two copy launches
# per pair, as in round 5), and the beta merges of a layer's pair-batched cells are one grouped launch (FS_MERGE_GROUP=0: one each).
_PAIR_DIRECT = bool(int(os.environ.get("FS_PAIR_DIRECT", "1")))
_MERGE_GROUP = bool(int(os.environ.get("FS_MERGE_GROUP", "1")))


# Eager ("random" / Gumbel width) passes replay one launch program per MixedOp.  On ONE stream such a pass is bound by the serial
# latency of ~4.4 k kernels of a few microseconds each (36 ms, as long as the much larger max-width pass takes from its graph), while
# the host needs ~2 us to issue one: the MixedOps of a layer only depend on the previous layer, so their programs are issued
# round-robin on a few side streams (fork after the previous layer, join before the beta merges) and autograd replays the same
# fork / join in backward (a node's backward runs on the stream of its forward).  FS_EAGER_LANES=1 keeps one stream.
_EAGER_LANES = int(os.environ.get("FS_EAGER_LANES", "4"))     # measured on C3: 1 lane 130.6 ms, 4: 120.0, 6: 122.1, 10: 134.2


# The MixedOps of a layer are replayed TOGETHER by fs_exec_program_group (csrc/program.hip): the next commands of all their launch
# programs are scheduled so that commands of one kind - conv -> BN units, bare convolutions, weight / data gradients and, since round 6,
# the BatchNorm passes, bilinear resamples and weighted sums (csrc/group.h) - go out as ONE grouped launch each, whatever the programs'
# structure (stride-1 and stride-2 MixedOps, with or without input gradient meet at their common commands).  The step is the sum of its
# kernel durations and every launch pays ~4 us of ramp-up + boundary.  FS_GROUP_PROGRAMS=0: one program per stream lane.
_GROUP_PROGRAMS = bool(int(os.environ.get("FS_GROUP_PROGRAMS", "1")))
# Inside a hipGraph capture, FS_GROUP_CAPTURE selects where the layer calls go: 1 (default) the capture's origin stream - the captured
# pass is then one LINEAR graph, which ROCm replays at ~0.5 us of host time per kernel node instead of ~4 us for a forked graph
# (profiles/r05_host_vs_device_c3.txt); 2 one dedicated side lane; 0 no grouping inside captures (one program per forked lane, the
# round-3..5 layout).  Rounds 4-5 refused 1: the graphs gave inf gradients from the ~5th replay on.  Root cause, found in round 6
# (profiles/r06_capture_fault_matrix*.txt): the launch programs of rounds 3-4 cleared their accumulators with
# hipMemsetAsync, and ROCm 7.x does not keep a MEMSET node captured on the origin stream of a multi-stream capture ordered against the
# kernel nodes around it in later replays (on forked side streams it does) - BatchNorm statistics accumulated across replays.  Round 5
# replaced those memsets by one fill KERNEL per captured pass (kernels.zero_pool.begin_capture) for launch-count reasons, which removed
# the trigger; FS_ZERO_MEMSET=1 brings memset nodes and the fault back.  Layer calls on SEVERAL side lanes (>= 3) still crash
# hipStreamEndCapture on ROCm 7.2 (core dump; a multi-output autograd node whose backward runs on a forked lane) and are refused.
_GROUP_CAPTURE = int(os.environ.get("FS_GROUP_CAPTURE", "1"))
if _GROUP_CAPTURE >= 3 and not int(os.environ.get("FS_ALLOW_BROKEN_CAPTURE", "0")):
    raise RuntimeError("FS_GROUP_CAPTURE >= 3 (layer calls on several side lanes of a capture) crashes hipStreamEndCapture on ROCm 7.2; "
                       "use 1 (origin stream, default), 2 (one side lane) or 0 (no grouping inside captures)")
# Eager passes: the layer's programs go out as FS_LAYER_SPLIT calls side by side on that many lanes (1: one call on the current stream).
_LAYER_SPLIT = max(1, int(os.environ.get("FS_LAYER_SPLIT", "1")))


# SupernetStep.step(force_eager=True) - bench.py's census step - sets this: the fixed-width passes, which the timed steps replay from
# hipGraphs, then issue the launches the capture would have recorded (ungrouped unless FS_GROUP_CAPTURE), so that the census times the
# kernels of the timed steps and agrees with a rocprofv3 table of them.
MIMIC_CAPTURE = False
# set by train_step.SupernetStep._set_phase while only the ~600 "probe" tensors follow the phase flips (see there)
FAST_PHASE_ACTIVE = False

_RECORD_STREAM = bool(int(os.environ.get("FS_RECORD_STREAM", "1")))


# how a layer's evaluations are issued - stream lanes, layer calls of the launch-program executor, conflict-free chunks - is layer_exec.py
from .layer_exec import branch_lanes, conflict_free_chunks, layer_lanes          # noqa: E402,F401
from .layer_exec import run_branches as _run_branches, run_tasks as _run_tasks   # noqa: E402,F401


# forward_multi(batch_tails=True): refinement + heads of a pass pair in one evaluation on the batch-concatenated maps; FS_TAIL_BATCH=0: per pass
_TAIL_BATCH = bool(int(os.environ.get("FS_TAIL_BATCH", "1")))
# ... and, OPT-IN (FS_STEM_SHARE=1), their stem evaluated once: the two passes feed the same images through the same stem weights, so the
# second evaluation is a common subexpression (outputs, gradients and BatchNorm statistics end up identical, tested) - but the reference
# does evaluate it per pass, and the benchmarked step does the reference's work: default off (C3 fp32 +1.8 ms without it)
_STEM_SHARE = bool(int(os.environ.get("FS_STEM_SHARE", "0")))


class _DeferredTail:
    """What a pass hands back when its refinement / heads are evaluated by the caller: architecture index + the last layer's maps."""
    __slots__ = ("k", "o0", "o1", "o2")

    def __init__(self, k, o0, o1, o2):
        self.k, self.o0, self.o1, self.o2 = k, o0, o1, o2


class JointLogits:
    """The five logits tensors of `passes` passes concatenated along the batch (pass after pass)."""
    __slots__ = ("logits", "passes")

    def __init__(self, logits, passes):
        self.logits, self.passes = logits, passes

    def split(self):
        n = self.logits[0].shape[0] // self.passes
        return [tuple(t[p * n:(p + 1) * n] for t in self.logits) for p in range(self.passes)]


class _PreCoef:
    """The five coefficients of one MixedOp evaluation, already multiplied by the width scores (Network_Multi_Path computes
    those of a whole pass in a few batched ops, see `_coefficient_rows`)."""
    __slots__ = ("t",)

    def __init__(self, t):
        self.t = t


class MixedOp(nn.Module):

    def __init__(self, C_in, C_out, stride=1, width_mult_list=[1.]):
        super(MixedOp, self).__init__()
        self._ops = nn.ModuleList()
        self._width_mult_list = width_mult_list
        for primitive in PRIMITIVES:
            self._ops.append(OPS[primitive](C_in, C_out, stride, True, width_mult_list=width_mult_list))

    def set_prun_ratio(self, ratio):
        for op in self._ops:
            op.set_ratio(ratio)

    def _coefficients(self, x, weights, ratios, widths=None, set_ratio=True):
        """Selects the widths (set_prun_ratio, unless the caller does it when needed) and returns the five coefficients
        w_k * r_score0 * r_score1 (reference :64-78)."""
        ratio0, r_score0 = _width_and_score(ratios[0], self._width_mult_list)
        ratio1, r_score1 = _width_and_score(ratios[1], self._width_mult_list)
        if set_ratio:
            self.set_prun_ratio((ratio0, ratio1))
        if widths is not None:
            widths[:] = [ratio0, ratio1]
        if isinstance(weights, _PreCoef):
            return weights.t
        coef = weights                                 # sum_k w_k * r_score0 * r_score1 * op_k(x), reference :76-78
        if not torch.is_tensor(coef):
            coef = torch.stack([torch.as_tensor(w, dtype=torch.float32, device=x.device).reshape(()) for w in coef])
        if torch.is_tensor(r_score0):                  # (a forced width has score 1.: no `* 1.` launches)
            coef = coef * r_score0
        if torch.is_tensor(r_score1):
            coef = coef * r_score1
        return coef

    def forward(self, x, weights, ratios):
        widths = [None, None]
        coef = self._coefficients(x, weights, ratios, widths)
        ratio0, ratio1 = widths
        if x.is_cuda and torch.cuda.is_current_stream_capturing():
            if _BRANCH_LANES > 1:
                return FN.weighted_sum(_run_branches(self._ops, x), coef)
        elif _PROGRAMS and self.training and x.is_cuda and torch.is_grad_enabled():
            prog = self._program(FN.as_nhwc(x), coef, ratio0, ratio1)
            if prog is not None:
                return FN.mixed_op_program(FN.as_nhwc(x), coef, prog)
        return FN.weighted_sum([op(x) for op in self._ops], coef)

    def _program(self, x, coef, ratio0, ratio1):
        """The launch programs of this MixedOp for the current widths (built on first use), or None when the per-module
        path must run: network weights that want gradients outside a FlatGradientSync backward, nothing to differentiate."""
        w0 = self._ops[1].conv1.weight
        want_w = w0.requires_grad
        sink = FN._grad_sink if want_w else None
        if want_w and (sink is None or w0.grad is None or not sink.accepts(w0)):
            return None
        need_x, need_coef = x.requires_grad, coef.requires_grad
        if not (need_x or need_coef):
            return None
        groups = FN._bn_groups
        key = (ratio0, ratio1, tuple(x.shape), x.stride(3), x.dtype, need_x, need_coef, want_w, id(sink), groups)
        cache = self.__dict__.setdefault("_programs", {})
        prog = cache.get(key)
        if prog is None or not prog.valid():
            from . import program
            self.set_prun_ratio((ratio0, ratio1))          # the lowering reads the active widths from the modules
            prog = cache[key] = program.lower_mixed_op(self, tuple(x.shape), x.stride(3), x.dtype, x.device, need_x, need_coef,
                                                       want_w, sink, groups)
        if _SAMPLING_PASS and x.stride(3) == x.shape[1]:
            # call site of a width-sampling pass, for prewarm_programs(): everything in the key but the widths.  Recorded on cache hits
            # too: a draw that happens to equal a width pair the fixed-width passes already lowered must still register the site.
            site = (x.shape[0], x.shape[2], x.shape[3], x.dtype, x.device, need_x, need_coef, want_w, groups)
            self.__dict__.setdefault("_sites", {}).setdefault(site, (ratio0, ratio1))
        return prog

    def prewarm_programs(self):
        """Lower the programs of EVERY width combination this MixedOp can be asked for at the call sites seen so far (same
        input geometry / gradient needs, the sampled ratios ranging over the width list): a "random" or Gumbel pass draws new
        widths every step, and lowering on first use (~1 ms of Python each, 25 combinations x ~74 MixedOps) would otherwise be
        spread over the first hundreds of training steps.  Call inside the same gradient-sink state as the passes (the
        programs hard-code the flat-buffer slices).  Returns the number of programs built."""
        sampled = self.__dict__.get("_ratio_sampled")
        if sampled is None:
            return 0
        want_now = self._ops[1].conv1.weight.requires_grad
        built = 0
        for site, (r0_seen, r1_seen) in list(self.__dict__.get("_sites", {}).items()):
            n, h, w, dtype, device, need_x, need_coef, want_w, groups = site
            if want_w != want_now:
                continue
            for r0 in (self._width_mult_list if sampled[0] else [r0_seen]):
                for r1 in (self._width_mult_list if sampled[1] else [r1_seen]):
                    self.set_prun_ratio((r0, r1))
                    cin = self._ops[1].conv1.active_channels()[1]
                    x = torch.empty_strided((n, cin, h, w), (h * w * cin, 1, w * cin, cin), dtype=dtype, device=device)
                    x.requires_grad_(need_x)
                    coef = torch.empty(len(self._ops), device=device).requires_grad_(need_coef)
                    before = len(self.__dict__.get("_programs", {}))
                    with FN.bn_groups(groups):
                        self._program(x, coef, r0, r1)
                    built += len(self.__dict__.get("_programs", {})) - before
        return built

    def forward_latency(self, size, weights, ratios):
        """sum_k latency_k * w_k * r_score0 * r_score1 (reference :80-93) as ONE dot product with a cached device vector of
        the five LUT latencies instead of ten scalar kernels per MixedOp (~150 MixedOp visits per architecture step)."""
        ratio0, r_score0 = _width_and_score(ratios[0], self._width_mult_list)
        ratio1, r_score1 = _width_and_score(ratios[1], self._width_mult_list)
        self.set_prun_ratio((ratio0, ratio1))
        lats = []
        for op in self._ops:
            latency, size_out = op.forward_latency(size)
            lats.append(float(latency))
        if not torch.is_tensor(weights):
            result = sum(latency * w for latency, w in zip(lats, weights))
        else:
            key = (tuple(lats), weights.device, weights.dtype)
            vec = _latency_vectors.get(key)
            if vec is None:
                vec = _latency_vectors[key] = torch.tensor(lats, dtype=weights.dtype, device=weights.device)
            result = torch.dot(vec, weights.reshape(-1))
        if torch.is_tensor(r_score0) or r_score0 != 1.:
            result = result * r_score0
        if torch.is_tensor(r_score1) or r_score1 != 1.:
            result = result * r_score1
        return result, size_out


_latency_vectors = {}      # (five LUT latencies, device, dtype) -> device vector
# forward_latency(beta=False) as one dot product over tabulated LUT rows (Network_Multi_Path._forward_latency_linear);
# FS_LINEAR_LATENCY=0 keeps the per-MixedOp evaluation for every call.
_LINEAR_LATENCY = bool(int(os.environ.get("FS_LINEAR_LATENCY", "1")))
# the MixedOp coefficients of a whole pass in a few batched ops (Network_Multi_Path._coefficient_rows); FS_BATCHED_COEFS=0: per MixedOp
_BATCHED_COEFS = bool(int(os.environ.get("FS_BATCHED_COEFS", "1")))
# the host read of the sampled width indices from a side stream (sample_prun_ratio); FS_SAMPLE_STREAM=0: on the launch stream (a drain)
_SAMPLE_STREAM = bool(int(os.environ.get("FS_SAMPLE_STREAM", "1")))
_SAMPLED = object()        # stands in for a sampled width in _cell_ratio probes
_SAMPLING_PASS = False     # the forward in progress draws its widths ("random" / Gumbel "arch_ratio"): its call sites are worth prewarming


class Cell(nn.Module):
    def __init__(self, C_in, C_out=None, down=True, width_mult_list=[1.]):
        super(Cell, self).__init__()
        self._C_in = C_in
        if C_out is None: C_out = C_in
        self._C_out = C_out
        self._down = down
        self._width_mult_list = width_mult_list
        self._op = MixedOp(C_in, C_out, width_mult_list=width_mult_list)
        if self._down:
            self.downsample = MixedOp(C_in, C_in * 2, stride=2, width_mult_list=width_mult_list)

    def forward(self, input, alphas, ratios):
        # ratios: (in, out, down)
        out = self._op(input, alphas, (ratios[0], ratios[1]))
        assert (self._down and (ratios[2] is not None)) or ((not self._down) and (ratios[2] is None))
        down = self.downsample(input, alphas, (ratios[0], ratios[2])) if self._down else None
        return out, down

    def forward_latency(self, size, alphas, ratios):
        out = self._op.forward_latency(size, alphas, (ratios[0], ratios[1]))
        assert (self._down and (ratios[2] is not None)) or ((not self._down) and (ratios[2] is None))
        down = self.downsample.forward_latency(size, alphas, (ratios[0], ratios[2])) if self._down else None
        return out, down


def _positive_table(betas, key=None, cache=None):
    """[(b0 > 0, b1 > 0), ...] per beta row (reference :326-328 tests `betas[..] > 0` once per cell: one device sync each).  Under hipGraph
    capture (no host sync allowed) the softmax outputs are taken as positive, which they always are.

    Round 5: the supernet step is HOST-bound (tools/host_vs_device.py), and this read was a full device drain at the head of every eager
    forward - the host stood still until the replayed passes before it had finished, then the device stood still while the host caught up.
    `cache` (a dict owned by the network) / `key` (arch index + the beta parameters' version counters): the table is read back ONCE per value
    of the parameters - never again in a pretrain run, where they are frozen.  When the parameters have changed (the architect's Adam
    step), the previous table is used and the new values are CHECKED asynchronously: `(b > 0).all()` goes to pinned host memory behind an
    event, and the flag of the previous check is examined here without blocking; a softmax output can only become 0 when two logits differ
    by > 87, i.e. after ~3e5 Adam steps of 3e-4 in one direction, so the optimistic table is the exact one in any run one can afford - and
    if a check ever fails, this function says so and falls back to the blocking read for good."""
    live = betas[1:]
    on_gpu = bool(live) and live[0].is_cuda
    if on_gpu and torch.cuda.is_current_stream_capturing():
        return [None] + [[[True, True]] * b.shape[0] for b in live]
    if cache is None or key is None or not live or cache.get("blocking"):
        return [None] + [(b > 0).tolist() for b in live]
    # examine finished asynchronous checks (never blocks)
    pending = cache.setdefault("pending", [])
    while pending and (pending[0][0] is None or pending[0][0].query()):
        _, flag = pending.pop(0)
        if not bool(flag.item()):                   # (host tensor: no device access)
            import warnings
            warnings.warn("fasterseg_amd: a softmaxed beta reached 0 - switching to the blocking per-forward read of the beta tables")
            cache["blocking"] = True
            return [None] + [(b > 0).tolist() for b in live]
    hit = cache.get("table")
    if hit is not None and hit[0] == key:
        return hit[1]
    shapes = tuple(b.shape[0] for b in live)
    if hit is not None and hit[2] == shapes and len(pending) < 8:
        # new parameter values: keep the table, verify the new values behind the launch stream
        ok = torch.stack([(b > 0).all() for b in live]).all()
        if on_gpu:
            flag = torch.empty((), dtype=torch.bool).pin_memory()
            flag.copy_(ok, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
        else:                                       # (host tensors: the check is complete as soon as it is made; unit tests drive this path)
            flag, ev = ok.clone(), None
        pending.append((ev, flag))
        cache["table"] = (key, hit[1], shapes)
        return hit[1]
    table = [None] + [(b > 0).tolist() for b in live]       # first use: one blocking read
    cache["table"] = (key, table, shapes)
    return table


def _weighted_sum(weights, tensors):
    """sum(w * t) over the non-None tensors with device-resident scalar weights (model_search.py:331-332)."""
    live = [k for k, t in enumerate(tensors) if t is not None]
    if not live:
        return 0
    if len(live) == len(tensors):
        return FN.weighted_sum(list(tensors), weights)
    acc = None
    for k in live:
        acc = FN.scale_accumulate(acc, tensors[k], weights[k])
    return acc


class Network_Multi_Path(LatencyModelMixin, nn.Module):
    def __init__(self, num_classes=19, layers=16, criterion=nn.CrossEntropyLoss(ignore_index=-1), Fch=12, width_mult_list=[1., ],
                 prun_modes=['arch_ratio', ], stem_head_width=[(1., 1.), ]):
        super(Network_Multi_Path, self).__init__()
        self._num_classes = num_classes
        assert layers >= 3
        self._layers = layers
        self._criterion = criterion
        self._Fch = Fch
        self._width_mult_list = width_mult_list
        self._prun_modes = prun_modes
        self.prun_mode = None  # prun_mode is higher priority than _prun_modes
        self._stem_head_width = stem_head_width
        self._flops = 0
        self._params = 0
        nf = self.num_filters

        self.stem = nn.ModuleList([
            nn.Sequential(
                ConvNorm(3, nf(2, sr) * 2, kernel_size=3, stride=2, padding=1, bias=False, groups=1, slimmable=False),
                BasicResidual2x(nf(2, sr) * 2, nf(4, sr) * 2, kernel_size=3, stride=2, groups=1, slimmable=False),
                BasicResidual2x(nf(4, sr) * 2, nf(8, sr), kernel_size=3, stride=2, groups=1, slimmable=False)
            ) for sr, _ in self._stem_head_width])

        self.cells = nn.ModuleList()
        for l in range(layers):
            last = (l == layers - 1)
            scales = 1 if l == 0 else (2 if l == 1 else 3)
            row = nn.ModuleList()
            for s in range(scales):
                # a cell can down-sample unless it sits on the coarsest scale or in the last layer
                down = (not last) and s < 2
                row.append(Cell(nf(8 * 2 ** s), down=down, width_mult_list=width_mult_list))
            self.cells.append(row)

        def cn(cin, cout, k, hr):
            return ConvNorm(nf(cin, hr), nf(cout, hr), kernel_size=k, padding=(1 if k == 3 else None), bias=False, groups=1,
                            slimmable=False)
        self.refine32 = nn.ModuleList([nn.ModuleList([cn(32, 16, 1, hr), cn(32, 16, 3, hr), cn(16, 8, 1, hr), cn(16, 8, 3, hr)])
                                       for _, hr in self._stem_head_width])
        self.refine16 = nn.ModuleList([nn.ModuleList([cn(16, 8, 1, hr), cn(16, 8, 3, hr)]) for _, hr in self._stem_head_width])

        self.head0 = nn.ModuleList([Head(nf(8, hr), num_classes, False) for _, hr in self._stem_head_width])
        self.head1 = nn.ModuleList([Head(nf(8, hr), num_classes, False) for _, hr in self._stem_head_width])
        self.head2 = nn.ModuleList([Head(nf(8, hr), num_classes, False) for _, hr in self._stem_head_width])
        self.head02 = nn.ModuleList([Head(nf(8, hr) * 2, num_classes, False) for _, hr in self._stem_head_width])
        self.head12 = nn.ModuleList([Head(nf(8, hr) * 2, num_classes, False) for _, hr in self._stem_head_width])

        # contains arch_param names: {"alphas": alphas, "betas": betas, "ratios": ratios}
        self._arch_names = []
        self._arch_parameters = []
        for i in range(len(self._prun_modes)):
            arch_name, arch_param = self._build_arch_parameters(i)
            self._arch_names.append(arch_name)
            self._arch_parameters.append(arch_param)
            self._reset_arch_parameters(i)
        # switch set of arch if we have more than 1 arch
        self.arch_idx = 0

    def num_filters(self, scale, width=1.0):
        return int(np.round(scale * self._Fch * width))

    # ------------------------------------------------------------------------------------------------------
    def sample_prun_ratio(self, mode="arch_ratio", read_indices=True):
        '''
        mode: "min"|"max"|"random"|"arch_ratio"(default)
        read_indices=False (arch_ratio): leave the sampled width indices on the device (`_fs_index_t`), no host sync.
        '''
        assert mode in ["min", "max", "random", "arch_ratio"]
        counts = (self._layers - 1, self._layers - 1, self._layers - 2)
        if mode == "arch_ratio":
            names = self._arch_names[self.arch_idx]["ratios"]
            params = [getattr(self, names[s]) for s in range(3)]
            assert all(p.shape[0] == counts[s] for s, p in enumerate(params))
            # all slots (scale-major, layer-minor: the reference's draw order) in ONE batch; per-slot views for the per-cell consumers.
            # The reference reads `ratio.argmax()` on the host once per MixedOp (model_search.py:64-65): ~230 device syncs per
            # forward, each draining the launch queue.  Read all sampled indices back in ONE transfer instead - and (round 5) from a
            # SIDE stream that only waits for the last write of the architecture parameters (note_arch_update), not for everything the
            # step has queued on the launch stream: the widths are needed on the host to shape the pass, the step is host-bound, and a
            # drain here left the host idle for ~10 ms per Gumbel pass (profiles/r05_host_vs_device_c3_c5_c2.txt).
            host = None
            on_gpu = params[0].is_cuda
            if read_indices and on_gpu and not torch.cuda.is_current_stream_capturing() and _SAMPLE_STREAM:
                main = torch.cuda.current_stream()
                side = self.__dict__.get("_sample_stream")
                if side is None or side.device != params[0].device:
                    side = self.__dict__["_sample_stream"] = torch.cuda.Stream(device=params[0].device)
                upd = self.__dict__.get("_arch_update")
                if upd is not None and upd[1] == self._arch_versions():
                    side.wait_event(upd[0])             # the parameters' last writer
                else:
                    side.wait_stream(main)              # written by something that did not say so: behind everything queued (a drain)
                with torch.cuda.stream(side):
                    out, ind = gumbel_softmax_rows(F.log_softmax(torch.cat(params), dim=-1))
                    host = ind.tolist()                 # waits for `side` only
                main.wait_stream(side)
                out.record_stream(main)
                ind.record_stream(main)
            else:
                out, ind = gumbel_softmax_rows(F.log_softmax(torch.cat(params), dim=-1))
                if read_indices and not (on_gpu and torch.cuda.is_current_stream_capturing()):
                    host = ind.tolist()
            rows = out.unbind(0)
            ratios, n = _SampledRatios(), 0
            for s in range(3):
                scale = []
                for _ in range(counts[s]):
                    r = rows[n]
                    r._fs_index_t = ind[n:n + 1]      # device-side arg-max (a view: no launch)
                    if host is not None:
                        r._fs_index = host[n]
                    scale.append(r)
                    n += 1
                ratios.append(scale)
            ratios.stacked, ratios.index = out, ind
            return ratios
        if mode == "random":      # same draw order as the reference: all of scale 0, then scale 1, then scale 2
            return [[np.random.choice(self._width_mult_list) for _ in range(counts[s])] for s in range(3)]
        w = self._width_mult_list[0] if mode == "min" else self._width_mult_list[-1]
        return [[w] * counts[s] for s in range(3)]

    def _arch_versions(self):
        return tuple(p._version for group in self._arch_parameters for p in group) + tuple(p.data_ptr() for p in self._arch_parameters[0][:1])

    def note_arch_update(self):
        """Call right after the architecture parameters were written on the current stream (the architect's optimizer step): the next
        host read of sampled widths waits for THIS point of the stream instead of for everything queued behind it."""
        p = self._arch_parameters[0][0]
        if p.is_cuda:
            ev = torch.cuda.Event()
            ev.record()
            self.__dict__["_arch_update"] = (ev, self._arch_versions())

    def _arch_tensors(self, alpha=True, beta=True):
        names = self._arch_names[self.arch_idx]
        if alpha:
            alphas = [F.softmax(getattr(self, n), dim=-1) for n in names["alphas"]]
        else:
            alphas = [torch.ones_like(getattr(self, n)) * 1. / len(PRIMITIVES) for n in names["alphas"]]
        if beta:
            betas = [None] + [F.softmax(getattr(self, n), dim=-1) for n in names["betas"]]
        else:
            betas = [None] + [torch.ones_like(getattr(self, n)) * 1. / 2 for n in names["betas"]]
        return alphas, betas

    def _cell_ratio(self, i, j, ratios, k=None):
        """(in, out, down) width spec of cell (layer i, scale j) — reference forward :300-316.  k: architecture index (default: the active one)."""
        shw = self._stem_head_width[self.arch_idx if k is None else k]
        if i == 0 and j == 0:
            return (shw[0], ratios[j][i - j], ratios[j + 1][i - j])
        if i == self._layers - 1:
            return (ratios[j][i - j - 1] if j == 0 else ratios[j][i - j], shw[1], None)
        if j == 2:
            return (ratios[j][i - j], ratios[j][i - j + 1], None)
        if j == 0:
            return (ratios[j][i - j - 1], ratios[j][i - j], ratios[j + 1][i - j])
        return (ratios[j][i - j], ratios[j][i - j + 1], ratios[j + 1][i - j])

    def _coefficient_rows(self, alphas, ratios, mode):
        """{(layer, scale, 0 op | 1 downsample): _PreCoef} = alpha row * score(in width) * score(out width) (reference :64-78) for
        every MixedOp of a pass in a handful of batched ops: one gather of the alpha rows, one gather of the Gumbel scores,
        two multiplies, one unbind - whose backward is one stack instead of a zero-fill + copy + add per MixedOp and pass
        (~2 k of the architecture step's launches).  None while capturing the first time (index tensors need a host copy)."""
        plans = self.__dict__.setdefault("_coef_plans", {})
        plan = plans.get(self.arch_idx)
        dev = alphas[0].device
        if plan is None:
            if dev.type == "cuda" and torch.cuda.is_current_stream_capturing():
                return None
            counts = (self._layers - 1, self._layers - 1, self._layers - 2)
            offs = (0, counts[0], counts[0] + counts[1])
            n_slots = sum(counts)
            probe = [[("slot", offs[s] + n) for n in range(counts[s])] for s in range(3)]
            alpha_off = (0, self._layers, 2 * self._layers - 1)
            keys, rows, s_in, s_out = [], [], [], []
            slot = lambda r: r[1] if isinstance(r, tuple) else n_slots
            for i, cells in enumerate(self.cells):
                for j, cell in enumerate(cells):
                    r = self._cell_ratio(i, j, probe)
                    for which, out in ((0, r[1]), (1, r[2])):
                        if which == 1 and not cell._down:
                            continue
                        keys.append((i, j, which)); rows.append(alpha_off[j] + i - j); s_in.append(slot(r[0])); s_out.append(slot(out))
            plan = plans[self.arch_idx] = dict(keys=keys, rows=torch.tensor(rows, device=dev), s_in=torch.tensor(s_in, device=dev),
                                               s_out=torch.tensor(s_out, device=dev))
        A = torch.cat(list(alphas))[plan["rows"]]
        if mode == "arch_ratio":
            if getattr(ratios, "stacked", None) is not None:
                R, idx = ratios.stacked, ratios.index
            else:
                flat = [r for scale in ratios for r in scale]
                R, idx = torch.stack(flat), torch.cat([r._fs_index_t for r in flat])
            score = R.gather(1, idx[:, None]).squeeze(1)
            score = torch.cat([score, score.new_ones(1)])
            A = A * score[plan["s_in"]][:, None] * score[plan["s_out"]][:, None]
        return {key: _PreCoef(row) for key, row in zip(plan["keys"], A.unbind(0))}

    def forward(self, input):
        gen = self._forward_steps(input)
        try:
            tasks, dest_fn, _ = next(gen)
            while True:
                tasks, dest_fn, _ = gen.send(_run_tasks(tasks, dest_fn))
        except StopIteration as done:
            return done.value

    def forward_multi(self, input, specs, batch_tails=False):
        """Several passes of `_loss` (reference :392-411: the same batch through the supernet once per width mode) evaluated TOGETHER,
        layer by layer: the MixedOp evaluations of all passes at one layer only depend on the previous layer of their own pass, so they
        go to ONE layer call (_run_tasks -> fs_exec_program_group) and every kernel of the layer is one grouped launch over the passes'
        problems - the supernet step is the sum of its kernel durations at ~4 us of boundary per launch, and a pass alone leaves most
        of the chip idle.  specs: [(arch_idx or None = keep, prun_mode)] in the reference's order; returns the passes' logits tuples.
        The arithmetic of every pass is what `forward` computes; what is shared between passes is state, and it is kept in the
        reference's order: width draws happen pass by pass before the first layer (the host RNG streams are consumed as by sequential
        forwards), stems / refinement / heads run pass by pass, and two evaluations of one MixedOp at the same OUTPUT width - the only
        way two passes meet in a BatchNorm's running statistics - are never put into one grouped launch (_run_tasks issues them in pass
        order).  Weight gradients of the passes add into the same slices with fp32 atomics, as the passes of one backward already do."""
        global _SAMPLING_PASS
        # batch_tails (two passes of one architecture index, training on the GPU): refinement and heads - the same modules on maps of the
        # same shape in both passes - run ONCE on the two passes' maps concatenated along the batch, their BatchNorms normalising the
        # halves independently and updating the running statistics half after half (functional.bn_groups: the arithmetic of two
        # evaluations in pass order, as for a doubly-fed cell); returns ONE JointLogits instead of the passes' tuples.
        joint_tail = (batch_tails and _TAIL_BATCH and len(specs) == 2 and specs[1][0] is None and self.training and input.is_cuda
                      and torch.is_grad_enabled())
        gens, reqs = [], []
        shared_stem = None
        for n, (arch_idx, mode) in enumerate(specs):
            if arch_idx is not None:
                self.arch_idx = arch_idx
            self.prun_mode = mode
            if joint_tail and _STEM_SHARE:
                # The stem (fixed width per architecture index, reference forward :286-288) maps the same images through the same weights
                # in both passes: evaluated ONCE, used by both (autograd adds the two passes' gradients of its output before the one
                # backward: J^T g1 + J^T g2 = J^T (g1 + g2)).  Its BatchNorms owe the second pass's momentum update, with the same batch
                # statistics b: r1 = (1 - m) r0 + m b, r2 = (1 - m) r1 + m b = r1 + (1 - m) (r1 - r0); num_batches_tracked += 1.
                if n == 0:
                    shared_stem = self._stem_once(self.stem[self.arch_idx], input)
            g = self._forward_steps(input, defer_tail=joint_tail, stem_out=shared_stem)
            gens.append(g)
            reqs.append(next(g))            # width draws + stem of this pass, up to its first layer's task list
        results = [None] * len(gens)
        live = list(range(len(gens)))
        while live:
            tasks, spans = [], []
            for p in live:
                t, d, _ = reqs[p]
                spans.append((p, len(tasks), len(t), d))
                tasks += t

            def dest_fn(t, shape, dtype, device, spans=spans):
                for _, start, n, d in spans:
                    if start <= t < start + n:
                        return d(t - start, shape, dtype, device) if d is not None else None
                return None
            _SAMPLING_PASS = any(reqs[p][2] for p in live)
            outs = _run_tasks(tasks, dest_fn if any(sp[3] is not None for sp in spans) else None)
            nxt = []
            for p, start, n, _ in spans:
                try:
                    reqs[p] = gens[p].send(outs[start:start + n])
                    nxt.append(p)
                except StopIteration as done:
                    results[p] = done.value
            live = nxt
        if joint_tail:
            a, b = results
            assert a.k == b.k
            with FN.bn_groups(2):
                logits = self._tail(a.k, FN.batch_pair(a.o0, b.o0), FN.batch_pair(a.o1, b.o1), FN.batch_pair(a.o2, b.o2))
            return JointLogits(logits, 2)
        return results

    def _stem_once(self, stem, input):
        """stem(input) with the BatchNorm running statistics of TWO identical evaluations (forward_multi)."""
        plan = self.__dict__.setdefault("_stem_bn", {}).get(id(stem))
        if plan is None:
            bns = [m for m in stem.modules() if isinstance(m, nn.BatchNorm2d) and m.track_running_stats]
            assert all(m.momentum == bns[0].momentum and m.momentum is not None for m in bns)
            plan = self.__dict__["_stem_bn"][id(stem)] = ([t for m in bns for t in (m.running_mean, m.running_var)],
                                                          [m.num_batches_tracked for m in bns], float(bns[0].momentum))
        floats, counts, momentum = plan
        with torch.no_grad():
            before = torch._foreach_add(floats, 0.0)
        out = stem(input)
        with torch.no_grad():
            delta = torch._foreach_sub(floats, before)
            torch._foreach_add_(floats, delta, alpha=1.0 - momentum)
            torch._foreach_add_(counts, 1)
        return out

    def _forward_steps(self, input, defer_tail=False, stem_out=None):
        """`forward` as a generator: yields (MixedOp tasks of the next layer, destination planner, pass draws its widths) and is sent
        their outputs (_run_tasks); returns the logits.  Between a yield and its send another pass may have run (forward_multi): everything
        the pass needs afterwards is local, and the two module-level switches are put back on resume."""
        k = self.arch_idx
        stem = self.stem[k]
        alphas, betas = self._arch_tensors()
        mode = self.prun_mode if self.prun_mode is not None else self._prun_modes[k]
        global _SAMPLING_PASS
        sampling = mode in ("random", "arch_ratio")
        _SAMPLING_PASS = sampling
        ratios = self.sample_prun_ratio(mode=mode)
        coef_rows = self._coefficient_rows(alphas, ratios, mode) if _BATCHED_COEFS else None
        # one host read of the whole beta tables instead of one implicit sync per cell (reference :326-328)
        names = self._arch_names[k]["betas"]
        on_gpu = betas[1].is_cuda          # (host tensors: the plain read costs nothing and is exact, as in the reference)
        beta_pos = _positive_table(betas, (k,) + tuple(getattr(self, n)._version for n in names) + tuple(getattr(self, n).data_ptr() for n in names),
                                   self.__dict__.setdefault("_beta_pos_cache", {}).setdefault(k, {}) if on_gpu else None)
        # rows handed out by ONE unbind per table: `betas[j][row]` per cell is a select whose backward is a zero-fill + copy + add per cell
        beta_rows = [None] + [b.unbind(0) for b in betas[1:]]

        out_prev = [[stem(input) if stem_out is None else stem_out, None]]  # stem: one cell
        probe = [[_SAMPLED] * len(r) for r in ratios]          # which entries of a cell's (in, out, down) widths are sampled
        # Every cell output has ONE consumer cell in the next layer; a consumer fed from two scales evaluates once on both inputs
        # concatenated along the batch.  The joint buffers are planned one layer ahead so that the producers write into their halves
        # (functional.PairBuffers: no copy launches), and the beta merges of a layer go out as one grouped launch.
        pair_plan = FN.PairBuffers() if (_PAIR_BATCH and _PAIR_DIRECT and self.training and input.is_cuda and torch.is_grad_enabled()) else None
        FN._pair_buffers = pair_plan
        n_layers = len(self.cells)

        def will_pair(i2, j2):
            if pair_plan is None or i2 >= n_layers or j2 >= len(self.cells[i2]) or j2 == 0 or i2 == j2:
                return False
            bp = beta_pos[j2][i2 - j2 - 1]
            return bool(bp[0] and bp[1])

        def dest_of(i, j, which, shape, dtype, device):
            """Where output `which` (0 keep, 1 down) of cell j of layer i should be written: its half of the consumer's joint buffer."""
            jn = j + which
            if not will_pair(i + 1, jn):
                return None
            return pair_plan.half((i + 1, jn), 1 - which, tuple(shape), dtype, device)
        # i: layer | j: scale
        for i, cells in enumerate(self.cells):
            # every MixedOp evaluation of this layer (reference :303-333: cell(out_prev[..], alpha, ratio) = its `_op` and,
            # if the cell can down-sample, its `downsample`), listed first and run together (see _run_tasks)
            tasks, slots = [], []
            for j, cell in enumerate(cells):
                alpha = alphas[j][i - j] if coef_rows is None else None
                ratio = self._cell_ratio(i, j, ratios, k)
                assert (cell._down and (ratio[2] is not None)) or ((not cell._down) and (ratio[2] is None))
                if "_ratio_sampled" not in cell._op.__dict__:
                    which = [r is _SAMPLED for r in self._cell_ratio(i, j, probe, k)]
                    cell._op.__dict__["_ratio_sampled"] = (which[0], which[1])
                    if cell._down:
                        cell.downsample.__dict__["_ratio_sampled"] = (which[0], which[2])
                # sources -- 0: from down; 1: from keep
                if j == 0:
                    srcs = [(1, out_prev[0][0])]
                elif i == j:
                    srcs = [(0, out_prev[j - 1][1])]
                else:
                    srcs = [(0, out_prev[j - 1][1])] if beta_pos[j][i - j - 1][0] else []
                    if beta_pos[j][i - j - 1][1]:
                        srcs.append((1, out_prev[j][0]))
                groups = 1
                if len(srcs) == 2 and _PAIR_BATCH and self.training and srcs[0][1].is_cuda:
                    srcs, groups = [(2, FN.batch_pair(srcs[0][1], srcs[1][1]))], 2          # tag 2: both inputs in one batch
                a_op = alpha if coef_rows is None else coef_rows[(i, j, 0)]
                a_down = alpha if coef_rows is None else coef_rows.get((i, j, 1))
                for tag, x in srcs:
                    tasks.append((cell._op, x, a_op, (ratio[0], ratio[1]), groups))
                    slots.append((j, tag, 0))
                    if cell._down:
                        tasks.append((cell.downsample, x, a_down, (ratio[0], ratio[2]), groups))
                        slots.append((j, tag, 1))
            def task_dest(t, shape, dtype, device, i=i, slots=slots):
                j, tag, which = slots[t]
                if tag == 2:                    # a pair-batched evaluation: its beta merge (below) is what the next layer consumes
                    return None
                if not (j == 0 or i == j):      # one of two separately evaluated sources: their weighted sum is the output
                    return None
                return dest_of(i, j, which, shape, dtype, device)
            outs = yield (tasks, task_dest if pair_plan is not None else None, sampling)
            FN._pair_buffers = pair_plan            # (another pass may have run since the yield)
            _SAMPLING_PASS = sampling
            res = dict(zip(slots, outs))
            out = [None] * len(cells)
            merges = []                         # (j, which, x, beta row): the pair-batched cells' beta merges, one grouped launch
            for j, cell in enumerate(cells):
                if j == 0:
                    out[j] = (res[(0, 1, 0)], res.get((0, 1, 1)))
                elif i == j:
                    out[j] = (res[(j, 0, 0)], res.get((j, 0, 1)))
                else:
                    b = beta_rows[j][i - j - 1]
                    if (j, 2, 0) in res:
                        out[j] = [None, 0]
                        merges.append((j, 0, res[(j, 2, 0)], b))
                        if (j, 2, 1) in res:
                            merges.append((j, 1, res[(j, 2, 1)], b))
                    else:
                        out[j] = (_weighted_sum(b, [res.get((j, 0, 0)), res.get((j, 1, 0))]),
                                  _weighted_sum(b, [res.get((j, 0, 1)), res.get((j, 1, 1))]))
            if merges:
                dests = None
                if pair_plan is not None:
                    dests = []
                    for j, which, x, b in merges:
                        n2, C, H, W = x.shape
                        dests.append(dest_of(i, j, which, (n2 // 2, C, H, W), x.dtype, x.device))
                if _MERGE_GROUP and len(merges) > 1:
                    merged = FN.pair_merge_group([m[2] for m in merges], [m[3] for m in merges], dests)
                else:
                    merged = [FN.pair_merge(m[2], m[3], dests[k] if dests else None) for k, m in enumerate(merges)]
                for (j, which, x, b), y in zip(merges, merged):
                    out[j][which] = y
            out_prev = out
        FN._pair_buffers = None
        if defer_tail:          # forward_multi evaluates the refinement / heads of its passes together (_tail on the batched maps)
            return _DeferredTail(k, out[0][0], out[1][0], out[2][0])
        return self._tail(k, out[0][0], out[1][0], out[2][0])

    def _tail(self, k, o0, o1, o2):
        """Refinement + the five heads (reference forward :335-353) on the last layer's three maps."""
        refine16, refine32 = self.refine16[k], self.refine32[k]
        ###################################
        up2 = lambda t: FN.interpolate(t, scale_factor=2)
        out0 = o0
        out1 = refine16[1](FN.cat([up2(refine16[0](o1)), o0]))
        out2 = refine32[1](FN.cat([up2(refine32[0](o2)), o1]))
        out2 = refine32[3](FN.cat([up2(refine32[2](out2)), o0]))

        preds = [self.head0[k](out0), self.head1[k](out1), self.head2[k](out2),
                 self.head02[k](FN.cat([out0, out2])), self.head12[k](FN.cat([out1, out2]))]
        if not self.training:
            return tuple(FN.interpolate(p, scale_factor=8, out_nchw=1) for p in preds)
        # train mode: 1/8-resolution logits (labels are down-sampled x8, search/dataloader.py:25); hand the loss a
        # contiguous NCHW fp32 tensor like the reference does
        return tuple(FN.interpolate(p, size=(p.size(2), p.size(3)), out_nchw=1) for p in preds)
        ###################################

    def forward_latency(self, size, alpha=True, beta=True, ratio=True):
        if not beta and _LINEAR_LATENCY:
            fast = self._forward_latency_linear(size, alpha, ratio)
            if fast is not None:
                return fast
        if beta and not alpha and not ratio and _LINEAR_LATENCY:
            return self._forward_latency_beta(size)
        k = self.arch_idx
        stem = self.stem[k]
        alphas, betas = self._arch_tensors(alpha, beta)
        if ratio:
            mode = self.prun_mode if self.prun_mode is not None else self._prun_modes[k]
            ratios = self.sample_prun_ratio(mode=mode)
        else:
            ratios = self.sample_prun_ratio(mode='max')
        beta_pos = _positive_table(betas)

        stem_latency = 0
        for m in stem:
            latency, size = m.forward_latency(size)
            stem_latency = stem_latency + latency
        out_prev = [[size, None]]  # stem: one cell
        latency_total = [[stem_latency, 0], [0, 0], [0, 0]]  # (out, down)

        for i, cells in enumerate(self.cells):
            out = []
            latency = []
            for j, cell in enumerate(cells):
                a = alphas[j][i - j]
                r = self._cell_ratio(i, j, ratios)
                if j == 0 or i == j:
                    src = out_prev[0][0] if j == 0 else out_prev[j - 1][1]
                    o, d = cell.forward_latency(src, a, r)
                    out.append((o[1], d[1] if d is not None else None))
                    latency.append([o[0], d[0] if d is not None else None])
                else:
                    out0 = down0 = out1 = down1 = None
                    if beta_pos[j][i - j - 1][0]:      # from down
                        out0, down0 = cell.forward_latency(out_prev[j - 1][1], a, r)
                    if beta_pos[j][i - j - 1][1]:      # from keep
                        out1, down1 = cell.forward_latency(out_prev[j][0], a, r)
                    assert (out0 is None and out1 is None) or out0[1] == out1[1]
                    assert (down0 is None and down1 is None) or down0[1] == down1[1]
                    out.append((out0[1], down0[1] if down0 is not None else None))
                    b = betas[j][i - j - 1]
                    latency.append([
                        sum(w * o for w, o in zip(b, [out0[0], out1[0]])),
                        sum(w * d if d is not None else 0 for w, d in zip(b, [down0[0] if down0 is not None else None,
                                                                           down1[0] if down1 is not None else None])),
                    ])
            out_prev = out
            for ii, lat in enumerate(latency):
                # layer: i | scale: ii.  NOTE the mixing weights below use the loop variable `j` left over from the
                # scale loop (= the last scale of this layer), exactly as the reference does (:468-469).
                if ii == 0:
                    if lat[0] is not None: latency_total[ii][0] = latency_total[ii][0] + lat[0]
                    if lat[1] is not None: latency_total[ii][1] = latency_total[ii][0] + lat[1]
                elif i == ii:
                    if lat[0] is not None: latency_total[ii][0] = latency_total[ii - 1][1] + lat[0]
                    if lat[1] is not None: latency_total[ii][1] = latency_total[ii - 1][1] + lat[1]
                else:
                    bw = betas[j][i - j - 1]
                    if lat[0] is not None: latency_total[ii][0] = bw[1] * latency_total[ii][0] + bw[0] * latency_total[ii - 1][1] + lat[0]
                    if lat[1] is not None: latency_total[ii][1] = bw[1] * latency_total[ii][0] + bw[0] * latency_total[ii - 1][1] + lat[1]
        ###################################
        return sum([latency_total[0][0], latency_total[1][0], latency_total[2][0]])
        ###################################

    def _loss(self, input, target, pretrain=False):
        loss = 0
        run = lambda: sum(self._criterion(logit, target) for logit in self(input))
        if pretrain is not True:
            # "random width": sampled by gambel softmax
            self.prun_mode = None
            for idx in range(len(self._arch_names)):
                self.arch_idx = idx
                loss = loss + run()
        if len(self._width_mult_list) > 1:
            for mode in (("max", "min", "random", "random") if pretrain == True else ("max", "min")):
                self.prun_mode = mode
                loss = loss + run()
        elif pretrain == True and len(self._width_mult_list) == 1:
            self.prun_mode = "max"
            loss = loss + run()
        return loss

    # ------------------------------------------------------------------------------------------------------
    def _arch_shapes(self, idx):
        num_ops = len(PRIMITIVES)
        num_widths = len(self._width_mult_list) if self._prun_modes[idx] == 'arch_ratio' else 1
        L = self._layers
        return {"alphas": [(L, num_ops), (L - 1, num_ops), (L - 2, num_ops)],
                "betas": [(L - 2, 2), (L - 3, 2)],              # in-degree probs; 0: from down, 1: from keep
                "ratios": [(L - 1, num_widths), (L - 1, num_widths), (L - 2, num_widths)]}

    def _build_arch_parameters(self, idx):
        names = {"alphas": ["alpha_" + str(idx) + "_" + str(s) for s in [0, 1, 2]],
                 "betas": ["beta_" + str(idx) + "_" + str(s) for s in [1, 2]],
                 "ratios": ["ratio_" + str(idx) + "_" + str(s) for s in [0, 1, 2]]}
        shapes = self._arch_shapes(idx)
        params = []
        for kind in ("alphas", "betas", "ratios"):
            for name, shape in zip(names[kind], shapes[kind]):
                setattr(self, name, nn.Parameter(1e-3 * torch.ones(*shape), requires_grad=True))
                params.append(getattr(self, name))
        return names, params

    def _reset_arch_parameters(self, idx):
        shapes = self._arch_shapes(idx)
        for kind in ("alphas", "betas", "ratios"):
            for name, shape in zip(self._arch_names[idx][kind], shapes[kind]):
                p = getattr(self, name)
                p.data = 1e-3 * torch.ones(*shape, device=p.device)
