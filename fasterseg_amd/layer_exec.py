"""Execution of one supernet layer's MixedOp evaluations (split out of model_search.py, round 6).

`run_tasks` takes the evaluations a layer needs - (MixedOp, input, alpha row, (in, out) widths, BatchNorm groups), possibly of several
passes (Network_Multi_Path.forward_multi) - and issues them: as layer calls of the launch-program executor (fs_exec_program_group; the
default, eager and under hipGraph capture), one program per stream lane, or primitive by primitive on forked lanes (the round-3..5 capture
layout).  `conflict_free_chunks` decides which evaluations may share a layer call.  The switches live in model_search (tests and
train_step set them there) and are read at call time.
"""
import torch

from . import functional as FN
from . import kernels as K


class _Switches:
    """model_search's module-level switches, looked up when they are read (the two modules import each other)."""

    def __getattr__(self, name):
        from . import model_search
        return getattr(model_search, name)


ms = _Switches()


_lane_pool = {}


def branch_lanes(stream):
    """Side streams that fork from `stream` (created on first use; call once BEFORE capturing on `stream`, stream creation
    is not a capturable operation)."""
    key = (stream.device, stream.cuda_stream)
    lanes = _lane_pool.get(key)
    if lanes is None:
        lanes = _lane_pool[key] = [torch.cuda.Stream(device=stream.device) for _ in range(max(0, ms._BRANCH_LANES - 1))]
    return lanes


_task_pool = {}


def layer_lanes(stream):
    """Side streams for the primitives of one layer (create BEFORE capturing on `stream`)."""
    key = (stream.device, stream.cuda_stream)
    pool = _task_pool.get(key)
    if pool is None:
        pool = _task_pool[key] = [torch.cuda.Stream(device=stream.device) for _ in range(max(0, ms._LAYER_LANES - 1))]
    return pool


def _eval(op, x, alpha, ratios, groups):
    if groups == 1:
        return op(x, alpha, ratios)
    with FN.bn_groups(groups):
        return op(x, alpha, ratios)


def conflict_free_chunks(items, per, key):
    """Partition `items` (in order) into chunks of at most `per` such that no chunk holds two items of one key, and items of one key
    keep their order across chunks; an item goes into the earliest chunk behind the last one that holds its key.  The layer calls use
    it with key = (MixedOp, output width): two evaluations of one MixedOp at one output width update the same BatchNorm running
    statistics (USBatchNorm2d keeps one BatchNorm per width, reference search/slimmable_ops.py:51-70) and must stay stream-ordered -
    the two sources of a cell that is not pair-batched, two passes of forward_multi that drew the same width."""
    chunks, keys = [], []
    for item in items:
        k = key(item)
        first = 0
        for c in range(len(chunks) - 1, -1, -1):
            if k in keys[c]:
                first = c + 1
                break
        for c in range(first, len(chunks)):
            if len(chunks[c]) < per:
                chunks[c].append(item)
                keys[c].add(k)
                break
        else:
            chunks.append([item])
            keys.append({k})
    return chunks


def run_tasks(tasks, dest_fn=None):
    """tasks: [(mixed_op, x, alpha, ratios, bn_groups)] -> outputs.  dest_fn(task index, out shape, dtype, device): a tensor the task's
    output should be written into (a functional.PairBuffers half), or None - honoured by the grouped launch programs only.  While capturing, every primitive of every task runs on its own
    stream; the alpha-weighted sums follow on the capturing stream after the join.  Eager training passes put whole MixedOp
    programs on side streams."""
    on_gpu = len(tasks) > 1 and tasks[0][1].is_cuda
    capturing = on_gpu and torch.cuda.is_current_stream_capturing()
    eager_lanes = (on_gpu and not capturing and ms._EAGER_LANES > 1 and ms._PROGRAMS and torch.is_grad_enabled() and tasks[0][0].training)
    # bit-reproducible mode: the ordered slab reduction of the weight gradient finishes with a plain read-modify-write of the gradient,
    # which needs the launches that touch one tensor stream-ordered (the same cell._op on two lanes would race: ADVICE r3) - no lanes
    ordered = on_gpu and K.deterministic_on()
    if ordered or not ((capturing and ms._LAYER_LANES > 1) or eager_lanes):
        return [_eval(op, x, alpha, ratios, g) for op, x, alpha, ratios, g in tasks]
    main = torch.cuda.current_stream()
    pool = layer_lanes(main)
    if not capturing:
        pool = pool[:ms._EAGER_LANES]
    used, slot, pending = [], 0, []
    grouped = []                 # (index into pending, x, coef, prog) of the tasks that run from launch programs
    crossing = []                # (tensor, lane) of eager passes: inputs made on `main` and read on a lane, outputs made on a lane

    def hand_over(t, lane):
        """Caching-allocator bookkeeping of a tensor that crosses streams in an eager pass: without it the block returns to
        its home stream's pool when the last reference dies and can be rewritten while the other stream's kernels still read it."""
        if ms._RECORD_STREAM and not capturing and torch.is_tensor(t) and t.is_cuda and lane is not main:
            t.record_stream(lane)

    def lane_for(k):
        if not pool:                                       # FS_LAYER_LANES=1: everything on the current stream
            return main
        lane = pool[k % len(pool)]
        if lane not in used:
            lane.wait_stream(main)                         # fork: the previous layer's outputs are complete on `main`
            used.append(lane)
        return lane
    for op, x, alpha, ratios, groups in tasks:
        widths = [None, None]
        # (the widths are applied to the five primitives' modules only when something reads them there: a launch-program cache miss or
        # the per-module path - 23 us of attribute stores per MixedOp otherwise, on a host-bound step)
        coef = op._coefficients(x, alpha, ratios, widths, set_ratio=False)
        prog = None
        if ms._PROGRAMS and (ms._CAPTURE_PROGRAMS or not capturing) and op.training and torch.is_grad_enabled():
            with FN.bn_groups(groups):
                prog = op._program(FN.as_nhwc(x), coef, widths[0], widths[1])
        if prog is None:
            # train_step's fast phase flip leaves the cell weights trainable during the architecture phase: correct only while every
            # MixedOp runs from a launch program (which asks its probe weight); a per-module fallback would compute - and accumulate into
            # the live flat gradient - weight gradients nobody wants
            assert not (ms.FAST_PHASE_ACTIVE and op.training and torch.is_grad_enabled() and not op._ops[1].conv1.weight.requires_grad
                        and any(p.requires_grad for p in op._ops[3].parameters())), \
                "a MixedOp fell back to the per-module path while only the probe weights carry the phase (FS_FAST_PHASE=0 to disable)"
            op.set_prun_ratio((widths[0], widths[1]))
            if FN._touch_log is not None and op.training and torch.is_grad_enabled():
                FN._touch_log.append(None)          # a MixedOp off the launch programs: its gradient writes are not in the log
        group = ms._GROUP_PROGRAMS and (not capturing or ms._GROUP_CAPTURE)
        if ms.MIMIC_CAPTURE and not capturing and not ms._SAMPLING_PASS:
            group = group and bool(ms._GROUP_CAPTURE)
        if prog is not None and group:
            dest = dest_fn(len(pending), prog.out_shape, x.dtype, x.device) if dest_fn is not None else None
            grouped.append((len(pending), FN.as_nhwc(x), coef, prog, dest, (id(op), widths[1])))
            pending.append(None)
            continue
        if prog is not None:          # the whole MixedOp (five primitives, their sum, and in backward the sum of the five input
            lane = lane_for(slot)                          # gradients) as one launch program on one lane
            xn = FN.as_nhwc(x)
            hand_over(xn, lane)
            hand_over(coef, lane)
            with torch.cuda.stream(lane):
                out = FN.mixed_op_program(xn, coef, prog)
            crossing.append((out, lane))
            pending.append((out, None))
            slot += 1
            continue
        if not capturing:             # no program (nothing to differentiate, gradients outside the sink): per-module path, in place
            with FN.bn_groups(groups):
                pending.append((FN.weighted_sum([prim(x) for prim in op._ops], coef), None))
            continue
        outs = []
        for prim in op._ops:
            with torch.cuda.stream(lane_for(slot)), FN.bn_groups(groups):
                outs.append(prim(x))
            slot += 1
        pending.append((outs, coef))
    if grouped:
        from .program import MAX_GROUP
        buckets = {}
        for item in grouped:                                # one call needs one dtype; the executor sorts out everything else
            buckets.setdefault(item[1].dtype, []).append(item)
        for items in buckets.values():
            per = MAX_GROUP
            if not capturing and ms._LAYER_SPLIT > 1:
                per = min(MAX_GROUP, max(1, -(-len(items) // ms._LAYER_SPLIT)))
            for chunk in conflict_free_chunks(items, per, lambda item: item[5]):
                if capturing and ms._GROUP_CAPTURE == 2:
                    with torch.cuda.stream(lane_for(0)):
                        outs = FN.mixed_op_program_group([c[1] for c in chunk], [c[2] for c in chunk], [c[3] for c in chunk], [c[4] for c in chunk])
                elif capturing and ms._GROUP_CAPTURE >= 3:      # (crashes hipStreamEndCapture: kept for the reproduction only)
                    with torch.cuda.stream(lane_for(slot % (ms._GROUP_CAPTURE - 1))):
                        outs = FN.mixed_op_program_group([c[1] for c in chunk], [c[2] for c in chunk], [c[3] for c in chunk], [c[4] for c in chunk])
                    slot += 1
                elif capturing or ms._LAYER_SPLIT == 1:        # on the current stream itself
                    outs = FN.mixed_op_program_group([c[1] for c in chunk], [c[2] for c in chunk], [c[3] for c in chunk], [c[4] for c in chunk])
                else:
                    lane = lane_for(slot)                   # FS_LAYER_SPLIT calls side by side on the lanes
                    for c in chunk:
                        hand_over(c[1], lane)
                        hand_over(c[2], lane)
                    with torch.cuda.stream(lane):
                        outs = FN.mixed_op_program_group([c[1] for c in chunk], [c[2] for c in chunk], [c[3] for c in chunk], [c[4] for c in chunk])
                    crossing.extend((o, lane) for o in outs)
                    slot += 1
                for c, o in zip(chunk, outs):
                    pending[c[0]] = (o, None)
    for lane in used:
        main.wait_stream(lane)                             # one join per layer
    if ms._RECORD_STREAM and not capturing:
        for t, lane in crossing:                           # made on a lane, consumed (and eventually freed) on `main`
            if lane is not main:
                t.record_stream(main)
    return [outs if coef is None else FN.weighted_sum(outs, coef) for outs, coef in pending]


def run_branches(ops, x):
    main = torch.cuda.current_stream()
    lanes = branch_lanes(main)
    used = []
    plan = []
    for k, op in enumerate(ops):                  # branch 0 stays on the current stream, the others round-robin the lanes
        lane = None if k == 0 else lanes[(k - 1) % len(lanes)]
        if lane is not None and lane not in used:
            lane.wait_stream(main)                # fork: every lane starts after x is ready
            used.append(lane)
        plan.append((op, lane))
    outs = []
    for op, lane in plan:
        if lane is None:
            outs.append(op(x))
        else:
            with torch.cuda.stream(lane):
                outs.append(op(x))
    for lane in used:
        main.wait_stream(lane)                    # join before the weighted sum
    return outs
