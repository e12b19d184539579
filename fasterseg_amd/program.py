"""Relocatable launch programs for the supernet's MixedOp (reference search/model_search.py:46-99), replayed by
fs_exec_program (csrc/program.hip) from one FFI call.

Why: an eager supernet pass is ~12 k launches of a few microseconds.  Issued module by module from Python (one autograd
node, four allocations and one FFI crossing per conv->BN->ReLU) the host needs ~35 us per module; the kernels ~10.  For
given widths a MixedOp is a FIXED sequence - five primitives (skip / conv / conv_downup / conv_2x / conv_2x_downup,
operations.py:131-534) and the alpha-weighted sum - so it is lowered once per (MixedOp, widths, input geometry, which
gradients are wanted) into two command lists, forward and backward, whose pointers are (slot, offset) pairs:

    slot 0  absolute   parameters, BN buffers, slices of the flat gradient buffer (never move)
    slot 1  X          the input feature map             slot 5  DY    incoming gradient (dense NHWC)
    slot 2  COEF       the 5 mixing coefficients         slot 6  TMPB  backward scratch
    slot 3  OUT        the mixed output                  slot 7  GX    gradient w.r.t. X
    slot 4  SAVE       activations kept for backward     slot 8  TMPF  forward scratch
                                                         slot 9  WS    the stream's split-K conv workspace
    slot 10 ZF         zero-initialised accumulators of the forward list (BN statistics)
    slot 11 ZB         ... of the backward list (BN reductions, coefficient gradients)

Per call the autograd Function (functional._MixedOpProgram) allocates three arenas and the output and crosses the FFI once
per direction.  The zero-initialised accumulators come out of the step's zero arena (kernels.zero_pool: ONE fill per step, or one
captured fill per graph replay) - round 4 cleared a zero region per program and direction, ~600 fill launches per C3 step and ~1200
per C5 iteration.  The per-module path in functional.py computes exactly the same thing and remains the reference
implementation of these programs (tests compare the two); it is also what runs under hipGraph capture and whenever a
precondition below does not hold.
"""
import ctypes
import struct

import torch

from . import kernels as K
from ._lib import ConvDesc, ResizeDesc

ABS, X, COEF, OUT, SAVE, DY, TMPB, GX, TMPF, WS, ZF, ZB = range(12)
N_SLOTS = 12
_ALIGN = 256

(OP_MEMSET, OP_PACK_WEIGHT, OP_CONV_FWD, OP_UNIT_FWD, OP_UNIT_BWD, OP_WGRAD_STRIDED, OP_CHANNEL_STATS, OP_BN_FINALIZE,
 OP_AFFINE_ACT, OP_BN_BWD_REDUCE, OP_BN_BWD_APPLY, OP_BILINEAR_FWD, OP_BILINEAR_BWD, OP_WSUM, OP_WSUM_BWD, OP_WSUM_DOTS,
 OP_AXPY, OP_CONV3X3_S1, OP_STEM, OP_COPY_CHANNELS, OP_EVENT_RECORD, OP_EVENT_WAIT, OP_ZOOM_CELL, OP_BILINEAR_ARGMAX,
 OP_BN_UNIT_FWD, OP_BN_UNIT_BWD) = range(26)      # enum in include/fasterseg_hip.h


class Ref:
    """A device address = slots[slot] + off."""
    __slots__ = ("slot", "off")

    def __init__(self, slot, off):
        self.slot, self.off = slot, off

    def __add__(self, nbytes):
        return Ref(self.slot, self.off + nbytes)


NULL = Ref(ABS, 0)


def absolute(t):
    return NULL if t is None else Ref(ABS, t.data_ptr())


class Buf:
    """Symbolic NHWC feature map."""
    __slots__ = ("ref", "N", "C", "H", "W", "cs", "esize")

    def __init__(self, ref, N, C, H, W, cs, esize):
        self.ref, self.N, self.C, self.H, self.W, self.cs, self.esize = ref, N, C, H, W, cs, esize

    @property
    def pixels(self):
        return self.N * self.H * self.W

    def channels(self, start, count):
        return Buf(self.ref + start * self.esize, self.N, count, self.H, self.W, self.cs, self.esize)


class _Desc:
    def __init__(self, struct):
        self.raw = bytes(struct)


class _List:
    """One direction (forward or backward) of a program."""

    def __init__(self, zero_slot=ZF):
        self.words = []
        self.blob = bytearray()
        self.sizes = {}             # arena slot -> bytes
        self.zero_slot = zero_slot  # where `zero=True` allocations live: an arena the caller hands over already cleared

    def alloc(self, slot, nbytes, zero=False):
        if zero:
            slot = self.zero_slot
        nbytes = (nbytes + _ALIGN - 1) // _ALIGN * _ALIGN
        off = self.sizes.get(slot, 0)
        self.sizes[slot] = off + nbytes
        return Ref(slot, off)

    def emit(self, op, *args, lane=0, join=False):
        """join=True: this command and the NEXT one (same op) are independent; the executor issues the run as one grouped launch."""
        self.words.append((op | (lane << 16) | (int(bool(join)) << 40), args))

    def finish(self):
        out = []
        for op, args in self.words:
            out += [op, len(args)]
            for a in args:
                if isinstance(a, Ref):
                    out += [2, a.slot, a.off]
                elif isinstance(a, _Desc):
                    off = len(self.blob)
                    self.blob += a.raw
                    self.blob += b"\0" * (-len(self.blob) % 8)
                    out += [3, 0, off]
                elif isinstance(a, float):
                    out += [1, 0, struct.unpack("<q", struct.pack("<d", a))[0]]
                elif isinstance(a, (list, tuple)):
                    if any(r is None or isinstance(r, Ref) for r in a):
                        out += [4, len(a), 0]
                        for r in a:
                            if r is None:
                                out += [0, -1]
                            else:
                                out += [r.slot, r.off]
                    else:
                        out += [5, len(a), 0] + [int(v) for v in a]
                else:
                    out += [0, 0, int(a)]
        words = (ctypes.c_longlong * len(out))(*out)
        blob = (ctypes.c_ubyte * max(1, len(self.blob))).from_buffer_copy(bytes(self.blob) or b"\0")
        return words, len(out), blob, dict(self.sizes)


class MixedOpProgram:
    """Forward + backward command lists of one MixedOp configuration."""

    def __init__(self, fwd, bwd, out_shape, touched, need_x, need_coef, gcoef_off, guard):
        self.f_words, self.f_n, self.f_blob, f_sizes = fwd.finish()
        self.b_words, self.b_n, self.b_blob, b_sizes = bwd.finish()
        self.save_bytes = max(_ALIGN, f_sizes.get(SAVE, 0))
        self.tmpf_bytes = max(_ALIGN, f_sizes.get(TMPF, 0))
        self.tmpb_bytes = max(_ALIGN, b_sizes.get(TMPB, 0))
        self.zf_bytes = f_sizes.get(ZF, 0)    # zero-initialised accumulators: slices of the step's zero arena (functional.zero_arena)
        self.zb_bytes = b_sizes.get(ZB, 0)
        self.out_shape = out_shape            # (N, C, H, W)
        self.touched = touched                # parameters whose flat-gradient slices the backward writes
        self.need_x, self.need_coef, self.gcoef_off = need_x, need_coef, gcoef_off
        self.guard = guard[:2] + guard[-2:]   # (tensor, data_ptr) samples of the storage the program hard-codes
        self.n_launches = (len(fwd.words), len(bwd.words))
        # command structure (op + argument count per command, both directions): programs with equal signatures can be replayed in
        # lockstep by fs_exec_program_group, their convolutions going out as grouped launches
        self.signature = (tuple((op & 0xffff, len(args)) for op, args in fwd.words), tuple(sorted(fwd.sizes)),
                          tuple((op & 0xffff, len(args)) for op, args in bwd.words), tuple(sorted(bwd.sizes)))

    def valid(self):
        return all(t.data_ptr() == p and (v is None or t._version == v) for t, p, v in self.guard)

    def run(self, words, n, blob, slots):
        arr = (ctypes.c_void_p * N_SLOTS)(*slots)
        K.call("fs_exec_program", K._stream(), words, n, blob, arr, N_SLOTS)


MAX_GROUP = 24         # MAX_LAYER of csrc/program.hip: programs per fs_exec_program_group call (2 x 12 problems per grouped launch)


def run_group(progs, backward, slot_lists):
    """fs_exec_program_group: the forward (or backward) lists of `progs` (any structures: the executor groups the commands of one kind
    that are pending at the same time) on the current stream."""
    k = len(progs)
    assert 1 <= k <= MAX_GROUP and len(slot_lists) == k
    lists = [(p.b_words, p.b_n, p.b_blob) if backward else (p.f_words, p.f_n, p.f_blob) for p in progs]
    words = (ctypes.c_void_p * k)(*[ctypes.addressof(w) for w, _, _ in lists])
    counts = (ctypes.c_longlong * k)(*[n for _, n, _ in lists])
    blobs = (ctypes.c_void_p * k)(*[ctypes.addressof(b) for _, _, b in lists])
    flat = []
    for sl in slot_lists:
        assert len(sl) == N_SLOTS
        flat.extend(sl)
    slots = (ctypes.c_void_p * (k * N_SLOTS))(*flat)
    K.call("fs_exec_program_group", K._stream(), k, words, counts, blobs, slots, N_SLOTS)


# ---------------------------------------------------------------------------------------------------
# lowering
# ---------------------------------------------------------------------------------------------------
class _Lowering:
    def __init__(self, dtype, want_w, sink, groups=1):
        self.f, self.b = _List(ZF), _List(ZB)
        self.groups = groups                  # BatchNorm groups: the batch is this many independently normalised inputs
        self.dtype = dtype
        self.dt = K.dtype_code(dtype)
        self.esize = 4 if dtype == torch.float32 else 2
        self.want_w = want_w                  # network weights receive gradients (through the flat-buffer sink)
        self.sink = sink
        self.touched = []
        self.guard = []
        self.back = []                        # closures emitting the backward, run in reverse

    # -- buffers -------------------------------------------------------------------------------------
    def new(self, lst, slot, N, C, H, W):
        return Buf(lst.alloc(slot, N * H * W * C * self.esize), N, C, H, W, C, self.esize)

    def grad_slot(self, p, n=None):
        """Address of p.grad inside the flat buffer (fused accumulation), asserting the sink owns it."""
        g = p.grad
        assert self.sink is not None and g is not None and self.sink.accepts(p), "parameter gradient is not a flat-buffer view"
        self.guard.append((g, g.data_ptr(), None))
        self.touched.append(p)
        return g

    def param(self, t):
        self.guard.append((t, t.data_ptr(), None))
        return absolute(t)

    def filter(self, lst, slot, w, cout, cin, flip):
        """Address + (row, tap) strides of the packed [:cout,:cin] filter block: the resident pack kept current by the
        optimizer when there is one (no launch), else a fs_pack_weight into scratch."""
        from . import functional as FN
        O, I, R, S = w.shape
        rp = FN.resident_pack(w, self.dtype)
        if rp is not None:
            self.guard.append((w, w.data_ptr(), w._version))       # resident packs go stale if the parameter is edited in place
            lead = O if flip else I
            return absolute(rp[1] if flip else rp[0]), R * S * lead, lead
        assert w.stride(3) == 1 and w.stride(2) == S
        self.guard.append((w, w.data_ptr(), None))
        wp = lst.alloc(slot, cout * R * S * cin * self.esize)
        lst.emit(OP_PACK_WEIGHT, absolute(w), w.stride(0), w.stride(1), cout, cin, R, S, self.dt, int(flip), wp)
        return wp, 0, 0

    # -- one conv -> BN -> [ReLU] module -----------------------------------------------------------------
    def unit(self, x, conv, bn, relu):
        f = self.f
        cout, cin = conv.active_channels()
        b = bn.active()
        assert b.training and b.track_running_stats and x.C == cin
        w = conv.weight
        R, S = w.shape[2], w.shape[3]
        stride, pad = conv.stride[0], conv.padding[0]
        Ho, Wo = (x.H + 2 * pad - R) // stride + 1, (x.W + 2 * pad - S) // stride + 1
        wp, w_os, w_ts = self.filter(f, TMPF, w, cout, cin, False)
        G = self.groups
        desc = ConvDesc(x.N, x.H, x.W, cin, cout, R, S, stride, pad, Ho, Wo, x.cs, cout, self.dt, K.FS_CONV_RELU if relu else 0,
                        w_os, w_ts, 0, 0, 0, G)
        stats = f.alloc(TMPF, G * 2 * cout * 4, zero=True)
        saved = f.alloc(SAVE, G * 4 * cout * 4)
        z = self.new(f, SAVE, x.N, cout, Ho, Wo)
        y = self.new(f, SAVE, x.N, cout, Ho, Wo)
        momentum = 0.1 if b.momentum is None else float(b.momentum)
        f.emit(OP_UNIT_FWD, _Desc(desc), x.ref, wp, self.param(b.weight), self.param(b.bias), self.param(b.running_mean),
               self.param(b.running_var), self.param(b.num_batches_tracked), float(b.eps), momentum, stats, saved, z.ref, y.ref,
               Ref(WS, 0), K.WORKSPACE_BYTES)

        def backward(dy, need_x, dx_into=None):
            bl = self.b
            red = bl.alloc(TMPB, (G + 1 if G > 1 else 1) * 2 * cout * 4, zero=True)
            dz = self.new(bl, TMPB, x.N, cout, Ho, Wo)
            wf, wf_os, wf_ts, dx = None, 0, 0, None
            if need_x:
                wf, wf_os, wf_ts = self.filter(bl, TMPB, w, cout, cin, True)
                dx = dx_into if dx_into is not None else self.new(bl, TMPB, x.N, cin, x.H, x.W)
            if self.want_w:
                g = self.grad_slot(w)
                assert g.stride(2) == S * g.stride(3)
                gg, gb = self.grad_slot(b.weight), self.grad_slot(b.bias)
                dw = (absolute(g), g.stride(0), g.stride(1), g.stride(3))
                acc = (absolute(gg), absolute(gb))
            else:
                dw, acc = (NULL, 0, 0, 0), (NULL, NULL)
            bl.emit(OP_UNIT_BWD, _Desc(desc), x.ref, wf or NULL, z.ref, y.ref if relu else NULL, dy.ref, dy.cs, saved,
                    absolute(b.weight), red, acc[0], acc[1], dz.ref, dw[0], dw[1], dw[2], dw[3], dx.ref if need_x else NULL,
                    dx.cs if need_x else cin, wf_os, wf_ts, Ref(WS, 0), K.WORKSPACE_BYTES)
            return dx
        return y, backward

    # -- two conv -> BN -> [ReLU] modules on the same input as ONE unit over their concatenated output channels --------------
    def fused_unit(self, x, conv_a, bn_a, conv_b, bn_b, relu_a, relu_b):
        """conv_a / conv_b: USConv2d of identical geometry reading x; bn_*: their USBatchNorm2d.  Output: one map with channels
        [A | B].  Returns None when the pair's storage is not laid out as two segments of one array (fusion.colocate / flat_order
        were not applied): the caller then lowers the primitives one by one."""
        from . import functional as FN
        from .fusion import adjacent
        f = self.f
        cout, cin = conv_a.active_channels()
        if conv_b.active_channels() != (cout, cin) or x.C != cin or not (relu_b or not relu_a):
            return None
        ba, bb = bn_a.active(), bn_b.active()
        wa, wb = conv_a.weight, conv_b.weight
        O, I, R, S = wa.shape
        taps = R * S
        if tuple(wb.shape) != (O, I, R, S) or conv_a.stride != conv_b.stride or conv_a.padding != conv_b.padding:
            return None
        for t in (ba, bb):
            if not (t.training and t.track_running_stats):
                return None
        momentum = 0.1 if ba.momentum is None else float(ba.momentum)
        if (0.1 if bb.momentum is None else float(bb.momentum)) != momentum or float(ba.eps) != float(bb.eps) or ba.num_features != cout:
            return None
        if not (adjacent(ba.weight, bb.weight) and adjacent(ba.bias, bb.bias) and adjacent(ba.running_mean, bb.running_mean)
                and adjacent(ba.running_var, bb.running_var) and adjacent(ba.num_batches_tracked, bb.num_batches_tracked)):
            return None
        g_jump = 0
        if self.want_w:
            ga, gb = wa.grad, wb.grad
            if ga is None or gb is None or not adjacent(ga, gb, ga.numel() * 4) or ga.stride() != gb.stride():
                return None
            for pa, pb in ((ba.weight, bb.weight), (ba.bias, bb.bias)):
                if pa.grad is None or pb.grad is None or not adjacent(pa.grad, pb.grad):
                    return None
            g_jump = O - cout
        rpa, rpb = FN.resident_pack(wa, self.dtype), FN.resident_pack(wb, self.dtype)
        resident = rpa is not None and rpb is not None and adjacent(rpa[0], rpb[0]) and adjacent(rpa[1], rpb[1])
        stride, pad = conv_a.stride[0], conv_a.padding[0]
        Ho, Wo = (x.H + 2 * pad - R) // stride + 1, (x.W + 2 * pad - S) // stride + 1
        C2 = 2 * cout
        G = self.groups
        if resident:
            for w in (wa, wb):
                self.guard.append((w, w.data_ptr(), w._version))
            wp, w_os, w_ts, n_jump = absolute(rpa[0]), taps * I, I, O - cout
        else:       # dense scratch pack [A rows | B rows]
            for w in (wa, wb):
                assert w.stride(3) == 1 and w.stride(2) == S
                self.guard.append((w, w.data_ptr(), None))
            wp = f.alloc(TMPF, C2 * taps * cin * self.esize)
            for k, w in enumerate((wa, wb)):
                f.emit(OP_PACK_WEIGHT, absolute(w), w.stride(0), w.stride(1), cout, cin, R, S, self.dt, 0, wp + k * cout * taps * cin * self.esize)
            w_os, w_ts, n_jump = 0, 0, 0
        flags = 0
        if relu_b:
            flags = K.FS_CONV_RELU | (0 if relu_a else K.FS_CONV_RELU_TAIL)
        k_jump = (I * taps * O - cout) if resident else (cin * taps * cout - cout)
        desc = ConvDesc(x.N, x.H, x.W, cin, C2, R, S, stride, pad, Ho, Wo, x.cs, C2, self.dt, flags, w_os, w_ts, 0, 0, 0, G,
                        cout, n_jump, 0, k_jump, g_jump)
        stats = f.alloc(TMPF, G * 2 * C2 * 4, zero=True)
        saved = f.alloc(SAVE, G * 4 * C2 * 4)
        z = self.new(f, SAVE, x.N, C2, Ho, Wo)
        y = self.new(f, SAVE, x.N, C2, Ho, Wo)
        f.emit(OP_UNIT_FWD, _Desc(desc), x.ref, wp, self.param(ba.weight), self.param(ba.bias), self.param(ba.running_mean),
               self.param(ba.running_var), self.param(ba.num_batches_tracked), float(ba.eps), momentum, stats, saved, z.ref, y.ref,
               Ref(WS, 0), K.WORKSPACE_BYTES)
        for t in (bb.weight, bb.bias, bb.running_mean, bb.running_var):
            self.guard.append((t, t.data_ptr(), None))

        def backward(dcat, need_x):
            """dcat: gradient w.r.t. the [A | B] output (2 cout channels, any channel stride)."""
            bl = self.b
            red = bl.alloc(TMPB, (G + 1 if G > 1 else 1) * 2 * C2 * 4, zero=True)
            dz = self.new(bl, TMPB, x.N, C2, Ho, Wo)
            wf, wf_os, wf_ts, dx = None, 0, 0, None
            if need_x:
                if resident:
                    wf, wf_os, wf_ts = absolute(rpa[1]), taps * O, O
                else:       # dense rotated packs [cin][R][S][cout], A then B
                    wf = bl.alloc(TMPB, 2 * cin * taps * cout * self.esize)
                    for k, w in enumerate((wa, wb)):
                        bl.emit(OP_PACK_WEIGHT, absolute(w), w.stride(0), w.stride(1), cout, cin, R, S, self.dt, 1,
                                wf + k * cin * taps * cout * self.esize)
                    wf_os, wf_ts = taps * cout, cout
                dx = self.new(bl, TMPB, x.N, cin, x.H, x.W)
            if self.want_w:
                g = self.grad_slot(wa)
                self.grad_slot(wb)
                assert g.stride(2) == S * g.stride(3)
                gg, gb_ = self.grad_slot(ba.weight), self.grad_slot(ba.bias)
                self.grad_slot(bb.weight)
                self.grad_slot(bb.bias)
                dw = (absolute(g), g.stride(0), g.stride(1), g.stride(3))
                acc = (absolute(gg), absolute(gb_))
            else:
                dw, acc = (NULL, 0, 0, 0), (NULL, NULL)
            bl.emit(OP_UNIT_BWD, _Desc(desc), x.ref, wf or NULL, z.ref, y.ref if relu_b else NULL, dcat.ref, dcat.cs, saved,
                    absolute(ba.weight), red, acc[0], acc[1], dz.ref, dw[0], dw[1], dw[2], dw[3], dx.ref if need_x else NULL,
                    cin, wf_os, wf_ts, Ref(WS, 0), K.WORKSPACE_BYTES)
            return dx
        return y, backward, cout

    # -- bilinear resize (align_corners=True) --------------------------------------------------------
    def resize(self, x, Ho, Wo, relu, join=False):
        """join: the NEXT command is another, independent resample (the executor issues the run as one grouped launch)."""
        f = self.f
        y = self.new(f, SAVE, x.N, x.C, Ho, Wo)
        f.emit(OP_BILINEAR_FWD, _Desc(ResizeDesc(x.N, x.H, x.W, Ho, Wo, x.C, x.cs, y.cs, self.dt, int(relu), 0)), x.ref, y.ref, join=join)

        def backward(dy, need_x, dx_into=None, join=False):
            if not need_x:
                return None
            bl = self.b
            if relu and dy.cs != y.cs:
                raise NotImplementedError("strided gradient into a ReLU-fused resize")
            dx = dx_into if dx_into is not None else self.new(bl, TMPB, x.N, x.C, x.H, x.W)
            bl.emit(OP_BILINEAR_BWD, _Desc(ResizeDesc(x.N, x.H, x.W, Ho, Wo, x.C, dx.cs, dy.cs, self.dt, int(relu), 0)), dy.ref,
                    y.ref if relu else NULL, dx.ref, join=join)
            return dx
        return y, backward

    # -- 'skip' with stride 2: FactorizedReduce (operations.py:521-526) -------------------------------
    def factorized_reduce(self, x, op):
        f = self.f
        half, cin = op.conv1.active_channels()
        assert op.conv2.active_channels() == (half, cin) and x.C == cin and x.H % 2 == 0 and x.W % 2 == 0
        b = op.bn.active()
        assert b.training and b.track_running_stats
        C2 = 2 * half
        Ho, Wo = x.H // 2, x.W // 2
        z = self.new(f, SAVE, x.N, C2, Ho, Wo)
        y = self.new(f, SAVE, x.N, C2, Ho, Wo)
        descs, packs = [], []
        for conv, pad in ((op.conv1, 0), (op.conv2, -1)):           # (the filters first: a non-resident one emits its pack command here)
            wp, w_os, w_ts = self.filter(f, TMPF, conv.weight, half, cin, False)
            packs.append(wp)
            descs.append(ConvDesc(x.N, x.H, x.W, cin, half, 1, 1, 2, pad, Ho, Wo, x.cs, C2, self.dt, 0, w_os, w_ts))
        for k in range(2):      # both read x, each writes its own channel half of z: ONE grouped launch (JOIN, csrc/program.hip)
            f.emit(OP_CONV_FWD, _Desc(descs[k]), x.ref, packs[k], NULL, NULL, z.ref + k * half * self.esize, NULL, Ref(WS, 0), K.WORKSPACE_BYTES,
                   join=(k == 0 and _JOIN_FR))
        G = self.groups
        stats = f.alloc(TMPF, G * 2 * C2 * 4, zero=True)
        saved = f.alloc(SAVE, G * 4 * C2 * 4)
        momentum = 0.1 if b.momentum is None else float(b.momentum)
        f.emit(OP_BN_UNIT_FWD, z.pixels, C2, G, z.ref, z.cs, self.param(b.weight), self.param(b.bias), float(b.eps), momentum,
               self.param(b.running_mean), self.param(b.running_var), self.param(b.num_batches_tracked), stats, saved, y.ref, y.cs,
               self.dt, 1, Ref(WS, 0), K.WORKSPACE_BYTES)

        def backward(dy, need_x):
            bl = self.b
            red = bl.alloc(TMPB, (G + 1 if G > 1 else 1) * 2 * C2 * 4, zero=True)
            dz = self.new(bl, TMPB, x.N, C2, Ho, Wo)
            acc = (absolute(self.grad_slot(b.weight)), absolute(self.grad_slot(b.bias))) if self.want_w else (NULL, NULL)
            bl.emit(OP_BN_UNIT_BWD, z.pixels, C2, G, z.ref, z.cs, dy.ref, dy.cs, y.ref, y.cs, saved, absolute(b.weight), red,
                    self.dt, 1, dz.ref, dz.cs, acc[0], acc[1], Ref(WS, 0), K.WORKSPACE_BYTES)
            convs = (op.conv1, op.conv2)
            dzs = [dz.ref + k * half * self.esize for k in range(2)]
            if self.want_w:         # the two weight gradients: independent, one grouped launch
                for k, conv in enumerate(convs):
                    g = self.grad_slot(conv.weight)
                    bl.emit(OP_WGRAD_STRIDED, _Desc(descs[k]), x.ref, dzs[k], absolute(g), g.stride(0), g.stride(1), g.stride(3), Ref(WS, 0),
                            K.WORKSPACE_BYTES, join=(k == 0 and _JOIN_FR))
            if not need_x:
                return None
            flips = [self.filter(bl, TMPB, conv.weight, half, cin, True) for conv in convs]
            gks = [self.new(bl, TMPB, x.N, cin, x.H, x.W) for _ in convs]
            for k in range(2):      # the two data gradients into their own maps: one grouped launch, then the sum
                wf, wf_os, wf_ts = flips[k]
                g = ConvDesc(x.N, Ho, Wo, half, cin, 1, 1, 1, 0 - descs[k].pad, x.H, x.W, C2, cin, self.dt, K.FS_CONV_TRANSPOSED, wf_os, wf_ts)
                bl.emit(OP_CONV_FWD, _Desc(g), dzs[k], wf, NULL, NULL, gks[k].ref, NULL, Ref(WS, 0), K.WORKSPACE_BYTES, join=(k == 0 and _JOIN_FR))
            dx, gk = gks
            # the second branch touches only odd (h, w): disjoint from the first one's even taps
            bl.emit(OP_AXPY, dx.pixels, cin, gk.ref, gk.cs, absolute(_ones(b.weight.device)), dx.ref, dx.cs, self.dt, 1)
            return dx
        return y, backward

    # -- the five primitives ---------------------------------------------------------------------------
    def primitive(self, x, op):
        from .operations import FactorizedReduce, _Residual
        steps = []
        if isinstance(op, FactorizedReduce):
            assert op.slimmable
            if op.stride == 2:
                y, bw = self.factorized_reduce(x, op)
                steps.append(bw)
            else:
                y, bw = self.unit(x, op.conv1, op.bn, True)
                steps.append(bw)
        elif isinstance(op, _Residual):
            assert op.slimmable
            upsample = op.ZOOM and op.stride == 1
            y = x
            if op.ZOOM:
                y, bw = self.resize(y, x.H // 2, x.W // 2, False)
                steps.append(bw)
            y, bw = self.unit(y, op.conv1, op.bn1, op.NUM_CONVS == 2 or not upsample)
            steps.append(bw)
            if op.NUM_CONVS == 2:
                y, bw = self.unit(y, op.conv2, op.bn2, not upsample)
                steps.append(bw)
            if upsample:
                y, bw = self.resize(y, x.H, x.W, True)
                steps.append(bw)
        else:
            raise NotImplementedError(type(op).__name__)

        def backward(dy, need_x):
            for k in range(len(steps) - 1, -1, -1):
                dy = steps[k](dy, need_x or k > 0)
            return dy
        return y, backward


_one_vectors = {}


def _ones(device):
    key = str(device)
    if key not in _one_vectors:
        _one_vectors[key] = torch.ones(8, dtype=torch.float32, device=device)
    return _one_vectors[key]


# FS_FUSE_MIXEDOP=0 lowers every primitive on its own (the round-2 programs)
import os
_FUSE = bool(int(os.environ.get("FS_FUSE_MIXEDOP", "1")))
# FactorizedReduce (operations.py:521-526): its two 1x1 stride-2 convolutions, their weight gradients and their data gradients as one
# grouped launch each (JOIN bit).  FS_JOIN_FR=0: one launch per convolution (the round-4 programs).
_JOIN_FR = bool(int(os.environ.get("FS_JOIN_FR", "1")))
# the two up-samples of a stride-1 MixedOp's zoomed primitives, forward and backward, as one grouped launch each (round 6; 0: two launches)
_JOIN_UP = bool(int(os.environ.get("FS_JOIN_UP", "1")))


def _lower_fused(lo, x, mixed, need_x):
    """The five primitives with the first convs of ('conv', 'conv_2x') and of ('conv_downup', 'conv_2x_downup') as one unit each and
    one shared down-sample (fusion.py).  Returns (outs, grads, run_backward) or None when the pairs' storage is not adjacent.
    grads[k]: where the weighted sum's backward must write d out_k (a channel slice of a pair's gradient buffer where that saves a
    copy); run_backward() emits the backward commands and returns the tensors whose sum is d x."""
    from .fusion import pair_modules
    from .operations import FactorizedReduce
    ops = mixed._ops
    pairs = pair_modules(mixed)
    if len(pairs) != 2 or not isinstance(ops[0], FactorizedReduce):
        return None
    (conv, conv2x), (du, du2x) = pairs
    if conv.ZOOM or not du.ZOOM or conv.NUM_CONVS != 1 or conv2x.NUM_CONVS != 2 or du.NUM_CONVS != 1 or du2x.NUM_CONVS != 2:
        return None
    mark_f, mark_b = (len(lo.f.words), dict(lo.f.sizes)), None
    stride = conv.stride
    upsample = stride == 1
    y0, bw0 = lo.primitive(x, ops[0])
    fa = lo.fused_unit(x, conv.conv1, conv.bn1, conv2x.conv1, conv2x.bn1, True, True)
    if fa is None:
        lo.f.words = lo.f.words[:mark_f[0]]
        lo.f.sizes = mark_f[1]
        return None
    ya, bwa, c = fa
    y3, bw3 = lo.unit(ya.channels(c, c), conv2x.conv2, conv2x.bn2, True)
    xd, bw_down = lo.resize(x, x.H // 2, x.W // 2, False)
    fb = lo.fused_unit(xd, du.conv1, du.bn1, du2x.conv1, du2x.bn1, not upsample, True)
    if fb is None:
        lo.f.words = lo.f.words[:mark_f[0]]
        lo.f.sizes = mark_f[1]
        return None
    yb, bwb, cb = fb
    assert cb == c
    y4, bw4 = lo.unit(yb.channels(c, c), du2x.conv2, du2x.bn2, not upsample)
    y2 = yb.channels(0, c)
    bw_up2 = bw_up4 = None
    if upsample:          # the two zoomed primitives' up-samples (operations.py:275,444): independent, one grouped launch (JOIN)
        y2, bw_up2 = lo.resize(y2, x.H, x.W, True, join=_JOIN_UP)
        y4, bw_up4 = lo.resize(y4, x.H, x.W, True)
    outs = [y0, ya.channels(0, c), y2, y3, y4]
    b = lo.b
    o = outs[0]
    dcat_a = lo.new(b, TMPB, ya.N, 2 * c, ya.H, ya.W)                 # d [conv out | conv_2x.conv1 out]
    dcat_b = lo.new(b, TMPB, yb.N, 2 * c, yb.H, yb.W)                 # d [conv_downup.conv1 out | conv_2x_downup.conv1 out]
    dense = lambda: lo.new(b, TMPB, o.N, o.C, o.H, o.W)
    grads = [dense(), dcat_a.channels(0, c), dense() if upsample else dcat_b.channels(0, c), dense(), dense()]

    def run_backward():
        if upsample:
            bw_up2(grads[2], True, dx_into=dcat_b.channels(0, c), join=_JOIN_UP)
            d4 = bw_up4(grads[4], True)
        else:
            d4 = grads[4]
        bw4(d4, True, dx_into=dcat_b.channels(c, c))
        dxd = bwb(dcat_b, need_x)
        dx_down = bw_down(dxd, need_x) if need_x else None
        bw3(grads[3], True, dx_into=dcat_a.channels(c, c))
        dx_a = bwa(dcat_a, need_x)
        dx_0 = bw0(grads[0], need_x)
        return [dx_0, dx_a, dx_down]
    return outs, grads, run_backward


def lower_mixed_op(mixed, x_shape, x_cs, dtype, device, need_x, need_coef, want_w, sink, groups=1):
    """Programs for `mixed` (ratios already set through set_prun_ratio) on an (N, C, H, W) NHWC input with channel stride x_cs;
    `groups` > 1: the batch is that many inputs the BatchNorms normalise independently (functional.bn_groups)."""
    N, C, H, W = x_shape
    assert N % groups == 0
    lo = _Lowering(dtype, want_w, sink, groups)
    x = Buf(Ref(X, 0), N, C, H, W, x_cs, lo.esize)
    fused = _lower_fused(lo, x, mixed, need_x) if (_FUSE and len(mixed._ops) == 5) else None
    outs, backs = [], []
    if fused is not None:
        outs = fused[0]
    else:
        for op in mixed._ops:
            y, bw = lo.primitive(x, op)
            outs.append(y)
            backs.append(bw)
    n = len(outs)
    y0 = outs[0]
    assert all((o.N, o.C, o.H, o.W) == (y0.N, y0.C, y0.H, y0.W) for o in outs)
    lo.f.emit(OP_WSUM, y0.pixels, y0.C, n, [o.ref for o in outs], [o.cs for o in outs], Ref(COEF, 0), Ref(OUT, 0), y0.C, lo.dt)
    # backward: dy -> the five branch gradients -> each primitive -> sum into GX
    b = lo.b
    dy = Buf(Ref(DY, 0), y0.N, y0.C, y0.H, y0.W, y0.C, lo.esize)
    gcoef = None
    if need_coef:
        gcoef = b.alloc(TMPB, 4 * 8, zero=True)
        b.emit(OP_WSUM_DOTS, dy.pixels, dy.C, n, dy.ref, dy.cs, [o.ref for o in outs], [o.cs for o in outs], lo.dt, gcoef)
    gys = fused[1] if fused is not None else [lo.new(b, TMPB, y0.N, y0.C, y0.H, y0.W) for _ in range(n)]
    b.emit(OP_WSUM_BWD, dy.pixels, dy.C, n, dy.ref, dy.cs, Ref(COEF, 0), [g.ref for g in gys], [g.cs for g in gys], lo.dt)
    dxs = fused[2]() if fused is not None else [backs[k](gys[k], need_x) for k in range(n)]
    if need_x:
        b.emit(OP_WSUM, x.pixels, C, len(dxs), [d.ref for d in dxs], [d.cs for d in dxs], absolute(_ones(device)), Ref(GX, 0), C, lo.dt)
    gcoef_off = None
    if gcoef is not None:
        gcoef_off = gcoef.off             # byte offset inside the backward list's zero arena (slot ZB)
    touched, seen = [], set()
    for p in lo.touched:
        if id(p) not in seen:
            seen.add(id(p))
            touched.append(p)
    prog = MixedOpProgram(lo.f, lo.b, (y0.N, y0.C, y0.H, y0.W), touched, need_x, need_coef, gcoef_off, lo.guard)
    prog.fused = fused is not None
    return prog


# ---------------------------------------------------------------------------------------------------
# one-command programs for the grouped beta merges of a supernet layer (functional.pair_merge_group)
# ---------------------------------------------------------------------------------------------------
class _MiniProgram:
    def __init__(self, fwd, bwd, need_coef):
        self.f_words, self.f_n, self.f_blob, _ = fwd.finish()
        self.b_words, self.b_n, self.b_blob, b_sizes = bwd.finish()
        self.zb_bytes = b_sizes.get(ZB, 0)
        self.need_coef = need_coef


_pair_merge_programs = {}


def pair_merge_program(x_shape, dtype, need_coef):
    """out = coef[0] * x[:n] + coef[1] * x[n:] for a dense NHWC x of shape (2n, C, H, W) (slots X, COEF, OUT) and its backward (DY, GX; the
    two coefficient gradients land in the first two floats of the ZB slice).  Cached per (shape, dtype, need_coef)."""
    key = (tuple(x_shape), dtype, bool(need_coef))
    prog = _pair_merge_programs.get(key)
    if prog is None:
        n2, C, H, W = x_shape
        assert n2 % 2 == 0
        esize = 4 if dtype == torch.float32 else 2
        dt = K.dtype_code(dtype)
        pixels = (n2 // 2) * H * W
        half = pixels * C * esize
        f, b = _List(ZF), _List(ZB)
        f.emit(OP_WSUM, pixels, C, 2, [Ref(X, 0), Ref(X, half)], [C, C], Ref(COEF, 0), Ref(OUT, 0), C, dt)
        if need_coef:
            gc = b.alloc(TMPB, 4 * 8, zero=True)
            b.emit(OP_WSUM_DOTS, pixels, C, 2, Ref(DY, 0), C, [Ref(X, 0), Ref(X, half)], [C, C], dt, gc)
        b.emit(OP_WSUM_BWD, pixels, C, 2, Ref(DY, 0), C, Ref(COEF, 0), [Ref(GX, 0), Ref(GX, half)], [C, C], dt)
        prog = _pair_merge_programs[key] = _MiniProgram(f, b, bool(need_coef))
    return prog
