"""Tensor-level wrappers over the C ABI: torch tensors in, raw pointers + sizes out.

Activation convention: a feature map is a torch tensor of logical shape (N, C, H, W) whose memory is NHWC with a
channel stride cs >= C (strides (H*W*cs, 1, W*cs, cs)); channel slices of a wider buffer are therefore valid
operands and torch.cat becomes "write into a slice".  torch is used only for memory, streams and autograd plumbing.
"""
import ctypes

import torch

from . import _lib
from ._lib import FS_BF16, FS_CONV_RELU, FS_CONV_RELU_TAIL, FS_CONV_TRANSPOSED, FS_F32, ConvDesc, ResizeDesc, ZoomDesc, call

_DT = {torch.float32: FS_F32, torch.bfloat16: FS_BF16}


def dtype_code(dt):
    try:
        return _DT[dt]
    except KeyError:
        raise TypeError("fasterseg_amd kernels support float32 and bfloat16, got %s" % dt)


def vec_of(dt):
    return 4 if dt == torch.float32 else 8


def round_up(v, m):
    return (v + m - 1) // m * m


def _stream():
    """hipStream_t of torch's current stream.  torch.cuda.current_stream() costs ~15 us per call (it re-reads an environment
    variable to resolve the device index), which was a quarter of an eager supernet step; the raw C accessors are ~0.3 us."""
    return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


class _ZeroPool:
    """Arena of pre-zeroed fp32 scratch (BN statistic / reduction accumulators, dot outputs).  A train step needs
    thousands of 100-byte zero buffers; taking them from one arena that is cleared with a single memset per step
    (reset(), called by parallel.FlatGradientSync.prepare) removes one fill launch per request.  Views are only valid
    until the next reset; outside an active step (or when the arena is full) requests fall back to torch.zeros."""

    def __init__(self, capacity=1 << 24):
        self.capacity = capacity
        self.buf = None
        self.off = 0
        self.active = False
        self.cap_buf = None          # arena owned by the hipGraph being captured (begin_capture / end_capture)
        self.cap_off = 0

    def begin_capture(self, device, capacity=1 << 22):
        """Call as the FIRST thing inside a graph capture: one captured fill re-zeroes the arena on every replay, and the
        requests of the captured pass are slices of it instead of one fill launch each (~1.5 k per supernet pass)."""
        self.cap_buf = torch.zeros(capacity, dtype=torch.float32, device=device)
        self.cap_off = 0

    def end_capture(self):
        """Returns the arena; the caller keeps it alive as long as the graph."""
        buf, self.cap_buf = self.cap_buf, None
        return buf

    def reset(self, device):
        if self.buf is None or self.buf.device != torch.device(device):
            self.buf = torch.zeros(self.capacity, dtype=torch.float32, device=device)
        elif self.off:
            self.buf[:self.off].zero_()
        self.off = 0
        self.active = True

    def stop(self):
        self.active = False

    def take(self, n, device, align=4):
        """n zeroed floats; `align` (floats, power of two): alignment of the view's first element inside the arena (the arena itself is
        allocator-aligned: 512 bytes)."""
        if self.cap_buf is not None and torch.cuda.is_current_stream_capturing():
            off = (self.cap_off + align - 1) & ~(align - 1)
            if off + n <= self.cap_buf.numel():
                self.cap_off = off + ((n + 3) & ~3)
                return self.cap_buf[off:off + n]
        if (not self.active or self.buf is None or self.buf.device != torch.device(device)
                or torch.cuda.is_current_stream_capturing()):      # a captured graph must own (and re-zero) its scratch
            return torch.zeros(n, dtype=torch.float32, device=device)
        off = (self.off + align - 1) & ~(align - 1)
        if off + n > self.capacity:
            return torch.zeros(n, dtype=torch.float32, device=device)
        self.off = off + ((n + 3) & ~3)         # keep 16-byte alignment
        return self.buf[off:off + n]


zero_pool = _ZeroPool()


def zeros_f32(n, device):
    return zero_pool.take(n, device)


def empty_nhwc(N, C, H, W, dtype, device, cs=None, zero=False):
    """(N,C,H,W) view over a fresh NHWC buffer with channel stride cs (default: C rounded up to the vector width)."""
    cs = cs or round_up(C, vec_of(dtype))
    make = torch.zeros if zero else torch.empty
    buf = make((N, H, W, cs), dtype=dtype, device=device)
    return buf.permute(0, 3, 1, 2)[:, :C]


def channel_stride(t):
    """Channel stride of an NHWC view, or None if `t` is not laid out that way."""
    if t.dim() != 4:
        return None
    N, C, H, W = t.shape
    sN, sC, sH, sW = t.stride()
    cs = sW
    if C > 1 and sC != 1:
        return None
    if cs < C:
        return None
    if (H > 1 and sH != W * cs) or (N > 1 and sN != H * W * cs):
        return None
    return cs


def is_nhwc(t, dtype=None):
    if not t.is_cuda or t.dtype not in _DT or (dtype is not None and t.dtype != dtype):
        return False
    cs = channel_stride(t)
    if cs is None:
        return False
    v = vec_of(t.dtype)
    return cs % v == 0 and t.data_ptr() % 16 == 0


def require_nhwc(t, what="tensor"):
    if not is_nhwc(t):
        raise ValueError("%s must be an NHWC-strided CUDA tensor (shape %s, strides %s, dtype %s)" %
                         (what, tuple(t.shape), t.stride(), t.dtype))
    return channel_stride(t)


def to_nhwc(x, dtype):
    """Any (N,C,H,W) CUDA tensor -> NHWC view of `dtype` (fs_nchw_to_nhwc for contiguous fp32 NCHW inputs)."""
    if is_nhwc(x, dtype):
        return x
    N, C, H, W = x.shape
    if is_nhwc(x):            # NHWC but other dtype: widen/narrow through NCHW fp32
        x = to_nchw(x)
    src = x.detach().to(torch.float32).contiguous()
    cpad = round_up(C, vec_of(dtype))
    out = empty_nhwc(N, C, H, W, dtype, x.device, cs=cpad)
    call("fs_nchw_to_nhwc", _stream(), N, C, H, W, _p(src), _p(out), cpad, cpad, dtype_code(dtype))
    return out


def to_nchw(x):
    """NHWC view -> contiguous NCHW fp32."""
    cs = require_nhwc(x, "x")
    N, C, H, W = x.shape
    out = torch.empty((N, C, H, W), dtype=torch.float32, device=x.device)
    call("fs_nhwc_to_nchw", _stream(), N, C, H, W, _p(x), cs, dtype_code(x.dtype), _p(out))
    return out


# ---------------------------------------------------------------------------------------------------
# filters
# ---------------------------------------------------------------------------------------------------
def pack_weight(w, dtype, cout=None, cin=None, flip=False, rows=None):
    """OIHW fp32 (optionally the leading [:cout,:cin] slice, slimmable_ops.py:42) -> packed [Cout][R][S][Cin] `dtype`.
    flip=True gives the data-gradient filter [Cin][R][S][Cout] rotated by 180 degrees.  `rows` zero-pads the leading
    dimension (used for the 19-class classifier)."""
    O, I, R, S = w.shape
    cout = O if cout is None else cout
    cin = I if cin is None else cin
    assert w.dtype == torch.float32 and w.stride(3) == 1 and w.stride(2) == S, "filter must be fp32 with contiguous taps"
    lead, inner = (cin, cout) if flip else (cout, cin)
    nrows = lead if rows is None else rows
    make = torch.zeros if nrows != lead else torch.empty
    out = make((nrows, R, S, inner), dtype=dtype, device=w.device)
    call("fs_pack_weight", _stream(), _p(w), w.stride(0), w.stride(1), cout, cin, R, S, dtype_code(dtype), int(flip), _p(out))
    return out


def unpack_weight_grad(dw_packed, grad, cout, cin, accumulate=False):
    """packed fp32 [cout][R][S][cin] -> grad[:cout,:cin] of an OIHW fp32 tensor."""
    R, S = grad.shape[2], grad.shape[3]
    call("fs_unpack_weight_grad", _stream(), _p(dw_packed), cout, cin, R, S, _p(grad), grad.stride(0), grad.stride(1),
         int(accumulate))


# ---------------------------------------------------------------------------------------------------
# convolution
# ---------------------------------------------------------------------------------------------------
WORKSPACE_BYTES = 16 << 20          # FS_CONV_WORKSPACE_BYTES
WS_COUNTER_BYTES = 65536            # FS_WS_COUNTER_BYTES: the workspace's tail holds zero-initialised arrival counters
_workspaces = {}


def stream_workspace(device):
    """(address, bytes) of the split-K scratch buffer of the current stream: convs that share a workspace must be ordered,
    which holds for everything issued on one stream (inside a capture too: one buffer per forked lane)."""
    raw = torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())
    ws = _workspaces.get(raw)
    if ws is None:
        ws = _workspaces[raw] = torch.empty(WORKSPACE_BYTES, dtype=torch.uint8, device=device)
        ws[-WS_COUNTER_BYTES:].zero_()         # arrival counters of the deterministic reductions: zero once, every kernel leaves them zero
    return ws.data_ptr(), WORKSPACE_BYTES


def conv_desc(x_shape, x_cs, cout, R, S, stride, pad, y_cs, dtype, flags=0, out_hw=None):
    N, Cin, H, W = x_shape
    if out_hw is None:
        Ho = (H + 2 * pad - R) // stride + 1
        Wo = (W + 2 * pad - S) // stride + 1
    else:
        Ho, Wo = out_hw
    return ConvDesc(N, H, W, Cin, cout, R, S, stride, pad, Ho, Wo, x_cs, y_cs, dtype_code(dtype), flags)


def conv2d(x, w_packed, cout, R, S, stride, pad, scale=None, shift=None, relu=False, out=None, stats=None,
           transposed=False, out_hw=None, w_strides=None, workspace=True, vres=None):
    """y = relu?(conv(x, w) * scale + shift); `out` may be a channel slice of a wider NHWC buffer.  w_strides = (row, tap)
    element strides when the filter is the leading block of a wider packed bank; workspace=False keeps the whole contraction
    in one block per tile (no cross-block split-K).  vres = (H, W, relu): the convolution reads x bilinearly resampled
    (align_corners=True, optional ReLU after the interpolation) to H x W without materialising that map."""
    x_cs = require_nhwc(x, "x")
    N, Cin, H, W = x.shape
    flags = (FS_CONV_RELU if relu else 0) | (FS_CONV_TRANSPOSED if transposed else 0)
    d = conv_desc((N, Cin, vres[0], vres[1]) if vres else x.shape, x_cs, cout, R, S, stride, pad, 0, x.dtype, flags, out_hw)
    if vres:
        d.vr_H, d.vr_W, d.vr_relu = H, W, int(bool(vres[2]))
    if out is None:
        out = empty_nhwc(N, cout, d.Ho, d.Wo, x.dtype, x.device)
    else:
        assert tuple(out.shape) == (N, cout, d.Ho, d.Wo) and out.dtype == x.dtype, (tuple(out.shape), (N, cout, d.Ho, d.Wo))
    d.y_cs = channel_stride(out)
    assert d.y_cs is not None
    if w_strides is not None:
        d.w_os, d.w_ts = w_strides
    ws, ws_bytes = stream_workspace(x.device) if workspace else (None, 0)
    call("fs_conv2d_fwd_ws", _stream(), ctypes.byref(d), _p(x), _p(w_packed), _p(scale), _p(shift), _p(out), _p(stats), ws,
         ws_bytes)
    return out


def pack_weight_frag(w, dtype, cout=None, cin=None):
    """OIHW fp32 3x3 filter (optionally the leading [:cout,:cin] block) -> MFMA-fragment-ordered bank for conv3x3_halo."""
    O, I, R, S = w.shape
    assert (R, S) == (3, 3) and w.dtype == torch.float32 and w.stride(3) == 1 and w.stride(2) == 3
    cout = O if cout is None else cout
    cin = I if cin is None else cin
    n = _lib.lib().fs_packed_weight_frag_elems(cout, cin, dtype_code(dtype))
    out = torch.empty(n, dtype=dtype, device=w.device)
    call("fs_pack_weight_frag", _stream(), _p(w), w.stride(0), w.stride(1), cout, cin, dtype_code(dtype), _p(out))
    return out


def conv3x3_halo(x, w_frag, cout, scale=None, shift=None, relu=False, out=None, stats=None, stride=1, tile=0):
    """3x3 / pad 1 conv at stride 1 or 2 through the halo-tiled kernel (same epilogue contract as conv2d); tile = forced
    output-channel tile (32 / 64 / 128, 0: heuristic)."""
    x_cs = require_nhwc(x, "x")
    N, Cin, H, W = x.shape
    flags = (FS_CONV_RELU if relu else 0) | {0: 0, 32: 0x1000, 64: 0x2000, 128: 0x3000}[tile]
    d = conv_desc(x.shape, x_cs, cout, 3, 3, stride, 1, 0, x.dtype, flags)
    if out is None:
        out = empty_nhwc(N, cout, d.Ho, d.Wo, x.dtype, x.device)
    d.y_cs = channel_stride(out)
    call("fs_conv3x3_s1_fwd", _stream(), ctypes.byref(d), _p(x), _p(w_frag), _p(scale), _p(shift), _p(out), _p(stats))
    return out


def zoom_desc(x_shape, x_cs, cmid, cout, down, up, y_cs, dtype):
    N, Cin, H, W = x_shape
    h, w = (H // 2, W // 2) if down else (H, W)
    Ho, Wo = (2 * h, 2 * w) if up else (h, w)
    return ZoomDesc(N, H, W, Cin, cmid, cout, h, w, Ho, Wo, x_cs, y_cs, dtype_code(dtype), int(bool(down)), int(bool(up)))


def zoom_cell_supported(d):
    return bool(_lib.lib().fs_zoom_cell_supported(ctypes.byref(d)))


def zoom_cell(x, w1_frag, scale1, shift1, w2_frag, scale2, shift2, cmid, cout, down=True, up=True, out=None):
    """A whole zoomed-conv cell in one launch (zoom_cell.hip): [1/2 bilinear] -> conv3x3*s1+b1, ReLU -> conv3x3*s2+b2 ->
    [x2 bilinear] -> ReLU.  Filters in fragment order (pack_weight_frag)."""
    x_cs = require_nhwc(x, "x")
    d = zoom_desc(x.shape, x_cs, cmid, cout, down, up, 0, x.dtype)
    if out is None:
        out = empty_nhwc(d.N, cout, d.Ho, d.Wo, x.dtype, x.device)
    else:
        assert tuple(out.shape) == (d.N, cout, d.Ho, d.Wo) and out.dtype == x.dtype
    d.y_cs = channel_stride(out)
    call("fs_zoom_cell_fwd", _stream(), ctypes.byref(d), _p(x), _p(w1_frag), _p(scale1), _p(shift1), _p(w2_frag), _p(scale2),
         _p(shift2), _p(out))
    return out


def conv2d_wgrad(x, dy, R, S, stride, pad, cout=None):
    """fp32 packed weight gradient [Cout][R][S][Cin] of conv(x) w.r.t. its filter."""
    x_cs = require_nhwc(x, "x")
    dy_cs = require_nhwc(dy, "dy")
    N, Cin, H, W = x.shape
    cout = dy.shape[1] if cout is None else cout
    d = conv_desc(x.shape, x_cs, cout, R, S, stride, pad, dy_cs, x.dtype, 0, (dy.shape[2], dy.shape[3]))
    dw = torch.zeros((cout, R, S, Cin), dtype=torch.float32, device=x.device)
    ws, ws_bytes = stream_workspace(x.device)       # deterministic slab reduction (fs_conv2d_wgrad alone would use fp32 atomics)
    call("fs_conv2d_wgrad_ws", _stream(), ctypes.byref(d), _p(x), _p(dy), _p(dw), R * S * Cin, 1, Cin, ws, ws_bytes)
    return dw


def conv2d_wgrad_into(x, dy, R, S, stride, pad, grad, cout=None):
    """Accumulate the weight gradient straight into `grad[:cout, :Cin]`, a logically-OIHW fp32 tensor whose taps have a
    uniform stride (OIHW-contiguous, or physically [O][R][S][I] = coalesced atomics): a parameter's .grad or a fresh zero
    tensor.  No packed temporary, no unpack pass."""
    x_cs = require_nhwc(x, "x")
    dy_cs = require_nhwc(dy, "dy")
    N, Cin, H, W = x.shape
    cout = dy.shape[1] if cout is None else cout
    assert grad.dtype == torch.float32 and grad.stride(2) == S * grad.stride(3) and grad.shape[0] >= cout and grad.shape[1] >= Cin
    d = conv_desc(x.shape, x_cs, cout, R, S, stride, pad, dy_cs, x.dtype, 0, (dy.shape[2], dy.shape[3]))
    ws, ws_bytes = stream_workspace(x.device)       # slab partials + arrival counters: bit-reproducible, no fp32 atomics
    call("fs_conv2d_wgrad_ws", _stream(), ctypes.byref(d), _p(x), _p(dy), _p(grad), grad.stride(0), grad.stride(1), grad.stride(3), ws, ws_bytes)
    return grad


def conv_stem(x_nchw, w_packed, cout, scale, shift, relu, dtype, out=None):
    N, C, H, W = x_nchw.shape
    assert C == 3 and x_nchw.dtype == torch.float32 and x_nchw.is_contiguous()
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    if out is None:
        out = empty_nhwc(N, cout, Ho, Wo, dtype, x_nchw.device)
    call("fs_conv_stem_fwd", _stream(), N, H, W, cout, _p(x_nchw), _p(w_packed), _p(scale), _p(shift), _p(out),
         channel_stride(out), dtype_code(dtype), int(relu))
    return out


# ---------------------------------------------------------------------------------------------------
# bilinear (align_corners=True)
# ---------------------------------------------------------------------------------------------------
def bilinear(x, size, relu=False, out=None, out_nchw=0, channels=None):
    """out_nchw: 0 -> NHWC view; 1 -> contiguous NCHW fp32; 2 -> contiguous NCHW in x.dtype.
    `channels` overrides C (reading the first C channels of a padded buffer)."""
    x_cs = channel_stride(x)
    N, C, Hi, Wi = x.shape
    C = C if channels is None else channels
    Ho, Wo = int(size[0]), int(size[1])
    if out_nchw:
        odt = torch.float32 if out_nchw == 1 else x.dtype
        if out is None:
            out = torch.empty((N, C, Ho, Wo), dtype=odt, device=x.device)
        y_cs = 0
    else:
        require_nhwc(x, "x")
        if out is None:
            out = empty_nhwc(N, C, Ho, Wo, x.dtype, x.device)
        y_cs = channel_stride(out)
    d = ResizeDesc(N, Hi, Wi, Ho, Wo, C, x_cs, y_cs, dtype_code(x.dtype), int(relu), int(out_nchw))
    call("fs_bilinear_fwd", _stream(), ctypes.byref(d), _p(x), _p(out))
    return out


def bilinear_bwd(dy, y_out, in_shape, relu, dtype, out_nchw=0, dx_cs=None):
    """Gradient w.r.t. the input of `bilinear`; dy NHWC (or contiguous NCHW fp32 when out_nchw)."""
    N, C, Hi, Wi = in_shape
    Ho, Wo = dy.shape[2], dy.shape[3]
    dx = empty_nhwc(N, C, Hi, Wi, dtype, dy.device, cs=dx_cs, zero=bool(out_nchw))
    if out_nchw:
        assert dy.is_contiguous() and dy.dtype == torch.float32
        y_cs = 0
    else:
        y_cs = require_nhwc(dy, "dy")
        if relu:
            assert channel_stride(y_out) == y_cs, "y_out and dy must share a channel stride"
    d = ResizeDesc(N, Hi, Wi, Ho, Wo, C, channel_stride(dx), y_cs, dtype_code(dtype), int(relu), int(out_nchw))
    if out_nchw and Ho >= 2 * Hi and Wo >= 2 * Wi:          # logits up-sample: separable two-pass gather
        d.out_nchw = 1
        ws = torch.empty((N, C, Ho, Wi), dtype=torch.float32, device=dy.device)
        call("fs_bilinear_bwd_nchw", _stream(), ctypes.byref(d), _p(dy), _p(ws), _p(dx))
        return dx
    call("fs_bilinear_bwd", _stream(), ctypes.byref(d), _p(dy), _p(y_out) if relu else None, _p(dx))
    return dx


# ---------------------------------------------------------------------------------------------------
# batch norm / elementwise
# ---------------------------------------------------------------------------------------------------
def _pix(t):
    return t.shape[0] * t.shape[2] * t.shape[3]


def bn_finalize(stats, count, gamma, beta, eps, momentum, running_mean, running_var, num_batches_tracked=None):
    C = stats.numel() // 2
    mean, invstd, scale, shift = torch.empty((4, C), dtype=torch.float32, device=stats.device).unbind(0)
    call("fs_bn_finalize", _stream(), C, int(count), _p(stats), _p(gamma), _p(beta), float(eps), float(momentum),
         _p(running_mean), _p(running_var), _p(mean), _p(invstd), _p(scale), _p(shift), _p(num_batches_tracked))
    return mean, invstd, scale, shift


def bn_train_apply(x, stats, gamma, beta, eps, momentum, running_mean, running_var, num_batches_tracked, relu):
    """fs_bn_finalize + fs_affine_act in one launch: returns (y, saved) with saved = [mean | invstd | scale | shift]."""
    x_cs = require_nhwc(x, "x")
    C = x.shape[1]
    y = empty_nhwc(x.shape[0], C, x.shape[2], x.shape[3], x.dtype, x.device)
    saved = torch.empty(4 * C, dtype=torch.float32, device=x.device)
    call("fs_bn_train_apply", _stream(), _pix(x), C, _p(x), x_cs, _p(stats), _p(gamma), _p(beta), float(eps), float(momentum),
         _p(running_mean), _p(running_var), _p(num_batches_tracked), _p(saved), _p(y), channel_stride(y), dtype_code(x.dtype),
         int(relu))
    return y, saved


def affine_act(x, scale, shift, relu, out=None):
    x_cs = require_nhwc(x, "x")
    if out is None:
        out = empty_nhwc(*x.shape[:1], x.shape[1], x.shape[2], x.shape[3], x.dtype, x.device)
    call("fs_affine_act", _stream(), _pix(x), x.shape[1], _p(x), x_cs, _p(scale), _p(shift), _p(out), channel_stride(out),
         dtype_code(x.dtype), int(relu))
    return out


def channel_stats(x, stats=None):
    x_cs = require_nhwc(x, "x")
    if stats is None:
        stats = zeros_f32(2 * x.shape[1], x.device)
    ws, ws_bytes = stream_workspace(x.device)       # block partials added up in block order: bit-reproducible, no float atomics
    call("fs_channel_stats_ws", _stream(), _pix(x), x.shape[1], 1, _p(x), x_cs, dtype_code(x.dtype), _p(stats), ws, ws_bytes)
    return stats


def bn_backward(z, dy, y_out, mean, invstd, gamma, relu, dgamma_acc=None, dbeta_acc=None):
    """Returns (dz, dgamma, dbeta) for y = relu?(gamma*(z-mean)*invstd + beta); with dgamma_acc/dbeta_acc (fp32 [C]) the two
    parameter gradients are also accumulated into those buffers by the apply pass."""
    z_cs, dy_cs = require_nhwc(z, "z"), require_nhwc(dy, "dy")
    C = z.shape[1]
    y_cs = require_nhwc(y_out, "y_out") if relu else 0
    red = zeros_f32(2 * C, z.device)
    dt = dtype_code(z.dtype)
    ws, ws_bytes = stream_workspace(z.device)
    call("fs_bn_bwd_reduce_ws", _stream(), _pix(z), C, 1, _p(z), z_cs, _p(dy), dy_cs, _p(y_out) if relu else None, y_cs, _p(mean),
         _p(invstd), 0, dt, int(relu), _p(red), ws, ws_bytes)
    dz = empty_nhwc(z.shape[0], C, z.shape[2], z.shape[3], z.dtype, z.device)
    call("fs_bn_bwd_apply", _stream(), _pix(z), C, _p(z), z_cs, _p(dy), dy_cs, _p(y_out) if relu else None, y_cs, _p(mean),
         _p(invstd), _p(gamma), _p(red), _pix(z), dt, int(relu), _p(dz), channel_stride(dz), _p(dgamma_acc), _p(dbeta_acc))
    return dz, red[C:], red[:C]


BN_COL_MAX_PIXELS = 512        # csrc/units.hip: maps up to this many pixels per group take the one-launch column-owner BatchNorm


def bn_act_train(z, gamma, beta, eps, momentum, running_mean, running_var, num_batches_tracked, relu, groups=1):
    """Train-mode BatchNorm(+ReLU) of z with `groups` independent equal parts of the batch (fs_bn_act_train_fwd):
    returns (y, saved) with saved = [groups][mean | invstd | scale | shift]."""
    z_cs = require_nhwc(z, "z")
    C = z.shape[1]
    y = empty_nhwc(z.shape[0], C, z.shape[2], z.shape[3], z.dtype, z.device)
    saved = torch.empty(groups * 4 * C, dtype=torch.float32, device=z.device)
    pixels = _pix(z)
    stats = zeros_f32(groups * 2 * C, z.device) if pixels // groups > BN_COL_MAX_PIXELS else None
    call("fs_bn_act_train_fwd", _stream(), pixels, C, groups, _p(z), z_cs, _p(gamma), _p(beta), float(eps), float(momentum),
         _p(running_mean), _p(running_var), _p(num_batches_tracked), _p(stats), _p(saved), _p(y), channel_stride(y),
         dtype_code(z.dtype), int(relu), *stream_workspace(z.device))
    return y, saved


def bn_act_train_bwd(z, dy, y_out, saved, gamma, relu, groups=1, dgamma_acc=None, dbeta_acc=None):
    """(dz, dgamma, dbeta) of bn_act_train; dgamma/dbeta are summed over the groups."""
    z_cs, dy_cs = require_nhwc(z, "z"), require_nhwc(dy, "dy")
    C = z.shape[1]
    y_cs = require_nhwc(y_out, "y_out") if relu else 0
    red = zeros_f32((groups + 1 if groups > 1 else 1) * 2 * C, z.device)
    dz = empty_nhwc(z.shape[0], C, z.shape[2], z.shape[3], z.dtype, z.device)
    call("fs_bn_act_train_bwd", _stream(), _pix(z), C, groups, _p(z), z_cs, _p(dy), dy_cs, _p(y_out) if relu else None, y_cs,
         _p(saved), _p(gamma), _p(red), dtype_code(z.dtype), int(relu), _p(dz), channel_stride(dz), _p(dgamma_acc), _p(dbeta_acc),
         *stream_workspace(z.device))
    return dz, red[C:2 * C], red[:C]


def copy_channels(x, out):
    call("fs_copy_channels", _stream(), _pix(x), x.shape[1], _p(x), require_nhwc(x, "x"), _p(out), require_nhwc(out, "out"),
         dtype_code(x.dtype))
    return out


def axpy(x, alpha, out, accumulate):
    """out (+)= alpha * x, alpha a 1-element fp32 device tensor."""
    call("fs_axpy_channels", _stream(), _pix(x), x.shape[1], _p(x), require_nhwc(x, "x"), _p(alpha), _p(out),
         require_nhwc(out, "out"), dtype_code(x.dtype), int(accumulate))
    return out


def dot(x, y):
    out = zeros_f32(1, x.device)
    call("fs_dot", _stream(), _pix(x), x.shape[1], _p(x), require_nhwc(x, "x"), _p(y), require_nhwc(y, "y"),
         dtype_code(x.dtype), _p(out))
    return out


def _operand_arrays(tensors):
    n = len(tensors)
    ptrs = (ctypes.c_void_p * n)(*[None if t is None else t.data_ptr() for t in tensors])
    strides = (ctypes.c_int * n)(*[0 if t is None else t.stride(3) for t in tensors])
    return n, ptrs, strides


def weighted_sum(xs, coef, out=None):
    """out = sum_k coef[k] * xs[k]; xs are NHWC views of one shape/dtype, coef a contiguous fp32 device vector."""
    x0 = xs[0]
    for t in xs:
        require_nhwc(t, "operand")
    n, ptrs, strides = _operand_arrays(xs)
    if out is None:
        out = empty_nhwc(*x0.shape, x0.dtype, x0.device)
    call("fs_weighted_sum", _stream(), _pix(x0), x0.shape[1], n, ptrs, strides, coef.data_ptr(), out.data_ptr(), out.stride(3),
         dtype_code(x0.dtype))
    return out


def weighted_sum_bwd(dy, coef, need, outs=None):
    """[coef[k] * dy if need[k] else None for k] (written into `outs` when given)."""
    dy_cs = require_nhwc(dy, "dy")
    if outs is None:
        outs = [empty_nhwc(*dy.shape, dy.dtype, dy.device) if nd else None for nd in need]
    n, ptrs, strides = _operand_arrays(outs)
    call("fs_weighted_sum_bwd", _stream(), _pix(dy), dy.shape[1], n, dy.data_ptr(), dy_cs, coef.data_ptr(), ptrs, strides,
         dtype_code(dy.dtype))
    return outs


def weighted_sum_dots(dy, xs):
    """fp32 vector [<dy, x_k>]."""
    dy_cs = require_nhwc(dy, "dy")
    n, ptrs, strides = _operand_arrays(xs)
    out = zeros_f32(n, dy.device)             # the step's zero arena (one fill per step / graph replay) instead of one fill launch per call
    call("fs_weighted_sum_dots", _stream(), _pix(dy), dy.shape[1], n, dy.data_ptr(), dy_cs, ptrs, strides, dtype_code(dy.dtype),
         out.data_ptr())
    return out


def cat_channels(tensors):
    """torch.cat(dim=1) for NHWC views (model_seg.py:307-331): one fs_copy_channels per operand."""
    N, _, H, W = tensors[0].shape
    total = sum(t.shape[1] for t in tensors)
    out = empty_nhwc(N, total, H, W, tensors[0].dtype, tensors[0].device)
    off = 0
    for t in tensors:
        copy_channels(t, out[:, off:off + t.shape[1]])
        off += t.shape[1]
    return out


def deterministic_on():
    """Is the library in its bit-reproducible mode (fs_set_deterministic / FS_DETERMINISTIC=1)?"""
    return bool(_lib.lib().fs_get_deterministic())


class deterministic:
    """with deterministic(): ...  -  bit-reproducible mode of the library for the duration (fs_set_deterministic): ordered partial sums
    instead of float atomics in the weight-gradient slabs and the BatchNorm reductions of large maps (slower, see the header)."""

    def __init__(self, on=True):
        self.on = bool(on)

    def __enter__(self):
        from . import _lib
        self.prev = _lib.lib().fs_get_deterministic()
        _lib.lib().fs_set_deterministic(int(self.on))
        return self

    def __exit__(self, *exc):
        from . import _lib
        _lib.lib().fs_set_deterministic(self.prev)
