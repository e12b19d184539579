"""torch.autograd.Function layer: the seam between the reference's operator semantics and the HIP kernels.

Each Function is one fused unit of the hot path (forward and backward both run on libfasterseg_hip kernels):
  conv_bn_act      nn.Conv2d/USConv2d -> BatchNorm2d (train or eval) -> [ReLU]     operations.py:125-128,196-200
  conv_bias        Head's biased 1x1 classifier                                     seg_oprs.py:245,273
  factorized_reduce  cat[conv1x1_s2(x), conv1x1_s2(x[:,:,1:,1:])] -> BN -> ReLU     operations.py:521-526
  interpolate      F.interpolate(bilinear, align_corners=True) [+ReLU], NHWC or final NCHW logits
  batch_norm / conv2d  the un-fused forms behind fasterseg_amd.nn.{BatchNorm2d,Conv2d}
  scale_accumulate  acc + coef * x with a device-resident scalar coef               model_search.py:76-78,330-333
There is no eager/ATen fallback: inputs that are not NHWC views are converted with fs_nchw_to_nhwc.
"""
import ctypes
import os
import weakref

import torch

from . import kernels as K

_compute_dtype = torch.float32
_tracer = None      # set by fasterseg_amd.engine while a network is being lowered to a static kernel plan


class SymTensor:
    """Shape-only stand-in for a feature map while engine.InferenceEngine traces a network's forward()."""
    _next = 0

    def __init__(self, shape, dtype, nchw=False):
        self.shape = tuple(int(v) for v in shape)
        self.dtype = dtype
        self.nchw = nchw
        self.storage = None          # (buffer id, channel offset), assigned by the planner
        self.producer = None
        self.id = SymTensor._next
        SymTensor._next += 1

    def size(self, dim=None):
        return self.shape if dim is None else self.shape[dim]

    def dim(self):
        return len(self.shape)


def set_compute_dtype(dtype):
    """Storage/compute dtype used when an operator receives a plain NCHW fp32 tensor (float32 or bfloat16)."""
    global _compute_dtype
    K.dtype_code(dtype)
    _compute_dtype = dtype


def get_compute_dtype():
    return _compute_dtype


class _ToNHWC(torch.autograd.Function):
    """Layout change at the boundary (fs_nchw_to_nhwc); the gradient flows back unchanged (same logical shape)."""

    @staticmethod
    def forward(ctx, x, dtype):
        ctx.in_dtype = x.dtype
        return K.to_nhwc(x, dtype)

    @staticmethod
    def backward(ctx, dy):
        return (dy if dy.dtype == ctx.in_dtype else dy.to(ctx.in_dtype)), None


def as_nhwc(x, dtype=None):
    """`x` as an NHWC view of `dtype` (default: keep an NHWC operand's dtype, else the compute dtype)."""
    if isinstance(x, SymTensor):
        return x
    if dtype is None:
        dtype = x.dtype if K.is_nhwc(x) else _compute_dtype
    if K.is_nhwc(x, dtype):
        return x
    if x.requires_grad and torch.is_grad_enabled():
        return _ToNHWC.apply(x, dtype)
    return K.to_nhwc(x, dtype)


# ---------------------------------------------------------------------------------------------------
# packed-filter cache: re-pack only when the parameter changed (optimizer steps bump Tensor._version)
# ---------------------------------------------------------------------------------------------------
_pack_cache = {}     # id(parameter) -> (weakref to it, {"stamp": (ptr, version, epoch), key: packed})
_weights_epoch = 0   # bumped by optimizers that update parameters without going through autograd's version counters


def bump_weights_epoch():
    global _weights_epoch
    _weights_epoch += 1


def _pack_entry(weight):
    """Cache record of one parameter *object* (identity, not value: tensors do not compare as keys).  A cache keyed on
    data_ptr alone would alias freed-and-reused memory; the weakref both validates identity and evicts on death."""
    k = id(weight)
    rec = _pack_cache.get(k)
    if rec is None or rec[0]() is not weight:
        rec = (weakref.ref(weight, lambda _, k=k: _pack_cache.pop(k, None)), {"stamp": None})
        _pack_cache[k] = rec
    return rec[1]


def packed_weight(weight, dtype, cout=None, cin=None, flip=False, rows=None):
    """Packed copy of (a leading block of) an OIHW parameter; re-packed only when the parameter's storage or version
    counter changed (optimizer steps bump Tensor._version)."""
    if weight.is_cuda and torch.cuda.is_current_stream_capturing():
        return K.pack_weight(weight.detach(), dtype, cout, cin, flip, rows)      # the pack becomes a node of the graph
    entry = _pack_entry(weight)
    stamp = (weight.data_ptr(), weight._version, _weights_epoch)
    if entry["stamp"] != stamp:
        entry.clear()
        entry["stamp"] = stamp
    key = (dtype, cout, cin, flip, rows)
    hit = entry.get(key)
    if hit is None:
        hit = K.pack_weight(weight.detach(), dtype, cout, cin, flip, rows)
        entry[key] = hit
    return hit


# Resident packs: optim.FlatSGD keeps packed copies of every conv filter up to date inside its update kernel.  Train-mode
# convs read the leading [:cout][..][:cin] block of them in place (fs_conv_desc.w_os / w_ts) - no fs_pack_weight launches.
_resident = {}      # id(parameter) -> [weakref, fwd [O][R][S][I], flip [I][R][S][O], version at registration]


def register_resident_pack(param, fwd, flip):
    k = id(param)
    _resident[k] = [weakref.ref(param, lambda _, k=k: _resident.pop(k, None)), fwd, flip, param._version]


def revalidate_resident_pack(param):
    e = _resident.get(id(param))
    if e is not None and e[0]() is param:
        e[3] = param._version


def resident_pack(param, dtype):
    """(fwd pack, flipped pack) of the FULL filter in `dtype`, or None when there is none or the parameter was modified behind
    the optimizer's back (in-place ops bump Tensor._version; FlatSGD's own updates rewrite the packs and do not)."""
    e = _resident.get(id(param))
    if e is None or e[1].dtype != dtype or e[3] != param._version or e[0]() is not param:
        return None
    return e[1], e[2]


def fold_bn(gamma, beta, running_mean, running_var, eps):
    """eval-mode BatchNorm as per-channel scale/shift (C-length vectors; plumbing, not on the pixel path)."""
    scale = gamma.detach().float() * torch.rsqrt(running_var.float() + eps)
    shift = beta.detach().float() - running_mean.float() * scale
    return scale.contiguous(), shift.contiguous()


# > 1 while a module is evaluated ONCE on several inputs concatenated along the batch dimension (model_search: the from-down
# and from-keep inputs of a supernet cell, reference model_search.py:322-329): every train-mode BatchNorm then normalises the
# `_bn_groups` equal parts of the batch independently and applies their running-statistics updates one after the other, i.e.
# the arithmetic of separate evaluations at half the launches (fs_conv_desc.bn_groups).
_bn_groups = 1


class bn_groups:
    def __init__(self, groups):
        self.groups = int(groups)

    def __enter__(self):
        global _bn_groups
        self.prev, _bn_groups = _bn_groups, self.groups

    def __exit__(self, *exc):
        global _bn_groups
        _bn_groups = self.prev


def _zero_stats(c, device):
    return K.zeros_f32(2 * c, device)


# Set by parallel.FlatGradientSync for the duration of a backward pass: parameters whose .grad is a view of its
# (pre-zeroed) flat buffer get their weight gradient accumulated IN PLACE by the wgrad kernel, i.e. without a
# packed temporary, an unpack pass, a zero-filled OIHW tensor and autograd's accumulate-add (4 launches per conv).
_grad_sink = None


def conv_weight_grad(weight, x, dz, R, S, stride, pad, cout, cin):
    """Weight gradient of conv(x)[:cout,:cin]: returns an OIHW fp32 tensor, or None when it was accumulated directly into
    weight.grad (fused gradient accumulation)."""
    sink, g = _grad_sink, weight.grad
    if (sink is not None and g is not None and g.dtype == torch.float32 and g.shape == weight.shape
            and g.stride(2) == S * g.stride(3) and sink.accepts(weight)):
        K.conv2d_wgrad_into(x, dz, R, S, stride, pad, g, cout)
        sink.touched(weight)
        return None
    O, I = weight.shape[0], weight.shape[1]           # physically [O][R][S][I]: the kernel's atomics are then coalesced
    g = torch.zeros((O, R, S, I), dtype=torch.float32, device=weight.device).permute(0, 3, 1, 2)
    K.conv2d_wgrad_into(x, dz, R, S, stride, pad, g, cout)
    return g


def _dgrad(dz, weight, cin, R, S, stride, pad, in_hw, rows=None):
    """data gradient: conv of dz with the 180-degree-rotated, IO-transposed filter (zero insertion for stride 2)."""
    cout = dz.shape[1] if rows is None else min(dz.shape[1], weight.shape[0])
    wf = packed_weight(weight, dz.dtype, cout, cin, flip=True)
    if rows is not None and rows != wf.shape[-1]:          # channel-padded dz (classifier): pad the contraction dim
        wfp = torch.zeros(wf.shape[:-1] + (rows,), dtype=wf.dtype, device=wf.device)
        wfp[..., :wf.shape[-1]] = wf
        wf = wfp
    return K.conv2d(dz, wf, cin, R, S, 1, R - 1 - pad, transposed=(stride == 2), out_hw=in_hw)


def _sink_slot(sink, param, n):
    """The flat-buffer slice behind `param.grad` if the active gradient sink owns it (fused accumulation), else None."""
    g = param.grad
    if sink is not None and g is not None and g.dtype == torch.float32 and g.numel() == n and g.is_contiguous() and sink.accepts(param):
        return g
    return None


class _ConvBNAct(torch.autograd.Function):
    """conv -> BN -> [ReLU].  Training mode goes through the fused-unit entry points (one FFI crossing forward, one
    backward: fs_conv_bn_act_train_fwd / _bwd) and touches as little Python as possible: the supernet calls this ~3000
    times per step on maps of a few thousand pixels."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, running_mean, running_var, num_batches_tracked, cfg):
        stride, pad, relu, training, momentum, eps, cout, cin, groups = cfg
        R, S = weight.shape[2], weight.shape[3]
        assert x.shape[1] == cin, "input has %d channels, conv expects %d" % (x.shape[1], cin)
        if not training:
            wp = packed_weight(weight, x.dtype, cout, cin)
            scale, shift = fold_bn(gamma, beta, running_mean, running_var, eps)
            y = K.conv2d(x, wp, cout, R, S, stride, pad, scale, shift, relu)
            ctx.eval_mode = True
            ctx.mark_non_differentiable(y)
            return y
        N, _, H, W = x.shape
        Ho, Wo = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - S) // stride + 1
        dtype, dev = x.dtype, x.device
        rp = resident_pack(weight, dtype)
        if rp is not None:          # leading block of the resident full-size pack, read in place
            wp, w_os, w_ts = rp[0], R * S * weight.shape[1], weight.shape[1]
        else:
            wp, w_os, w_ts = packed_weight(weight, dtype, cout, cin), 0, 0
        d = K.ConvDesc(N, H, W, cin, cout, R, S, stride, pad, Ho, Wo, x.stride(3), cout, K.dtype_code(dtype),
                       K.FS_CONV_RELU if relu else 0, w_os, w_ts, 0, 0, 0, groups)
        strides = (Ho * Wo * cout, 1, Wo * cout, cout)
        z = torch.empty_strided((N, cout, Ho, Wo), strides, dtype=dtype, device=dev)
        y = torch.empty_strided((N, cout, Ho, Wo), strides, dtype=dtype, device=dev)
        saved = torch.empty(groups * 4 * cout, dtype=torch.float32, device=dev)      # per group: mean | invstd | scale | shift
        stats = K.zeros_f32(groups * 2 * cout, dev)
        ws, ws_bytes = K.stream_workspace(dev)
        K.call("fs_conv_bn_act_train_fwd", K._stream(), ctypes.byref(d), x.data_ptr(), wp.data_ptr(), gamma.data_ptr(),
               beta.data_ptr(), running_mean.data_ptr() if running_mean is not None else None,
               running_var.data_ptr() if running_var is not None else None,
               num_batches_tracked.data_ptr() if num_batches_tracked is not None else None, eps, momentum,
               stats.data_ptr(), saved.data_ptr(), z.data_ptr(), y.data_ptr(), ws, ws_bytes)
        ctx.eval_mode = False
        ctx.cfg = cfg
        ctx.desc = d
        ctx.save_for_backward(x, weight, gamma, beta, z, y if relu else None, saved)
        return y

    @staticmethod
    def backward(ctx, dy):
        if ctx.eval_mode:
            raise RuntimeError("fasterseg_amd: backward through eval-mode conv+BN is not part of the hot path")
        x, weight, gamma, beta, z, y, saved = ctx.saved_tensors
        stride, pad, relu, training, momentum, eps, cout, cin, groups = ctx.cfg
        d = ctx.desc
        R, S = d.R, d.S
        dtype, dev = z.dtype, z.device
        dy = as_nhwc(dy, dtype)
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        sink = _grad_sink
        red = K.zeros_f32((groups + 1 if groups > 1 else 1) * 2 * cout, dev)       # totals, then the per-group partials
        gslot = bslot = None
        if ctx.needs_input_grad[2] and ctx.needs_input_grad[3]:
            gslot = _sink_slot(sink, gamma, cout)
            bslot = _sink_slot(sink, beta, cout) if gslot is not None else None
            if bslot is None:
                gslot = None
        dz = torch.empty_strided(z.shape, z.stride(), dtype=dtype, device=dev)
        gw = gx = wslot = None
        if need_w:
            g = weight.grad
            if (sink is not None and g is not None and g.dtype == torch.float32 and g.shape == weight.shape
                    and g.stride(2) == S * g.stride(3) and sink.accepts(weight)):
                wslot = g
            else:       # physically [O][R][S][I]: the kernel's atomics are then coalesced
                gw = torch.zeros((weight.shape[0], R, S, weight.shape[1]), dtype=torch.float32, device=dev).permute(0, 3, 1, 2)
            wdst = wslot if wslot is not None else gw
        wf, wf_os, wf_ts = None, 0, 0
        if need_x:
            rp = resident_pack(weight, dtype)
            if rp is not None:
                wf, wf_os, wf_ts = rp[1], R * S * weight.shape[0], weight.shape[0]
            else:
                wf = packed_weight(weight, dtype, cout, cin, flip=True)
            gx = torch.empty_strided(x.shape, (d.H * d.W * cin, 1, d.W * cin, cin), dtype=dtype, device=dev)
        K.call("fs_conv_bn_act_train_bwd", K._stream(), ctypes.byref(d), x.data_ptr(), wf.data_ptr() if need_x else None,
               z.data_ptr(), y.data_ptr() if relu else None, dy.data_ptr(), dy.stride(3), saved.data_ptr(), gamma.data_ptr(),
               red.data_ptr(), gslot.data_ptr() if gslot is not None else None, bslot.data_ptr() if bslot is not None else None,
               dz.data_ptr(), wdst.data_ptr() if need_w else None, wdst.stride(0) if need_w else 0,
               wdst.stride(1) if need_w else 0, wdst.stride(3) if need_w else 0, gx.data_ptr() if need_x else None, cin,
               wf_os, wf_ts, *K.stream_workspace(dev))
        if wslot is not None:
            sink.touched(weight)
        dgamma, dbeta = red[cout:2 * cout], red[:cout]
        if gslot is not None:
            sink.touched(gamma)
            sink.touched(beta)
            dgamma = dbeta = None
        return gx, gw, dgamma, dbeta, None, None, None, None


def conv_bn_act(x, weight, gamma, beta, running_mean, running_var, stride, pad, relu, training, momentum=0.1, eps=1e-5,
                cout=None, cin=None, num_batches_tracked=None):
    """`num_batches_tracked` (the BN module's counter, or None) is incremented on the device in training mode."""
    x = as_nhwc(x)
    cout = weight.shape[0] if cout is None else cout
    cin = weight.shape[1] if cin is None else cin
    if _tracer is not None:
        return _tracer.conv(x, weight, (gamma, beta, running_mean, running_var, eps), None, stride, pad, relu, training, cout, cin)
    cfg = (stride, pad, relu, training, momentum, eps, cout, cin, _bn_groups if training else 1)
    return _ConvBNAct.apply(x, weight, gamma, beta, running_mean, running_var, num_batches_tracked if training else None, cfg)


class _StemConvBNAct(torch.autograd.Function):
    """ConvNorm(3, C, 3, stride 2) on the NCHW fp32 image (model_seg.py:193)."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, running_mean, running_var, cfg):
        relu, training, momentum, eps, dtype = cfg
        cout = weight.shape[0]
        wp = packed_weight(weight, torch.float32)
        xc = x.detach().float().contiguous()
        if not training:
            scale, shift = fold_bn(gamma, beta, running_mean, running_var, eps)
            y = K.conv_stem(xc, wp, cout, scale, shift, relu, dtype)
            ctx.eval_mode = True
            ctx.mark_non_differentiable(y)
            return y
        z = K.conv_stem(xc, wp, cout, None, None, False, dtype)
        stats = K.channel_stats(z)
        count = z.shape[0] * z.shape[2] * z.shape[3]
        mean, invstd, scale, shift = K.bn_finalize(stats, count, gamma.detach(), beta.detach(), eps, momentum,
                                                   running_mean, running_var)
        y = K.affine_act(z, scale, shift, relu)
        ctx.eval_mode = False
        ctx.cfg = cfg
        ctx.save_for_backward(xc, weight, gamma, z, y if relu else None, mean, invstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        if ctx.eval_mode:
            raise RuntimeError("fasterseg_amd: backward through eval-mode conv+BN is not part of the hot path")
        xc, weight, gamma, z, y, mean, invstd = ctx.saved_tensors
        relu = ctx.cfg[0]
        dy = as_nhwc(dy, z.dtype)
        dz, dgamma, dbeta = K.bn_backward(z, dy, y, mean, invstd, gamma.detach(), relu)
        gw = gx = None
        vec = K.vec_of(z.dtype)
        if ctx.needs_input_grad[1]:
            xp = K.to_nhwc(xc, z.dtype)                       # 3 channels zero-padded to one vector
            xp_full = xp.as_strided((xp.shape[0], vec, xp.shape[2], xp.shape[3]), xp.stride(), xp.storage_offset())
            dw = K.conv2d_wgrad(xp_full, dz, 3, 3, 2, 1)      # [cout][3][3][vec]
            gpad = torch.zeros((weight.shape[0], vec, 3, 3), dtype=torch.float32, device=weight.device)
            K.unpack_weight_grad(dw, gpad, weight.shape[0], vec)
            gw = gpad[:, :3].contiguous()
        if ctx.needs_input_grad[0]:
            # data gradient with 3 output channels: generic transposed conv into a channel-padded buffer, then NCHW
            wf = packed_weight(weight, z.dtype, flip=True)                    # [3][3][3][cout]
            dx = K.conv2d(dz, wf, 3, 3, 3, 1, 1, transposed=True, out_hw=(xc.shape[2], xc.shape[3]))
            gx = K.to_nchw(dx)
        return gx, gw, dgamma, dbeta, None, None, None


def stem_conv_bn_act(x, weight, gamma, beta, running_mean, running_var, relu, training, momentum=0.1, eps=1e-5, dtype=None):
    if _tracer is not None:
        return _tracer.stem(x, weight, (gamma, beta, running_mean, running_var, eps), relu, training)
    cfg = (relu, training, momentum, eps, dtype or _compute_dtype)
    return _StemConvBNAct.apply(x, weight, gamma, beta, running_mean, running_var, cfg)


CLS_PAD = 32   # classifier logits live in a 32-channel NHWC buffer (19 valid, pad lanes zero)


class _ConvBias(torch.autograd.Function):
    """Conv without BN, optional bias: Head.conv_1x1 (seg_oprs.py:245), plain nn.Conv2d and a bare USConv2d slice."""

    @staticmethod
    def forward(ctx, x, weight, bias, cfg):
        stride, pad, cout, cin = cfg
        R, S = weight.shape[2], weight.shape[3]
        assert x.shape[1] == cin, "input has %d channels, conv expects %d" % (x.shape[1], cin)
        vec = K.vec_of(x.dtype)
        cpad = cout if cout % vec == 0 else K.round_up(cout, CLS_PAD)
        wp = packed_weight(weight, x.dtype, cout, cin)
        N, _, H, W = x.shape
        Ho, Wo = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - S) // stride + 1
        buf = K.empty_nhwc(N, cpad, Ho, Wo, x.dtype, x.device, zero=(cpad != cout))
        y = buf[:, :cout]
        shift = bias.detach().float()[:cout].contiguous() if bias is not None else None
        K.conv2d(x, wp, cout, R, S, stride, pad, None, shift, False, out=y)
        ctx.save_for_backward(x, weight)
        ctx.meta = (stride, pad, cpad, bias is not None, cout, cin)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        stride, pad, cpad, has_bias, cout, cin = ctx.meta
        R, S = weight.shape[2], weight.shape[3]
        N, _, Ho, Wo = dy.shape
        if cpad != cout:        # widen the gradient to the padded channel count (pad lanes zero)
            if K.is_nhwc(dy, x.dtype) and K.channel_stride(dy) == cpad and dy.storage_offset() * dy.element_size() % 16 == 0:
                dyp = dy.as_strided((N, cpad, Ho, Wo), dy.stride(), dy.storage_offset())   # already a padded buffer
            else:
                dyp = K.empty_nhwc(N, cpad, Ho, Wo, x.dtype, x.device, zero=True)
                dyp[:, :cout].copy_(dy)
        else:
            dyp = as_nhwc(dy, x.dtype)
        gw = gb = gx = None
        if ctx.needs_input_grad[1]:
            if cpad != cout:
                dw = K.conv2d_wgrad(x, dyp, R, S, stride, pad)
                gfull = torch.zeros((cpad, cin, R, S), dtype=torch.float32, device=x.device)
                K.unpack_weight_grad(dw, gfull, cpad, cin)
                gw = torch.zeros_like(weight, dtype=torch.float32)
                gw[:cout, :cin] = gfull[:cout]
            else:
                gw = conv_weight_grad(weight, x, dyp, R, S, stride, pad, cout, cin)
        if has_bias and ctx.needs_input_grad[2]:
            gb = K.channel_stats(dyp)[:cout].clone()
        if ctx.needs_input_grad[0]:
            gx = _dgrad(dyp, weight, cin, R, S, stride, pad, (x.shape[2], x.shape[3]), rows=cpad)
        return gx, gw, gb, None


def conv_bias(x, weight, bias, stride=1, pad=0, cout=None, cin=None):
    cout = weight.shape[0] if cout is None else cout
    cin = weight.shape[1] if cin is None else cin
    if _tracer is not None:
        return _tracer.conv(x, weight, None, bias, stride, pad, False, False, cout, cin)
    return _ConvBias.apply(as_nhwc(x), weight, bias, (stride, pad, cout, cin))


def conv_slice(x, weight, cout, cin, stride, pad):
    """Bias-free conv on the leading [:cout,:cin] block of `weight` (USConv2d.forward, slimmable_ops.py:42-47)."""
    return conv_bias(x, weight, None, stride, pad, cout, cin)


class _FactorizedReduce(torch.autograd.Function):
    """stride-2 'skip': two 1x1 stride-2 convs (the second on x[:,:,1:,1:], expressed as pad=-1) written into the two
    channel halves of one buffer, BN statistics from the conv epilogues, then BN+ReLU (operations.py:521-526)."""

    @staticmethod
    def forward(ctx, x, w1, w2, gamma, beta, running_mean, running_var, cfg):
        training, momentum, eps, half, cin, groups, nbt = cfg
        N, _, H, W = x.shape
        assert H % 2 == 0 and W % 2 == 0, "FactorizedReduce needs even spatial dims (reference torch.cat would fail too)"
        Ho, Wo = H // 2, W // 2
        p1, p2 = packed_weight(w1, x.dtype, half, cin), packed_weight(w2, x.dtype, half, cin)
        out_hw = (Ho, Wo)
        if not training:
            scale, shift = fold_bn(gamma, beta, running_mean, running_var, eps)
            y = K.empty_nhwc(N, 2 * half, Ho, Wo, x.dtype, x.device)
            K.conv2d(x, p1, half, 1, 1, 2, 0, scale[:half].contiguous(), shift[:half].contiguous(), True, out=y[:, :half],
                     out_hw=out_hw)
            K.conv2d(x, p2, half, 1, 1, 2, -1, scale[half:].contiguous(), shift[half:].contiguous(), True, out=y[:, half:],
                     out_hw=out_hw)
            ctx.eval_mode = True
            ctx.mark_non_differentiable(y)
            return y
        z = K.empty_nhwc(N, 2 * half, Ho, Wo, x.dtype, x.device)
        K.conv2d(x, p1, half, 1, 1, 2, 0, out=z[:, :half], out_hw=out_hw)
        K.conv2d(x, p2, half, 1, 1, 2, -1, out=z[:, half:], out_hw=out_hw)
        y, saved = K.bn_act_train(z, gamma.detach(), beta.detach(), eps, momentum, running_mean, running_var, nbt, True, groups)
        ctx.eval_mode = False
        ctx.cfg = cfg
        ctx.save_for_backward(x, w1, w2, gamma, z, y, saved)
        return y

    @staticmethod
    def backward(ctx, dy):
        if ctx.eval_mode:
            raise RuntimeError("fasterseg_amd: backward through eval-mode conv+BN is not part of the hot path")
        x, w1, w2, gamma, z, y, saved = ctx.saved_tensors
        training, momentum, eps, half, cin, groups, nbt = ctx.cfg
        dy = as_nhwc(dy, z.dtype)
        dz, dgamma, dbeta = K.bn_act_train_bwd(z, dy, y, saved, gamma.detach(), True, groups)
        da, db = dz[:, :half], dz[:, half:]
        g1 = g2 = gx = None
        if ctx.needs_input_grad[1]:
            g1 = conv_weight_grad(w1, x, da, 1, 1, 2, 0, half, cin)
        if ctx.needs_input_grad[2]:
            g2 = conv_weight_grad(w2, x, db, 1, 1, 2, -1, half, cin)
        if ctx.needs_input_grad[0]:
            hw = (x.shape[2], x.shape[3])
            gx = _dgrad(da, w1, cin, 1, 1, 2, 0, hw)
            g_odd = _dgrad(db, w2, cin, 1, 1, 2, -1, hw)      # touches only odd (h,w); disjoint from gx's even taps
            K.axpy(g_odd, _one(x.device), gx, True)
        return gx, g1, g2, dgamma, dbeta, None, None, None


_ones = {}


def _one(device):
    key = str(device)
    if key not in _ones:
        _ones[key] = torch.ones(1, dtype=torch.float32, device=device)
    return _ones[key]


def factorized_reduce(x, w1, w2, gamma, beta, running_mean, running_var, training, momentum=0.1, eps=1e-5, half=None,
                      cin=None, num_batches_tracked=None):
    x = as_nhwc(x)
    half = w1.shape[0] if half is None else half
    cin = w1.shape[1] if cin is None else cin
    if _tracer is not None:
        return _tracer.factorized_reduce(x, w1, w2, (gamma, beta, running_mean, running_var, eps), training, half, cin)
    return _FactorizedReduce.apply(x, w1, w2, gamma, beta, running_mean, running_var,
                                   (training, momentum, eps, half, cin, _bn_groups if training else 1,
                                    num_batches_tracked if training else None))


class _Interpolate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, size, relu, out_nchw):
        if out_nchw:
            cs = K.channel_stride(x)
            y = K.bilinear(x, size, out_nchw=out_nchw, channels=x.shape[1])
            ctx.meta = (tuple(x.shape), False, x.dtype, out_nchw, cs)
            return y
        y = K.bilinear(x, size, relu=relu)
        ctx.meta = (tuple(x.shape), relu, x.dtype, 0, None)
        if relu:
            ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        in_shape, relu, dtype, out_nchw, cs = ctx.meta
        if out_nchw:
            dyc = dy.float().contiguous()
            return K.bilinear_bwd(dyc, None, in_shape, False, dtype, out_nchw=1, dx_cs=cs), None, None, None
        y = ctx.saved_tensors[0] if relu else None
        dy = as_nhwc(dy, dtype)
        if relu and K.channel_stride(dy) != K.channel_stride(y):
            dy = K.copy_channels(dy, K.empty_nhwc(*y.shape, dtype, dy.device, cs=K.channel_stride(y)))
        return K.bilinear_bwd(dy, y, in_shape, relu, dtype), None, None, None


def interpolate(x, size=None, scale_factor=None, relu=False, out_nchw=0):
    """F.interpolate(x, size|scale_factor, mode='bilinear', align_corners=True) on the HIP kernel.
    out_nchw=1 returns a contiguous NCHW fp32 tensor (final logits, model_seg.py:359-365)."""
    if size is None:
        size = (int(x.shape[2] * scale_factor), int(x.shape[3] * scale_factor))
    if _tracer is not None:
        return _tracer.resize(x, (int(size[0]), int(size[1])), relu, out_nchw)
    x = as_nhwc(x)
    if out_nchw and K.channel_stride(x) < K.round_up(x.shape[1], 4):
        raise ValueError("NCHW-output resize needs a channel-padded input buffer")
    return _Interpolate.apply(x, (int(size[0]), int(size[1])), relu, out_nchw)


class _BatchNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, training, momentum, eps, relu):
        if not training:
            scale, shift = fold_bn(gamma, beta, running_mean, running_var, eps)
            y = K.affine_act(x, scale, shift, relu)
            ctx.eval_mode = True
            ctx.mark_non_differentiable(y)
            return y
        stats = K.channel_stats(x)
        count = x.shape[0] * x.shape[2] * x.shape[3]
        mean, invstd, scale, shift = K.bn_finalize(stats, count, gamma.detach(), beta.detach(), eps, momentum, running_mean,
                                                   running_var)
        y = K.affine_act(x, scale, shift, relu)
        ctx.eval_mode = False
        ctx.relu = relu
        ctx.save_for_backward(x, gamma, y if relu else None, mean, invstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        if ctx.eval_mode:
            raise RuntimeError("fasterseg_amd: backward through eval-mode BN is not part of the hot path")
        x, gamma, y, mean, invstd = ctx.saved_tensors
        dz, dgamma, dbeta = K.bn_backward(x, as_nhwc(dy, x.dtype), y, mean, invstd, gamma.detach(), ctx.relu)
        return dz, dgamma, dbeta, None, None, None, None, None, None


def batch_norm(x, gamma, beta, running_mean, running_var, training, momentum=0.1, eps=1e-5, relu=False):
    return _BatchNorm.apply(as_nhwc(x), gamma, beta, running_mean, running_var, training, momentum, eps, relu)


class _ScaleAccumulate(torch.autograd.Function):
    """out = acc + coef * x   (acc may be None).  coef is a 0-dim/1-element device tensor, so architecture weights
    never leave the GPU; its gradient is the full-tensor dot <dy, x> (fs_dot)."""

    @staticmethod
    def forward(ctx, acc, x, coef):
        c = coef.detach().float().reshape(1).contiguous()
        out = K.empty_nhwc(*x.shape, x.dtype, x.device)
        if acc is None:
            K.axpy(x, c, out, False)
        else:
            K.copy_channels(acc, out)
            K.axpy(x, c, out, True)
        ctx.has_acc = acc is not None
        ctx.save_for_backward(x, c)
        ctx.coef_shape = coef.shape
        return out

    @staticmethod
    def backward(ctx, dy):
        x, c = ctx.saved_tensors
        dy = as_nhwc(dy, x.dtype)
        gacc = dy if (ctx.has_acc and ctx.needs_input_grad[0]) else None
        gx = gc = None
        if ctx.needs_input_grad[1]:
            gx = K.axpy(dy, c, K.empty_nhwc(*x.shape, x.dtype, x.device), False)
        if ctx.needs_input_grad[2]:
            gc = K.dot(dy, x).reshape(ctx.coef_shape)
        return gacc, gx, gc


def scale_accumulate(acc, x, coef):
    x = as_nhwc(x)
    if acc is not None:
        acc = as_nhwc(acc, x.dtype)
    if not torch.is_tensor(coef):
        coef = torch.tensor(float(coef), dtype=torch.float32, device=x.device)
    return _ScaleAccumulate.apply(acc, x, coef)


def _own(grad, for_leaf):
    """Coefficient gradients are slices of the step's zero arena (kernels.zero_pool: valid until the next step clears it).  Every
    consumer inside the product copies or accumulates them (the coefficients are computed tensors: unbind / select / mul backward) -
    but an AccumulateGrad node of a LEAF with no .grad yet may adopt the incoming tensor as .grad, which must then own its memory.
    LIFETIME: a NON-leaf coefficient's gradient that escapes the backward - torch.autograd.grad(loss, coef), coef.retain_grad(),
    a tensor hook that keeps its argument - is a VIEW of the arena: read or clone it before the next FlatGradientSync.prepare() /
    zero_pool.reset(), which re-zeroes the arena (inside a captured pass: before the next replay)."""
    return grad.clone() if for_leaf and grad is not None else grad


class _WeightedSum(torch.autograd.Function):
    """out = sum_k coef[k] * x_k in one launch (MixedOp's five primitives, the beta mixing of two cell inputs); coef is a
    contiguous fp32 device vector.  Backward: one launch for all dx_k, one for all d coef[k] (only when coef needs it)."""

    @staticmethod
    def forward(ctx, coef, *xs):
        c = coef.detach()
        if c.dtype != torch.float32 or not c.is_contiguous():
            c = c.float().contiguous()
        out = K.weighted_sum(xs, c)
        ctx.n = len(xs)
        ctx.coef_leaf = coef.is_leaf          # (a leaf's AccumulateGrad may adopt the gradient tensor itself: see _own)
        ctx.save_for_backward(c, *(xs if ctx.needs_input_grad[0] else ()))
        return out

    @staticmethod
    def backward(ctx, dy):
        c = ctx.saved_tensors[0]
        dy = as_nhwc(dy)
        gxs = K.weighted_sum_bwd(dy, c, ctx.needs_input_grad[1:])
        gc = None
        if ctx.needs_input_grad[0]:
            gc = _own(K.weighted_sum_dots(dy, [as_nhwc(t, dy.dtype) for t in ctx.saved_tensors[1:]]), ctx.coef_leaf)
        return (gc,) + tuple(gxs)


class _PairMerge(torch.autograd.Function):
    """out = coef[0] * x[:n] + coef[1] * x[n:] for the output of a module evaluated once on two inputs concatenated along the
    batch (bn_groups): the beta mixing of a cell's from-down / from-keep results (model_search.py:331-332).  Backward writes
    both halves of dx with one launch - no slice / zero-pad / add nodes."""

    @staticmethod
    def forward(ctx, coef, x, dest=None):
        c = coef.detach()
        if c.dtype != torch.float32 or not c.is_contiguous():
            c = c.float().contiguous()
        n = x.shape[0] // 2
        out = K.weighted_sum([x[:n], x[n:]], c, out=dest[0] if dest else None)
        ctx.coef_leaf = coef.is_leaf
        ctx.save_for_backward(c, x)
        return out

    @staticmethod
    def backward(ctx, dy):
        c, x = ctx.saved_tensors
        n = x.shape[0] // 2
        dy = as_nhwc(dy, x.dtype)
        gx = gc = None
        if ctx.needs_input_grad[1]:
            gx = K.empty_nhwc(*x.shape, x.dtype, x.device)
            K.weighted_sum_bwd(dy, c, (True, True), outs=[gx[:n], gx[n:]])
        if ctx.needs_input_grad[0]:
            gc = _own(K.weighted_sum_dots(dy, [x[:n], x[n:]]), ctx.coef_leaf)
        return gc, gx, None


def pair_merge(x, coef, dest=None):
    """dest: an NHWC tensor of the result's shape to write into (a PairBuffers half), or None."""
    x = as_nhwc(x)
    assert x.shape[0] % 2 == 0 and coef.numel() == 2
    return _PairMerge.apply(coef, x, [dest] if dest is not None else None)


class _PairMergeGroup(torch.autograd.Function):
    """k pair_merge evaluations (the beta mixing of every pair-batched cell output of one supernet layer) as ONE grouped launch forward and
    one backward (fs_exec_program_group over one-command programs; csrc/group.h wsum_group / wsum_bwd_group)."""

    @staticmethod
    def forward(ctx, meta, *tensors):
        from . import program
        progs, dests = meta
        k = len(progs)
        outs, saved, slots = [], [], []
        for i, prog in enumerate(progs):
            coef, x = tensors[2 * i], tensors[2 * i + 1]
            c = coef.detach()
            if c.dtype != torch.float32 or not c.is_contiguous():
                c = c.float().contiguous()
            n2, C, H, W = x.shape
            out = dests[i] if dests[i] is not None else K.empty_nhwc(n2 // 2, C, H, W, x.dtype, x.device, cs=C)
            slots.append((None, x.data_ptr(), c.data_ptr(), out.data_ptr(), None, None, None, None, None, None, None, None))
            outs.append(out)
            saved += [c, x]
        program.run_group(progs, False, slots)
        ctx.progs = progs
        ctx.coef_leaf = [tensors[2 * i].is_leaf for i in range(k)]
        ctx.save_for_backward(*saved)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *dys):
        from . import program
        progs = ctx.progs
        saved = ctx.saved_tensors
        slots, keep, gxs, zbs = [], [], [], []
        for i, prog in enumerate(progs):
            c, x = saved[2 * i], saved[2 * i + 1]
            n2, C, H, W = x.shape
            dy = dys[i]
            if not (K.is_nhwc(dy, x.dtype) and K.channel_stride(dy) == C):
                dy = K.copy_channels(as_nhwc(dy, x.dtype), K.empty_nhwc(n2 // 2, C, H, W, x.dtype, x.device, cs=C))
            gx = K.empty_nhwc(n2, C, H, W, x.dtype, x.device, cs=C)
            zb = zero_arena(prog.zb_bytes, x.device)
            slots.append((None, x.data_ptr(), c.data_ptr(), None, None, dy.data_ptr(), None, gx.data_ptr(), None, None, None,
                          zb.data_ptr() if zb is not None else None))
            keep.append(dy)
            gxs.append(gx)
            zbs.append(zb)
        program.run_group(progs, True, slots)
        grads = [None]
        for i, prog in enumerate(progs):
            gc = None
            if prog.need_coef:
                gc = _own(zbs[i][:2], ctx.coef_leaf[i])
            grads += [gc, gxs[i]]
        return tuple(grads)


def pair_merge_group(xs, coefs, dests=None):
    """[pair_merge(x, coef) for x, coef] in one grouped launch per direction; dests[i]: where result i is written (or None).  Falls back to
    the single launches when an input is not a dense NHWC map."""
    from . import program
    xs = [as_nhwc(x) for x in xs]
    dests = list(dests) if dests is not None else [None] * len(xs)
    if len(xs) < 2 or len(xs) > program.MAX_GROUP or any(K.channel_stride(x) != x.shape[1] for x in xs) or not xs[0].is_cuda:
        return [pair_merge(x, c, d) for x, c, d in zip(xs, coefs, dests)]
    progs = [program.pair_merge_program(tuple(x.shape), x.dtype, bool(c.requires_grad) and torch.is_grad_enabled()) for x, c in zip(xs, coefs)]
    flat = []
    for x, c in zip(xs, coefs):
        flat += [c, x]
    return list(_PairMergeGroup.apply((tuple(progs), dests), *flat))


class PairBuffers:
    """Destination planning for batch_pair (round 6).  Every cell output of a supernet layer has exactly ONE consumer cell in the next
    layer (reference search/model_search.py:310-333: `keep` feeds the same scale, `down` the next one), and a consumer fed from both
    scales evaluates its MixedOps once on the two inputs concatenated along the batch (batch_pair).  Round 5 copied both inputs into
    the joint buffer (two launches per pair, 216 per C3 step); now the joint buffer is allocated when its first half is PRODUCED and
    the producers (the layer call's weighted sums, the beta merges) write straight into their half: `half(key, which, like)` hands out
    the slice, `joint(a, b)` recognises two halves of one buffer and returns it without a launch."""

    def __init__(self):
        self.bufs = {}

    def half(self, key, which, shape, dtype, device):
        """The `which`-th (0 / 1) half of the joint NHWC buffer `key` for halves of `shape` (n, C, H, W), or None when the other half was
        planned with another shape (the consumer then copies, as before)."""
        n, c, h, w = shape
        buf = self.bufs.get(key)
        if buf is None:
            buf = self.bufs[key] = K.empty_nhwc(2 * n, c, h, w, dtype, device, cs=c)
        if tuple(buf.shape) != (2 * n, c, h, w) or buf.dtype != dtype:
            return None
        return buf[:n] if which == 0 else buf[n:]

    def joint(self, a, b):
        n = a.shape[0]
        for buf in self.bufs.values():
            if buf.data_ptr() == a.data_ptr() and buf.shape[0] == 2 * n and buf[n:].data_ptr() == b.data_ptr() and a.shape == b.shape \
                    and a.stride() == buf.stride() and b.stride() == buf.stride():
                return buf
        return None


_pair_buffers = None          # the PairBuffers of the forward pass in progress (model_search.Network_Multi_Path.forward)
# [parameters the launch programs of the forward in progress will write in backward, one entry per (program, parameter)] while a caller
# wants them (train_step: comm / compute overlap of the last pass, parallel.FlatGradientSync.final_pass); [None] marks the log unusable
_touch_log = None


class _BatchPair(torch.autograd.Function):
    """torch.cat([a, b], dim=0) of two NHWC maps of one shape (one copy launch each - none when the producers already wrote the two
    halves of one PairBuffers buffer); backward = the two halves of dy."""

    @staticmethod
    def forward(ctx, a, b):
        n, c, h, w = a.shape
        ctx.n = n
        joint = _pair_buffers.joint(a, b) if _pair_buffers is not None else None
        if joint is not None:
            return joint
        out = K.empty_nhwc(2 * n, c, h, w, a.dtype, a.device)
        K.copy_channels(a, out[:n])
        K.copy_channels(b, out[n:])
        return out

    @staticmethod
    def backward(ctx, dy):
        return dy[:ctx.n], dy[ctx.n:]


def batch_pair(a, b):
    a = as_nhwc(a)
    b = as_nhwc(b, a.dtype)
    assert a.shape == b.shape
    return _BatchPair.apply(a, b)


def zero_arena(nbytes, device):
    """`nbytes` of zero-initialised fp32 scratch for one direction of a launch program (BN statistics / reduction accumulators, the
    coefficient gradients): a 256-byte aligned slice of the step's zero arena (ONE fill per step; inside a capture one captured fill per
    replay), or a fresh torch.zeros when no step is active.  None for 0 bytes."""
    if not nbytes:
        return None
    if _ZERO_MEMSET:
        return _memset_zeros(nbytes, device)
    return K.zero_pool.take(nbytes // 4, device, align=64)


# Diagnosis switch (round 6, DESIGN section 7 "capture fault"): FS_ZERO_MEMSET=1 restores what rounds 3-4 did - every launch program
# clears its accumulators with its own hipMemsetAsync (inside a capture: one memset node per program and direction).
_ZERO_MEMSET = bool(int(os.environ.get("FS_ZERO_MEMSET", "0")))
_memset_lists = {}


def _memset_zeros(nbytes, device):
    from . import program
    t = torch.empty(nbytes // 4, dtype=torch.float32, device=device)
    lst = _memset_lists.get(nbytes)
    if lst is None:
        l = program._List()
        l.emit(program.OP_MEMSET, program.Ref(1, 0), nbytes)
        lst = _memset_lists[nbytes] = l.finish()
    slots = (ctypes.c_void_p * program.N_SLOTS)(None, t.data_ptr())
    K.call("fs_exec_program", K._stream(), lst[0], lst[1], lst[2], slots, program.N_SLOTS)
    return t


class _MixedOpProgram(torch.autograd.Function):
    """A whole supernet MixedOp (five primitives + weighted sum) replayed from its pre-built launch programs
    (fasterseg_amd.program): one FFI crossing and three arena allocations per direction instead of ~12 autograd nodes."""

    @staticmethod
    def forward(ctx, x, coef, prog):
        c = coef.detach()
        if c.dtype != torch.float32 or not c.is_contiguous():
            c = c.float().contiguous()
        dev = x.device
        save = torch.empty(prog.save_bytes, dtype=torch.uint8, device=dev)
        tmp = torch.empty(prog.tmpf_bytes, dtype=torch.uint8, device=dev)
        N, C, H, W = prog.out_shape
        out = torch.empty_strided((N, C, H, W), (H * W * C, 1, W * C, C), dtype=x.dtype, device=dev)
        zf = zero_arena(prog.zf_bytes, dev)
        prog.run(prog.f_words, prog.f_n, prog.f_blob,
                 (None, x.data_ptr(), c.data_ptr(), out.data_ptr(), save.data_ptr(), None, None, None, tmp.data_ptr(),
                  K.stream_workspace(dev)[0], zf.data_ptr() if zf is not None else None, None))
        if _touch_log is not None:
            _touch_log.extend(prog.touched)
        ctx.prog = prog
        ctx.coef_leaf = coef.is_leaf
        ctx.save_for_backward(x, c, save)
        return out

    @staticmethod
    def backward(ctx, dy):
        prog = ctx.prog
        x, c, save = ctx.saved_tensors
        N, C, H, W = prog.out_shape
        if not (K.is_nhwc(dy, x.dtype) and K.channel_stride(dy) == C):          # the program reads a dense gradient
            dy = K.copy_channels(as_nhwc(dy, x.dtype), K.empty_nhwc(N, C, H, W, x.dtype, x.device))
        tmp = torch.empty(prog.tmpb_bytes, dtype=torch.uint8, device=x.device)
        gx = None
        if prog.need_x:
            n, ci, h, w = x.shape
            gx = torch.empty_strided((n, ci, h, w), (h * w * ci, 1, w * ci, ci), dtype=x.dtype, device=x.device)
        zb = zero_arena(prog.zb_bytes, x.device)
        prog.run(prog.b_words, prog.b_n, prog.b_blob,
                 (None, x.data_ptr(), c.data_ptr(), None, save.data_ptr(), dy.data_ptr(), tmp.data_ptr(),
                  gx.data_ptr() if gx is not None else None, None, K.stream_workspace(x.device)[0], None,
                  zb.data_ptr() if zb is not None else None))
        if prog.touched:
            sink = _grad_sink
            for p in prog.touched:
                sink.touched(p)
        gc = None
        if prog.need_coef:
            gc = _own(zb[prog.gcoef_off // 4:prog.gcoef_off // 4 + c.numel()].reshape(c.shape), ctx.coef_leaf)
        return gx, gc, None


def mixed_op_program(x, coef, prog):
    return _MixedOpProgram.apply(x, coef, prog)


class _MixedOpProgramGroup(torch.autograd.Function):
    """k MixedOps of one supernet layer (they only depend on the previous layer, reference model_search.py:310-333) replayed in
    LOCKSTEP from their launch programs (fs_exec_program_group): the convolutions, weight gradients and data gradients at the same
    position of the k programs are one grouped launch each.  One autograd node for the k outputs, so backward is grouped as well."""

    @staticmethod
    def forward(ctx, meta, *tensors):
        from . import program
        progs, dests = meta          # dests[i]: a dense NHWC tensor of program i's output shape to write into (a PairBuffers half), or None
        k = len(progs)
        assert len(tensors) == 2 * k
        outs, saved, slots, scratch = [], [], [], []
        for i, prog in enumerate(progs):
            x, coef = tensors[2 * i], tensors[2 * i + 1]
            c = coef.detach()
            if c.dtype != torch.float32 or not c.is_contiguous():
                c = c.float().contiguous()
            dev = x.device
            save = torch.empty(prog.save_bytes, dtype=torch.uint8, device=dev)
            tmp = torch.empty(prog.tmpf_bytes, dtype=torch.uint8, device=dev)
            N, C, H, W = prog.out_shape
            out = dests[i] if dests is not None and dests[i] is not None else None
            if out is None:
                out = torch.empty_strided((N, C, H, W), (H * W * C, 1, W * C, C), dtype=x.dtype, device=dev)
            else:
                assert tuple(out.shape) == (N, C, H, W) and out.stride() == (H * W * C, 1, W * C, C) and out.dtype == x.dtype
            zf = zero_arena(prog.zf_bytes, dev)
            slots.append((None, x.data_ptr(), c.data_ptr(), out.data_ptr(), save.data_ptr(), None, None, None, tmp.data_ptr(),
                          K.stream_workspace(dev)[0], zf.data_ptr() if zf is not None else None, None))
            outs.append(out)
            saved += [x, c, save]
            scratch.append((tmp, zf))             # forward scratch: alive until the launches are enqueued (stream-ordered reuse after)
        program.run_group(progs, False, slots)
        del scratch
        if _touch_log is not None:
            for prog in progs:
                _touch_log.extend(prog.touched)
        ctx.progs = progs
        ctx.coef_leaf = [tensors[2 * i + 1].is_leaf for i in range(k)]
        ctx.save_for_backward(*saved)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *dys):
        from . import program
        progs = ctx.progs
        saved = ctx.saved_tensors
        k = len(progs)
        slots, tmps, gxs = [], [], []
        for i, prog in enumerate(progs):
            x, c, save = saved[3 * i], saved[3 * i + 1], saved[3 * i + 2]
            N, C, H, W = prog.out_shape
            dy = dys[i]
            if not (K.is_nhwc(dy, x.dtype) and K.channel_stride(dy) == C):          # the program reads a dense gradient
                dy = K.copy_channels(as_nhwc(dy, x.dtype), K.empty_nhwc(N, C, H, W, x.dtype, x.device))
            tmp = torch.empty(prog.tmpb_bytes, dtype=torch.uint8, device=x.device)
            gx = None
            if prog.need_x:
                n, ci, h, w = x.shape
                gx = torch.empty_strided((n, ci, h, w), (h * w * ci, 1, w * ci, ci), dtype=x.dtype, device=x.device)
            zb = zero_arena(prog.zb_bytes, x.device)
            slots.append((None, x.data_ptr(), c.data_ptr(), None, save.data_ptr(), dy.data_ptr(), tmp.data_ptr(),
                          gx.data_ptr() if gx is not None else None, None, K.stream_workspace(x.device)[0], None,
                          zb.data_ptr() if zb is not None else None))
            tmps.append((tmp, dy, zb))
            gxs.append(gx)
        program.run_group(progs, True, slots)
        grads = [None]
        sink = _grad_sink
        for i, prog in enumerate(progs):
            for p in prog.touched:
                sink.touched(p)
            gc = None
            if prog.need_coef:
                c = saved[3 * i + 1]
                gc = _own(tmps[i][2][prog.gcoef_off // 4:prog.gcoef_off // 4 + c.numel()].reshape(c.shape), ctx.coef_leaf[i])
            grads += [gxs[i], gc]
        return tuple(grads)


def mixed_op_program_group(xs, coefs, progs, dests=None):
    """Outputs of k MixedOp programs executed together by fs_exec_program_group (1 <= k <= program.MAX_GROUP); dests[i]: where output i
    is written (a dense NHWC tensor of its shape, e.g. a PairBuffers half), or None."""
    flat = []
    for x, c in zip(xs, coefs):
        flat += [x, c]
    return _MixedOpProgramGroup.apply((tuple(progs), list(dests) if dests is not None else None), *flat)


def weighted_sum(xs, coef):
    """sum_k coef[k] * xs[k] for NHWC feature maps and a device-resident coefficient vector (len(xs) <= 8)."""
    xs = [as_nhwc(t) for t in xs]
    if any(t.dtype != xs[0].dtype for t in xs):
        xs = [as_nhwc(t, xs[0].dtype) for t in xs]
    if coef.device != xs[0].device:
        coef = coef.to(xs[0].device)
    assert coef.numel() == len(xs) <= 8, "weighted_sum: %d coefficients for %d operands" % (coef.numel(), len(xs))
    return _WeightedSum.apply(coef, *xs)


def cat(tensors):
    """torch.cat(dim=1) of NHWC feature maps via fs_copy_channels; autograd by slicing."""
    if _tracer is not None:
        return _tracer.cat(list(tensors))
    return _Cat.apply(*[as_nhwc(t) for t in tensors])


class _Cat(torch.autograd.Function):
    @staticmethod
    def forward(ctx, *tensors):
        ctx.sizes = [t.shape[1] for t in tensors]
        return K.cat_channels(list(tensors))

    @staticmethod
    def backward(ctx, dy):
        dy = as_nhwc(dy)
        outs, off = [], 0
        for c in ctx.sizes:
            outs.append(dy[:, off:off + c])
            off += c
        return tuple(outs)
