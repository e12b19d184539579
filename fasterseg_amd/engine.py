"""Static inference engine: lowers a built network (eval mode) to a flat plan of HIP kernel launches and replays it
from a hipGraph.

This is what replaces the reference's TensorRT path for the derived network (latency/run_latency.py:54-81,
tools/utils/darts_utils.py:113-177: ONNX export -> TensorRT engine -> timed execute loop): the network's own
forward() is traced once with shape-only tensors, BatchNorm is folded into per-channel scale/shift, filters are
packed once, every feature map gets a fixed NHWC buffer, torch.cat disappears (producers write straight into channel
slices of the consumer's buffer), and the resulting ~90 launches are captured into one hipGraph so a frame costs one
graph launch.  Nothing here runs on the CPU per pixel and there is no fallback: every op is a libfasterseg_hip call.

The plan also carries the algorithmic FLOPs / bytes of every launch (SURVEY.md §8d conventions) and can time each
launch with HIP events on the launch stream (`profile()`), which is what bench.py's roofline block is computed from.
"""
import ctypes
import os

import torch

from . import _lib
from . import functional as FN
from . import kernels as K
from ._lib import FS_CONV_RELU, ConvDesc, ResizeDesc, call


class _Tracer:
    def __init__(self, dtype):
        self.dtype = dtype
        self.ops = []

    def _emit(self, kind, out_shape, nchw=False, **kw):
        out = FN.SymTensor(out_shape, self.dtype, nchw)
        out.producer = len(self.ops)
        self.ops.append(dict(kind=kind, out=out, **kw))
        return out

    @staticmethod
    def _conv_hw(H, W, k, stride, pad):
        return (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1

    def stem(self, x, weight, bn, relu, training):
        assert not training, "the engine lowers eval-mode networks only"
        N, C, H, W = x.shape
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        return self._emit("stem", (N, weight.shape[0], Ho, Wo), x=x, weight=weight, bn=bn, relu=relu)

    def conv(self, x, weight, bn, bias, stride, pad, relu, training, cout, cin):
        assert not training, "the engine lowers eval-mode networks only"
        N, C, H, W = x.shape
        assert C == cin, (C, cin)
        k = weight.shape[2]
        Ho, Wo = self._conv_hw(H, W, k, stride, pad)
        return self._emit("conv", (N, cout, Ho, Wo), x=x, weight=weight, bn=bn, bias=bias, stride=stride, pad=pad, relu=relu,
                          cout=cout, cin=cin, k=k)

    def factorized_reduce(self, x, w1, w2, bn, training, half, cin):
        assert not training
        N, C, H, W = x.shape
        return self._emit("fr", (N, 2 * half, H // 2, W // 2), x=x, w1=w1, w2=w2, bn=bn, half=half, cin=cin)

    def resize(self, x, size, relu, out_nchw):
        N, C, H, W = x.shape
        return self._emit("resize", (N, C, size[0], size[1]), nchw=bool(out_nchw), x=x, relu=relu, out_nchw=out_nchw)

    def cat(self, tensors):
        N, _, H, W = tensors[0].shape
        return self._emit("cat", (N, sum(t.shape[1] for t in tensors), H, W), inputs=list(tensors))


class InferenceEngine:
    def __init__(self, net, input_shape, dtype=torch.bfloat16, logits_dtype=torch.float32, device="cuda", use_graph=True,
                 halo_min_pixels=16384, lanes=3, fuse_cells=None, output="logits"):
        assert not net.training, "call net.eval() first"
        self.dtype = dtype
        self.device = torch.device(device)
        self.input_shape = tuple(input_shape)
        self.logits_dtype = logits_dtype
        # "logits": contiguous NCHW logits at input resolution (what Network_Multi_Path_Infer.forward returns, model_seg.py:365);
        # "classes": uint8 (N, H, W) class map - the final up-sample and the evaluator's arg-max (evaluator.py:223) in one
        # launch, so a validation frame writes 2 MB instead of 159 MB
        # "lowres": the head's logits before the final up-sample, as an NHWC view (N, C, h, w) - for consumers that evaluate
        # the bilinear up-sample themselves (losses.distill_kl_lowres reads the teacher this way)
        assert output in ("logits", "classes", "lowres")
        self.output_mode = output
        self.vec = K.vec_of(dtype)
        # 3x3/s1 layers with at least this many output pixels run on the LDS-halo kernel (conv3x3_halo.hip)
        self.halo_min_pixels = int(os.environ.get("FS_HALO_MIN_PIXELS", halo_min_pixels))
        # per-layer kernel choice for 3x3 / stride-1 convs (LDS-halo with a 32 / 64 / 128-channel tile, or the implicit GEMM) by
        # timing every candidate at build
        self.autotune = bool(int(os.environ.get("FS_ENGINE_AUTOTUNE", "1")))
        self.autotuned = []
        self.n_lanes = max(1, int(os.environ.get("FS_ENGINE_LANES", lanes)))
        self.input = torch.zeros(self.input_shape, dtype=torch.float32, device=self.device)
        self._keep = []            # tensors referenced by raw pointers in the plan
        # whole zoomed-conv cells (resize -> conv -> conv -> resize) as ONE launch (zoom_cell.hip): 0 off, 1 always when
        # the geometry is supported, "auto" (default) = time the fused launch against the separate launches per cell at build
        self.fuse_cells = os.environ.get("FS_ENGINE_FUSE_CELLS", "auto" if fuse_cells is None else str(fuse_cells))
        self.cell_log = []
        # FS_ENGINE_PLAN=<file>: tune once, reuse - the per-layer kernel choices and the per-cell fused / separate choices are
        # written to <file>.<output> after a build that timed them and read back by later builds instead of timing (a plan
        # tuned on an idle device can be replayed under a profiler, whose counters perturb the timings it would otherwise tune on)
        self._plan_path = os.environ.get("FS_ENGINE_PLAN")
        self._plan_in, self._plan_out = None, {"convs": [], "cells": []}
        self._trace(net)
        if self._plan_path:
            # a plan is replayed by position, so it is only valid for the network and input it was tuned on: its file name and
            # its "signature" field carry a hash of (input shape, traced op kinds and output shapes); a mismatch re-tunes
            import hashlib
            import json
            sig = hashlib.sha1(repr((tuple(self.input_shape), [(op["kind"], tuple(op["out"].shape)) for op in self.ops])).encode()).hexdigest()[:12]
            self._plan_out["signature"] = sig
            self._plan_path = "%s.%s.%s.%s" % (self._plan_path, output, "bf16" if dtype == torch.bfloat16 else "fp32", sig)
            if os.path.exists(self._plan_path):
                with open(self._plan_path) as f:
                    plan = json.load(f)
                if plan.get("signature") == sig:
                    self._plan_in = plan
        self._fuse_cells()
        self._fuse_resizes()
        self._assign_buffers()
        self._lower()
        self.graph = None
        if use_graph:
            if self.fuse_cells == "auto" and self._plan_in is None:
                self._warm()
                self._tune_cells()
            self._capture()
        if self._plan_path and self._plan_in is None:
            import json
            self._plan_out["cells"] = [g["choice"] for g in self.groups if len(g["variants"]) > 1]
            with open(self._plan_path, "w") as f:
                json.dump(self._plan_out, f)

    # ---- 1. trace ----------------------------------------------------------------------------------
    def _trace(self, net):
        tracer = _Tracer(self.dtype)
        x = FN.SymTensor(self.input_shape, torch.float32, nchw=True)
        x.storage = ("input", 0)
        FN._tracer = tracer
        try:
            with torch.no_grad():
                out = net(x)
        finally:
            FN._tracer = None
        assert isinstance(out, FN.SymTensor) and out.nchw, "network must end in the NCHW logits up-sample"
        self.ops = tracer.ops
        self.out_sym = out
        if self.output_mode == "lowres":
            last = self.ops[out.producer]
            assert last["kind"] == "resize" and last["out_nchw"]
            last["dead"] = True
            self.out_sym = last["x"]

    # ---- 1a. whole zoomed-conv cells -> one fused launch ---------------------------------------------------------------
    def _consumers(self):
        consumers = {}
        for idx, op in enumerate(self.ops):
            if op.get("dead"):
                continue
            for s_ in (op["inputs"] if op["kind"] == "cat" else [op["x"]]):
                consumers.setdefault(s_.id, []).append(idx)
        return consumers

    def _fuse_cells(self):
        """BasicResidual_downup_2x (operations.py:435-446) traces as resize(1/2) -> conv3x3+BN+ReLU -> conv3x3+BN ->
        resize(x2)+ReLU (or, at stride 2, conv3x3+BN+ReLU with no up-sample); BasicResidual2x at stride 1 (:352-359) as
        the two convs alone.  Each such chain whose intermediates have no other reader becomes one "zoom" op (zoom_cell.hip)."""
        self.fused_cells = 0
        if self.fuse_cells in ("0", "off"):
            return
        plain_too = bool(int(os.environ.get("FS_ENGINE_FUSE_PLAIN_2X", "1")))
        consumers = self._consumers()
        by_out = {op["out"].id: i for i, op in enumerate(self.ops)}

        def is_c3(op):
            return (op["kind"] == "conv" and op["k"] == 3 and op["stride"] == 1 and op["pad"] == 1 and op["bn"] is not None
                    and op["bias"] is None and not op.get("dead"))
        for ia, A in enumerate(self.ops):
            if not is_c3(A) or not A["relu"] or A["out"] is self.out_sym:
                continue
            cons = consumers.get(A["out"].id, [])
            if len(cons) != 1 or not is_c3(self.ops[cons[0]]):
                continue
            ib = cons[0]
            B = self.ops[ib]
            if B["cin"] != A["cout"] or B["cout"] != A["cout"]:
                continue
            # input side: a 1/2 resample feeding A?
            src, down, r1 = A["x"], False, None
            ir = by_out.get(A["x"].id)
            if ir is not None and self.ops[ir]["kind"] == "resize":
                R1 = self.ops[ir]
                Hs, Ws = R1["x"].shape[2], R1["x"].shape[3]
                if (not R1["relu"] and not R1["out_nchw"] and not R1.get("dead")
                        and tuple(R1["out"].shape[2:]) == (Hs // 2, Ws // 2) and Hs % 2 == 0 and Ws % 2 == 0):
                    src, down, r1 = R1["x"], True, ir
            # output side: ReLU in conv2 (no up-sample) or resize(x2)+ReLU as its only reader
            last, up = ib, False
            if not B["relu"]:
                cb = consumers.get(B["out"].id, [])
                if len(cb) != 1 or B["out"] is self.out_sym:
                    continue
                R2 = self.ops[cb[0]]
                h, w = B["out"].shape[2], B["out"].shape[3]
                if not (R2["kind"] == "resize" and R2["relu"] and not R2["out_nchw"] and tuple(R2["out"].shape[2:]) == (2 * h, 2 * w)):
                    continue
                last, up = cb[0], True
            if up and not down:
                continue
            if not down and not up and not plain_too:
                continue
            N, _, H, W = src.shape
            d = K.zoom_desc((N, A["cin"], H, W), K.round_up(A["cin"], self.vec), A["cout"], B["cout"], down, up,
                            K.round_up(B["cout"], self.vec), self.dtype)
            if not K.zoom_cell_supported(d):
                continue
            out = self.ops[last]["out"]
            chain = [dict(self.ops[i]) for i in ([r1] if down else []) + [ia, ib] + ([last] if up else [])]
            zoom = dict(kind="zoom", out=out, x=src, A=A, B=B, down=down, up=up, chain=chain)
            A["dead"] = B["dead"] = True
            if up:
                self.ops[last]["dead"] = True
            self.ops[last] = zoom                      # out.producer keeps pointing at this index
            if down and consumers.get(self.ops[r1]["out"].id, []) == [ia]:
                self.ops[r1]["dead"] = True
            self.fused_cells += 1
            consumers = self._consumers()
            by_out = {op["out"].id: i for i, op in enumerate(self.ops)}

    # ---- 1b. fold bilinear resamples into the gather of the conv that consumes them -----------------------------
    def _fuse_resizes(self):
        """A zoomed conv is resize(1/2) -> conv [-> conv] -> resize(x2)+ReLU (operations.py:203-277,362-446) and the x2 map is
        usually read by exactly one conv of the next cell.  On the maps where that happens every launch is latency-sized, so a
        resize whose only consumer is an implicit-GEMM conv is not materialised: the conv interpolates its input on the fly
        (fs_conv_desc.vr_*).  Measured: this only pays for 1x1 consumers - an implicit-GEMM 3x3 conv re-gathers (and would
        re-interpolate) every input pixel for each of its 9 taps and each 32-wide output tile, which made the 14 fusable 3x3
        convs of the student 6.7 us slower each while saving 3 us per resize (2197 -> 1929 fps) - so 3x3 consumers keep the
        materialised resize unless FS_ENGINE_FUSE_RESIZE=2.  Resizes feeding a concat, several consumers or the logits stay
        launches too."""
        self.fused_resizes = 0
        self.shared_resizes = 0
        # identical resamples of one feature map (two cells zooming the same input) are computed once
        seen = {}
        for rop in self.ops:
            if rop["kind"] != "resize" or rop["out_nchw"] or rop.get("dead"):
                continue
            key = (rop["x"].id, tuple(rop["out"].shape), bool(rop["relu"]))
            first = seen.setdefault(key, rop)
            if first is rop:
                continue
            dup, keep = rop["out"], first["out"]
            for op in self.ops:
                if op["kind"] == "cat":
                    op["inputs"] = [keep if s_ is dup else s_ for s_ in op["inputs"]]
                elif op.get("x") is dup:
                    op["x"] = keep
            if dup is not self.out_sym:
                rop["dead"] = True
                self.shared_resizes += 1
        mode = int(os.environ.get("FS_ENGINE_FUSE_RESIZE", "1"))          # 0 off, 1 into 1x1 convs, 2 into any implicit-GEMM conv
        if not mode:
            return
        consumers = {}
        for idx, op in enumerate(self.ops):
            if op.get("dead"):
                continue
            for s_ in (op["inputs"] if op["kind"] == "cat" else [op["x"]]):
                consumers.setdefault(s_.id, []).append(idx)
        for rop in self.ops:
            if rop["kind"] != "resize" or rop["out_nchw"] or rop["out"] is self.out_sym or rop.get("dead"):
                continue
            cons = consumers.get(rop["out"].id, [])
            if len(cons) != 1:
                continue
            cop = self.ops[cons[0]]
            if cop["kind"] != "conv" or cop.get("vres") is not None or (cop["k"] != 1 and mode < 2):
                continue
            N, C, H, W = rop["out"].shape
            if cop["k"] == 3 and cop["stride"] == 1 and cop["pad"] == 1 and N * H * W >= self.halo_min_pixels:
                continue                                     # that conv runs on the halo kernel
            cop["vres"] = (H, W, bool(rop["relu"]))
            cop["x"] = rop["x"]
            rop["dead"] = True
            self.fused_resizes += 1

    # ---- 2. buffers: one NHWC buffer per feature map; cat operands alias slices of the cat buffer ------
    def _new_buffer(self, N, H, W, cs, zero=False):
        make = torch.zeros if zero else torch.empty
        buf = make((N, H, W, cs), dtype=self.dtype, device=self.device)
        self._keep.append(buf)
        return buf

    def _assign_buffers(self):
        self.buffers = {}
        self.copies = {}           # cat op index -> [(src sym, channel offset)] that could not be aliased
        for idx, op in enumerate(self.ops):
            if op["kind"] != "cat":
                continue
            out = op["out"]
            N, C, H, W = out.shape
            bid = "cat%d" % idx
            self.buffers[bid] = self._new_buffer(N, H, W, K.round_up(C, self.vec))
            out.storage = (bid, 0)
            off = 0
            for s in op["inputs"]:
                can_alias = (s.storage is None and s.producer is not None and off % self.vec == 0
                             and self.ops[s.producer]["kind"] in ("conv", "resize", "stem", "fr", "zoom") and not s.nchw)
                if can_alias:
                    s.storage = (bid, off)
                else:
                    self.copies.setdefault(idx, []).append((s, off))
                off += s.shape[1]
        for idx, op in enumerate(self.ops):
            out = op["out"]
            if out.storage is not None or op.get("dead"):
                continue
            N, C, H, W = out.shape
            if out.nchw and self.output_mode == "classes":
                t = torch.empty((N, H, W), dtype=torch.uint8, device=self.device)
                self._keep.append(t)
                self.buffers["out%d" % idx] = t
                out.storage = ("out%d" % idx, 0)
                continue
            if out.nchw:
                t = torch.empty((N, C, H, W), dtype=self.logits_dtype, device=self.device)
                self._keep.append(t)
                self.buffers["out%d" % idx] = t
                out.storage = ("out%d" % idx, 0)
                continue
            pad_to = 32 if C % self.vec else self.vec       # classifier logits: 19 -> 32 channel stride
            bid = "buf%d" % idx
            self.buffers[bid] = self._new_buffer(N, H, W, K.round_up(C, pad_to), zero=(C % self.vec != 0))
            out.storage = (bid, 0)
        self.output = self.buffers[self.out_sym.storage[0]]
        if self.output_mode == "lowres":
            self.output = self.output.permute(0, 3, 1, 2)[:, :self.out_sym.shape[1]]

    def _ptr(self, sym):
        bid, off = sym.storage
        if bid == "input":
            return self.input.data_ptr(), 0
        buf = self.buffers[bid]
        if sym.nchw:
            return buf.data_ptr(), 0
        return buf.data_ptr() + off * buf.element_size(), buf.shape[3]

    # ---- 3. lower to ctypes call records --------------------------------------------------------------
    def _fold(self, bn, bias, cout, lo=0):
        if bn is not None:
            gamma, beta, rm, rv, eps = bn
            scale, shift = FN.fold_bn(gamma, beta, rm, rv, eps)
            scale, shift = scale[lo:lo + cout].contiguous().to(self.device), shift[lo:lo + cout].contiguous().to(self.device)
        else:
            scale = None
            shift = bias.detach().float()[:cout].contiguous().to(self.device) if bias is not None else None
        self._keep += [scale, shift]
        return scale, shift

    def _add_conv(self, x, out, weight, scale, shift, k, stride, pad, relu, cout, cin, out_off=0, label="conv", vres=None):
        N, _, H, W = x.shape
        src_hw = None
        if vres is not None:                      # x is the un-resampled source; the conv reads its (H, W) resampling
            src_hw, (H, W) = (H, W), vres[:2]
            label = "%s[<-%dx%d%s]" % (label, src_hw[0], src_hw[1], "+relu" if vres[2] else "")
        _, _, Ho, Wo = out.shape
        halo_ok = k == 3 and stride in (1, 2) and pad == 1 and vres is None
        use_halo = halo_ok and stride == 1 and N * H * W >= self.halo_min_pixels
        xp, x_cs = self._ptr(x)
        yp, y_cs = self._ptr(out)
        yp += out_off * (2 if self.dtype == torch.bfloat16 else 4)
        d = ConvDesc(N, H, W, cin, cout, k, k, stride, pad, Ho, Wo, x_cs, y_cs, K.dtype_code(self.dtype), FS_CONV_RELU if relu else 0)
        if vres is not None:
            d.vr_H, d.vr_W, d.vr_relu = src_hw[0], src_hw[1], int(vres[2])
        self._keep.append(d)

        def variant(halo, tile=0):
            dd = d
            if halo:        # LDS-halo 3x3 kernel with the fragment-packed filter bank; tile = forced output-channel tile (0: heuristic)
                wp = K.pack_weight_frag(weight.detach().to(self.device), self.dtype, cout, cin)
                if tile:
                    dd = ConvDesc.from_buffer_copy(bytes(d))
                    dd.flags |= {32: 0x1000, 64: 0x2000, 128: 0x3000}[tile]
                    self._keep.append(dd)
            else:
                wp = K.pack_weight(weight.detach().to(self.device), self.dtype, cout, cin)
            args = (ctypes.byref(dd), ctypes.c_void_p(xp), ctypes.c_void_p(wp.data_ptr()), K._p(scale), K._p(shift), ctypes.c_void_p(yp), None)
            return ("fs_conv3x3_s1_fwd" if halo else "fs_conv2d_fwd"), args, wp, dd
        fn, args, wp, dsel = variant(use_halo)
        # Which kernel (LDS-halo with a 32 / 64 / 128-channel tile, or the implicit GEMM) is fastest for a 3x3 layer depends on
        # how many blocks each gives on this map: every candidate is timed on the device at build and the fastest kept
        # (FS_ENGINE_AUTOTUNE=0: the heuristics only).
        if halo_ok and self.autotune and N * Ho * Wo >= 512:
            cands = [("halo", variant(True)), ("igemm", variant(False))]
            cands += [("halo%d" % t, variant(True, t)) for t in (32, 64, 128) if t <= max(32, K.round_up(cout, 32)) * 2 and t <= 128]
            if self._plan_in is not None:
                want = self._plan_in["convs"][len(self._plan_out["convs"])]
                best = (0.0, want, dict(cands)[want])
                timed = []
            else:
                timed = [(self._time_call(v[0], v[1]), name, v) for name, v in cands]
                best = min(timed, key=lambda tv: tv[0])
            self._plan_out["convs"].append(best[1])
            self.autotuned.append((label, N * H * W, cin, cout, best[1], [(nm, round(t * 1e3, 2)) for t, nm, _ in timed]))
            fn, args, wp, dsel = best[2]
        self._keep.append(wp)
        es = 2 if self.dtype == torch.bfloat16 else 4
        flops = 2.0 * N * Ho * Wo * cout * cin * k * k
        in_px = N * (src_hw[0] * src_hw[1] if src_hw else H * W)
        nbytes = es * (in_px * cin + cout * cin * k * k + N * Ho * Wo * cout)
        self.calls.append(dict(fn=fn, args=args, desc=dsel, family="conv%dx%d" % (k, k), flops=flops, bytes=nbytes,
                               label="%s %dx%d s%d %d->%d @%dx%d" % (label, k, k, stride, cin, cout, H, W)))

    # ---- 3a. a zoomed-conv cell: one fused launch, or (when that measures slower) its separate launches -----------------
    def _tmp_sym(self, shape):
        N, C, H, W = shape
        sym = FN.SymTensor(shape, self.dtype)
        bid = "tmp%d" % sym.id
        self.buffers[bid] = self._new_buffer(N, H, W, K.round_up(C, self.vec))
        sym.storage = (bid, 0)
        return sym

    def _resize_call(self, x, out, relu):
        es = 2 if self.dtype == torch.bfloat16 else 4
        N, C, Hi, Wi = x.shape
        Ho, Wo = out.shape[2], out.shape[3]
        xp, x_cs = self._ptr(x)
        yp, y_cs = self._ptr(out)
        d = ResizeDesc(N, Hi, Wi, Ho, Wo, C, x_cs, y_cs, K.dtype_code(self.dtype), int(relu), 0)
        self._keep.append(d)
        return dict(fn="fs_bilinear_fwd", args=(ctypes.byref(d), ctypes.c_void_p(xp), ctypes.c_void_p(yp)), desc=d, family="resize",
                    flops=0.0, bytes=es * N * C * (Hi * Wi + Ho * Wo), label="resize %dx%d->%dx%d C%d" % (Hi, Wi, Ho, Wo, C))

    def _time_calls(self, calls, reps=20):
        """Device time (ms) of a dependent sequence of launches, replayed `reps` times from one hipGraph."""
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            st = ctypes.c_void_p(side.cuda_stream)
            for c in calls:
                call(c["fn"], st, *c["args"])
            side.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
                for _ in range(reps):
                    for c in calls:
                        call(c["fn"], st, *c["args"])
            g.replay()
            best = None
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                g.replay()
                e1.record()
                e1.synchronize()
                t = e0.elapsed_time(e1) / reps
                best = t if best is None or t < best else best
        torch.cuda.current_stream().wait_stream(side)
        return best

    def _add_zoom(self, op, out):
        A, B, x, down, up = op["A"], op["B"], op["x"], op["down"], op["up"]
        es = 2 if self.dtype == torch.bfloat16 else 4
        N, _, H, W = x.shape
        cin, cmid, cout = A["cin"], A["cout"], B["cout"]
        xp, x_cs = self._ptr(x)
        yp, y_cs = self._ptr(out)
        d = K.zoom_desc((N, cin, H, W), x_cs, cmid, cout, down, up, y_cs, self.dtype)
        sc1, sh1 = self._fold(A["bn"], None, cmid)
        sc2, sh2 = self._fold(B["bn"], None, cout)
        w1 = K.pack_weight_frag(A["weight"].detach().to(self.device), self.dtype, cmid, cin)
        w2 = K.pack_weight_frag(B["weight"].detach().to(self.device), self.dtype, cout, cmid)
        self._keep += [d, w1, w2]
        args = (ctypes.byref(d), ctypes.c_void_p(xp), ctypes.c_void_p(w1.data_ptr()), K._p(sc1), K._p(sh1), ctypes.c_void_p(w2.data_ptr()),
                K._p(sc2), K._p(sh2), ctypes.c_void_p(yp))
        flops = 2.0 * N * d.h * d.w * 9 * (cin * cmid + cmid * cout)
        nbytes = es * (N * H * W * cin + 9 * (cin * cmid + cmid * cout) + N * d.Ho * d.Wo * cout)
        fused = dict(fn="fs_zoom_cell_fwd", args=args, desc=d, family="zoomcell", flops=flops, bytes=nbytes,
                     label="cell[%s%s] %d->%d->%d conv@%dx%d" % ("dn " if down else "", "up" if up else "", cin, cmid, cout, d.h, d.w))
        if self.fuse_cells != "auto":
            self.calls.append(fused)
            self.cell_log.append([fused["label"], "fused", None, None])
            return None
        # the same cell as separate launches through temporaries, timed against the fused launch
        first = len(self.calls)
        cur = x
        if down:
            t0 = self._tmp_sym((N, cin, d.h, d.w))
            self.calls.append(self._resize_call(cur, t0, False))
            cur = t0
        t1 = self._tmp_sym((N, cmid, d.h, d.w))
        self._add_conv(cur, t1, A["weight"], sc1, sh1, 3, 1, 1, True, cmid, cin)
        t2 = self._tmp_sym((N, cout, d.h, d.w)) if up else out
        self._add_conv(t1, t2, B["weight"], sc2, sh2, 3, 1, 1, not up, cout, cmid)
        if up:
            self.calls.append(self._resize_call(t2, out, True))
        chain = self.calls[first:]
        del self.calls[first:]
        # prior from isolated (cache-warm) timings; _tune_cells() then decides on whole-frame time, where the filter banks and
        # the input are cold as they are in production
        if self._plan_in is not None:
            t_fused = t_chain = 0.0
            keep_fused = self._plan_in["cells"][len(self.cell_log)] == 0
        else:
            t_fused, t_chain = self._time_calls([fused]), self._time_calls(chain)
            keep_fused = t_fused <= t_chain
        self.cell_log.append([fused["label"], "fused" if keep_fused else "split", round(t_fused * 1e3, 2), round(t_chain * 1e3, 2)])
        self.calls += [fused] if keep_fused else chain
        return dict(variants=[[fused], chain], choice=0 if keep_fused else 1, log=self.cell_log[-1])

    def _time_call(self, fn, args, reps=20):
        """Device time (ms) of one launch, replayed back-to-back from a small hipGraph (same method as profile())."""
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            st = ctypes.c_void_p(side.cuda_stream)
            call(fn, st, *args)
            side.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
                for _ in range(reps):
                    call(fn, st, *args)
            g.replay()
            best = None
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                g.replay()
                e1.record()
                e1.synchronize()
                t = e0.elapsed_time(e1) / reps
                best = t if best is None or t < best else best
        torch.cuda.current_stream().wait_stream(side)
        return best

    def _ready(self, sym):
        """call indices that must have completed before `sym` is fully written"""
        return self._sym_ready.get(sym.id, [])

    def _lower(self):
        """Ops -> call groups (a group = the launches of one op; a cell has two alternative groups), then _flatten()."""
        self.calls = []
        self.groups = []
        es = 2 if self.dtype == torch.bfloat16 else 4
        for idx, op in enumerate(self.ops):
            if op.get("dead"):
                continue
            kind, out = op["kind"], op["out"]
            first_call = len(self.calls)
            in_syms = op["inputs"] if kind == "cat" else [op["x"]]
            alt = None
            if kind == "stem":
                N, _, H, W = op["x"].shape
                cout = out.shape[1]
                scale, shift = self._fold(op["bn"], None, cout)
                wp = K.pack_weight(op["weight"].detach().to(self.device), torch.float32)
                self._keep.append(wp)
                yp, y_cs = self._ptr(out)
                args = (N, H, W, cout, ctypes.c_void_p(self.input.data_ptr()), ctypes.c_void_p(wp.data_ptr()), K._p(scale), K._p(shift),
                        ctypes.c_void_p(yp), y_cs, K.dtype_code(self.dtype), int(op["relu"]))
                Ho, Wo = out.shape[2], out.shape[3]
                self.calls.append(dict(fn="fs_conv_stem_fwd", args=args, family="stem", flops=2.0 * N * Ho * Wo * cout * 27,
                                       bytes=4 * N * 3 * H * W + es * N * Ho * Wo * cout,
                                       label="stem 3x3 s2 3->%d @%dx%d" % (cout, H, W)))
            elif kind == "conv":
                scale, shift = self._fold(op["bn"], op["bias"], op["cout"])
                self._add_conv(op["x"], out, op["weight"], scale, shift, op["k"], op["stride"], op["pad"], op["relu"], op["cout"],
                               op["cin"], vres=op.get("vres"))
            elif kind == "zoom":
                alt = self._add_zoom(op, out)
            elif kind == "fr":
                half = op["half"]
                for j, (w, pad) in enumerate(((op["w1"], 0), (op["w2"], -1))):
                    scale, shift = self._fold(op["bn"], None, half, lo=j * half)
                    self._add_conv(op["x"], out, w, scale, shift, 1, 2, pad, True, half, op["cin"], out_off=j * half, label="fr")
            elif kind == "resize":
                x = op["x"]
                N, C, Hi, Wi = x.shape
                Ho, Wo = out.shape[2], out.shape[3]
                xp, x_cs = self._ptr(x)
                yp, y_cs = self._ptr(out)
                onchw = 0
                out_es = es
                if op["out_nchw"]:
                    onchw = 1 if self.logits_dtype == torch.float32 else 2
                    out_es = 4 if onchw == 1 else es
                d = ResizeDesc(N, Hi, Wi, Ho, Wo, C, x_cs, y_cs, K.dtype_code(self.dtype), int(op["relu"]), onchw)
                self._keep.append(d)
                args = (ctypes.byref(d), ctypes.c_void_p(xp), ctypes.c_void_p(yp))
                if op["out_nchw"] and self.output_mode == "classes":
                    self.calls.append(dict(fn="fs_bilinear_argmax", args=args, desc=d, family="resize_argmax", flops=0.0,
                                           bytes=es * N * Hi * Wi * C + N * Ho * Wo,
                                           label="resize %dx%d->%dx%d C%d + argmax -> uint8" % (Hi, Wi, Ho, Wo, C)))
                else:
                    self.calls.append(dict(fn="fs_bilinear_fwd", args=args, desc=d, family="resize_nchw" if onchw else "resize",
                                           flops=0.0, bytes=es * N * Hi * Wi * C + out_es * N * Ho * Wo * C,
                                           label="resize %dx%d->%dx%d C%d%s" % (Hi, Wi, Ho, Wo, C, " nchw" if onchw else "")))
            elif kind == "cat":
                for s, off in self.copies.get(idx, []):
                    N, C, H, W = s.shape
                    sp, s_cs = self._ptr(s)
                    yp, y_cs = self._ptr(out)
                    args = (N * H * W, C, ctypes.c_void_p(sp), s_cs, ctypes.c_void_p(yp + off * es), y_cs, K.dtype_code(self.dtype))
                    self.calls.append(dict(fn="fs_copy_channels", args=args, family="copy", flops=0.0, bytes=2 * es * N * H * W * C,
                                           label="copy C%d @%dx%d" % (C, H, W)))
            else:
                raise RuntimeError("unknown op kind " + kind)
            calls = self.calls[first_call:]
            del self.calls[first_call:]
            group = dict(kind=kind, in_syms=in_syms, out=out, variants=[calls], choice=0, log=None)
            if alt is not None:
                group.update(alt)
            self.groups.append(group)
        self._flatten()

    def _flatten(self):
        """The launch list of the current variant choices, with dependencies, stream lanes and totals."""
        self.calls = []
        self._sym_ready = {}
        for g in self.groups:
            deps = sorted({d for s_ in g["in_syms"] for d in self._ready(s_)})
            first = len(self.calls)
            chain = g["kind"] == "zoom" and len(g["variants"][g["choice"]]) > 1     # un-fused cell: a dependent chain of launches
            for k, c in enumerate(g["variants"][g["choice"]]):
                c = dict(c)
                c["deps"] = [first + k - 1] if (chain and k > 0) else deps
                self.calls.append(c)
            new_calls = list(range(first, len(self.calls)))
            if chain:
                self._sym_ready[g["out"].id] = [new_calls[-1]]
            else:   # a cat output is ready when its aliased producers and its copy calls are; other outputs when their calls are
                self._sym_ready[g["out"].id] = (deps + new_calls) if g["kind"] == "cat" else new_calls
        self._assign_lanes()
        self.total_flops = sum(c["flops"] for c in self.calls)
        self.total_bytes = sum(c["bytes"] for c in self.calls)

    def _tune_cells(self):
        """Fused or separate launches, per cell, decided on the time of the WHOLE frame (single-stream hipGraph): timed alone,
        back to back, a launch finds its filter bank and input in L2, which flatters the fused kernel (it waits on memory in
        several dependent phases).  Greedy: flip one cell at a time, keep the flip if the frame gets faster."""
        cells = [g for g in self.groups if len(g["variants"]) > 1]
        if not cells:
            return

        def frame_ms():
            self._flatten()
            return self._time_graph(self._capture_once(1), reps=30)
        best = frame_ms()
        for g in cells:
            g["choice"] ^= 1
            t = frame_ms()
            if t < best * 0.997:
                best = t
            else:
                g["choice"] ^= 1
            g["log"][1] = "fused" if g["choice"] == 0 else "split"
            g["log"].append(round(t, 4))
        self._flatten()


    # ---- 3b. independent chains (the two branches of the derived network) go to separate HIP streams ----
    def _assign_lanes(self):
        """Greedy chain assignment: a call continues the lane of its latest dependency when that dependency is still the
        lane's tail, otherwise it takes the lane whose tail is oldest.  Cross-lane edges become event waits, so the
        captured hipGraph has parallel branches and the launch-latency-sized kernels of the 1/16-1/32 branch overlap
        with those of the 1/8-1/16 branch."""
        n_lanes = self.n_lanes
        tails = [-1] * n_lanes
        for i, c in enumerate(self.calls):
            deps = c["deps"]
            lane = None
            for d in sorted(deps, reverse=True):
                if tails[self.calls[d]["lane"]] == d:
                    lane = self.calls[d]["lane"]
                    break
            if lane is None:
                lane = 0 if not deps else min(range(n_lanes), key=lambda l: tails[l])
            c["lane"] = lane
            tails[lane] = i
        self._finish_lanes()

    def _finish_lanes(self):
        """Cross-lane edges (event waits / signals) and the per-lane split-K workspaces for the current lane assignment."""
        n_lanes = max(self.n_lanes, 1 + max(c["lane"] for c in self.calls))
        for c in self.calls:
            c["signal"] = False
        for i, c in enumerate(self.calls):
            c["waits"] = [d for d in c["deps"] if self.calls[d]["lane"] != c["lane"]]
            for d in c["waits"]:
                self.calls[d]["signal"] = True
        # one split-K scratch buffer per lane (calls of a lane are ordered, lanes run concurrently): the implicit-GEMM conv
        # splits long contractions of launch-latency-sized layers across blocks when it gets a workspace
        if bool(int(os.environ.get("FS_ENGINE_SPLITK", "1"))):
            ws = getattr(self, "_workspaces", None) or []
            while len(ws) < n_lanes:
                ws.append(torch.empty(K.WORKSPACE_BYTES, dtype=torch.uint8, device=self.device))
            self._workspaces = ws
            for c in self.calls:
                if c["fn"] == "fs_conv2d_fwd":
                    c["fn"] = "fs_conv2d_fwd_ws"
                    c["base_args"] = c["args"]
                if c["fn"] == "fs_conv2d_fwd_ws":
                    c["args"] = c["base_args"] + (ctypes.c_void_p(ws[c["lane"]].data_ptr()), K.WORKSPACE_BYTES)

    def _schedule(self, durs, n_lanes, edge_us=1.5):
        """List scheduling with measured launch durations (HEFT): calls are ordered by the length of the dependency path that
        still follows them and each goes to the lane where it can start first (an edge that crosses lanes costs an event,
        ~`edge_us`).  The trace-order greedy of `_assign_lanes` serialised independent chains behind each other: in the searched
        arch_1 frame the 1/16 branch only started after the five launch-latency-sized 1/32 cells of the other branch
        (profiles/r02_c2_infer_bf16_frame_timeline.csv).  Reorders self.calls (a topological order) and sets lanes."""
        n = len(self.calls)
        succ = [[] for _ in range(n)]
        for i, c in enumerate(self.calls):
            for d in c["deps"]:
                succ[d].append(i)
        rank = [0.0] * n
        for i in range(n - 1, -1, -1):                     # trace order is topological
            rank[i] = durs[i] + max([rank[j] for j in succ[i]] or [0.0])
        order = sorted(range(n), key=lambda i: (-rank[i], i))
        lane_free = [0.0] * n_lanes
        finish, lane_of = [0.0] * n, [0] * n
        for i in order:
            deps = self.calls[i]["deps"]
            best = None
            for l in range(n_lanes):
                start = lane_free[l]
                for d in deps:
                    start = max(start, finish[d] + (edge_us * 1e-3 if lane_of[d] != l else 0.0))
                if best is None or start < best[0] - 1e-9:
                    best = (start, l)
            lane_of[i] = best[1]
            finish[i] = best[0] + durs[i]
            lane_free[best[1]] = finish[i]
        remap = {old: new for new, old in enumerate(order)}
        calls = []
        for old in order:
            c = self.calls[old]
            c["deps"] = sorted(remap[d] for d in c["deps"])
            c["lane"] = lane_of[old]
            calls.append(c)
        self.calls = calls
        self._finish_lanes()
        return max(finish)

    # ---- 4. run ----------------------------------------------------------------------------------------
    def _launch_all(self):
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        for c in self.calls:
            call(c["fn"], st, *c["args"])

    def _launch_all_lanes(self, main):
        """Issue the plan on `main` + side streams with event edges (used under graph capture)."""
        streams = [main] + self._side_streams
        events = {}
        fork = torch.cuda.Event()
        fork.record(main)
        for s_ in streams[1:]:
            s_.wait_event(fork)
        for i, c in enumerate(self.calls):
            s_ = streams[c["lane"]]
            for d in c["waits"]:
                s_.wait_event(events[d])
            call(c["fn"], ctypes.c_void_p(s_.cuda_stream), *c["args"])
            if c["signal"]:
                ev = torch.cuda.Event()
                ev.record(s_)
                events[i] = ev
        for s_ in streams[1:]:                      # join every lane back into the capturing stream
            ev = torch.cuda.Event()
            ev.record(s_)
            main.wait_event(ev)

    # ---- 4b. the plan as one multi-stream launch program (fs_exec_program_streams) ---------------------------------
    _OPS = {"fs_bilinear_argmax": "OP_BILINEAR_ARGMAX", "fs_zoom_cell_fwd": "OP_ZOOM_CELL", "fs_conv2d_fwd_ws": "OP_CONV_FWD", "fs_conv2d_fwd": "OP_CONV_FWD", "fs_conv3x3_s1_fwd": "OP_CONV3X3_S1",
            "fs_conv_stem_fwd": "OP_STEM", "fs_bilinear_fwd": "OP_BILINEAR_FWD", "fs_copy_channels": "OP_COPY_CHANNELS"}

    def _build_program(self):
        """Same launches, lanes and cross-lane edges as `_launch_all_lanes`, as a relocatable-free command list (all
        addresses absolute) that the C executor issues from one FFI call: lane 0 = the caller's stream."""
        from . import program as P
        lst = P._List()
        lib = _lib.lib()
        self._events = []

        def event():
            ev = lib.fs_event_create()
            if not ev:
                raise RuntimeError("fs_event_create failed")
            self._events.append(ev)
            return P.Ref(P.ABS, ev)

        def arg(a):
            if a is None:
                return P.NULL
            if isinstance(a, ctypes.c_void_p):
                return P.Ref(P.ABS, a.value or 0)
            return a
        n_lanes = self.n_lanes
        if n_lanes > 1:
            fork = event()
            lst.emit(P.OP_EVENT_RECORD, fork, lane=0)
            for lane in range(1, n_lanes):
                lst.emit(P.OP_EVENT_WAIT, fork, lane=lane)
        signals = {}
        for i, c in enumerate(self.calls):
            lane = c["lane"]
            for d in c["waits"]:
                lst.emit(P.OP_EVENT_WAIT, signals[d], lane=lane)
            args = [arg(a) for a in c["args"]]
            if "desc" in c:
                args[0] = P._Desc(c["desc"])
            if c["fn"] == "fs_conv2d_fwd":                      # no workspace assigned: whole contraction in one block
                args += [P.NULL, 0]
            lst.emit(getattr(P, self._OPS[c["fn"]]), *args, lane=lane)
            if c["signal"]:
                signals[i] = event()
                lst.emit(P.OP_EVENT_RECORD, signals[i], lane=lane)
        for lane in range(1, n_lanes):
            ev = event()
            lst.emit(P.OP_EVENT_RECORD, ev, lane=lane)
            lst.emit(P.OP_EVENT_WAIT, ev, lane=0)
        self._prog_words, self._prog_n, self._prog_blob, _ = lst.finish()
        self._prog_slots = (ctypes.c_void_p * P.N_SLOTS)()
        self._prog_side = [torch.cuda.Stream() for _ in range(n_lanes - 1)]

    def _run_program(self):
        streams = (ctypes.c_void_p * self.n_lanes)(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()),
                                                   *[s_.cuda_stream for s_ in self._prog_side])
        call("fs_exec_program_streams", streams, self.n_lanes, self._prog_words, self._prog_n, self._prog_blob, self._prog_slots,
             len(self._prog_slots))

    def __del__(self):
        import sys
        if sys.is_finalizing():        # interpreter shutdown: the HIP runtime may already be gone, leave the events to the process exit
            return
        try:
            lib = _lib.lib()
            for ev in getattr(self, "_events", []):
                lib.fs_event_destroy(ev)
        except Exception:
            pass

    def _capture_once(self, lanes):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        self._side_streams = [torch.cuda.Stream() for _ in range(lanes - 1)]
        with torch.cuda.stream(side):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                if lanes > 1:
                    self._launch_all_lanes(side)
                else:
                    self._launch_all()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        return g

    @staticmethod
    def _time_graph(g, reps=40):
        for _ in range(5):
            g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / reps

    def _warm(self):
        torch.cuda.synchronize()
        warm = torch.cuda.Stream()
        warm.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(warm):
            self._launch_all()                     # warm-up outside capture
        torch.cuda.current_stream().wait_stream(warm)
        torch.cuda.synchronize()

    def _capture(self):
        """Capture the plan and keep the fastest of a few instantiations.  How the runtime maps the forked lanes of a
        captured graph onto hardware queues is not under our control and is bimodal on MI355X (the same 3-lane graph replays
        in 0.49 ms or in 0.63 ms per frame, kernels identical): so the multi-lane capture is tried a few times on fresh
        streams, the single-stream graph (0.53 ms) is the fallback, and every candidate is timed on the device."""
        self._warm()
        tries = ([self.n_lanes] * 3 if self.n_lanes > 1 else []) + [1]
        self.capture_log = []
        best = None
        for lanes in tries:
            g = self._capture_once(lanes)
            ms = self._time_graph(g)
            self.capture_log.append((lanes, round(ms, 4)))
            if best is None or ms < best[0]:
                best = (ms, g, lanes, self._side_streams, None)
        # FS_ENGINE_HEFT=1: also try list-scheduled variants of the same launches (measured durations, 2..6 lanes), kept only if
        # the frame gets faster.  Off by default: on the searched arch_1 frame the scheduler's estimate is 0.3145 ms (the
        # dependency-critical path of the isolated launch times) but every variant replays in 0.378-0.381 ms, the same as the
        # trace-order greedy - in the frame the kernels run cold and share the CUs, the chain is not the limit.
        if self.n_lanes > 1 and bool(int(os.environ.get("FS_ENGINE_HEFT", "0"))):
            durs = [p_["ms"] for p_ in self.profile(repeats=10, rounds=1)]
            greedy = [(dict(c), list(c["deps"])) for c in self.calls]
            for lanes in (2, 3, 4, 6):
                self.calls = [dict(c, deps=list(d)) for c, d in greedy]
                est = self._schedule(durs, lanes)
                used = 1 + max(c["lane"] for c in self.calls)
                for _ in range(2):
                    g = self._capture_once(used)
                    ms = self._time_graph(g)
                    self.capture_log.append(("heft%d" % lanes, round(ms, 4), round(est, 4)))
                    if ms < best[0]:
                        best = (ms, g, used, self._side_streams, [dict(c) for c in self.calls])
            if best[4] is None:
                self.calls = [dict(c, deps=list(d)) for c, d in greedy]
                self._finish_lanes()
            else:
                self.calls = best[4]
                self.n_lanes = best[2]
        self.graph, self.graph_lanes, self._side_streams = best[1], best[2], best[3]
        # the same plan issued directly from the C executor (no hipGraph): cheaper on the host per launch
        self.use_program = False
        if bool(int(os.environ.get("FS_ENGINE_PROGRAM", "1"))):
            self._build_program()
            for _ in range(5):
                self._run_program()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(40):
                self._run_program()
            e1.record()
            e1.synchronize()
            ms = e0.elapsed_time(e1) / 40
            self.capture_log.append(("program", round(ms, 4)))
            self.use_program = ms < best[0]

    def run(self):
        """One forward on the current contents of `self.input`; result in `self.output` (contiguous NCHW logits)."""
        if getattr(self, "use_program", False):
            self._run_program()
        elif self.graph is not None:
            self.graph.replay()
        else:
            self._launch_all()
        return self.output

    def __call__(self, x):
        self.input.copy_(x)
        return self.run()

    def census_entries(self, timed_rows=None):
        """The plan's convolution launches as (family, descriptor, count, ms) tuples in fasterseg_amd.census form; ms from
        `timed_rows` = profile_in_frame() when given."""
        fam = {"fs_conv2d_fwd_ws": 0, "fs_conv2d_fwd": 0, "fs_conv3x3_s1_fwd": 1}
        return [(fam[c["fn"]], c["desc"], 1, timed_rows[i]["ms"] if timed_rows else 0.0) for i, c in enumerate(self.calls) if c["fn"] in fam]

    def profile_in_frame(self, frames=20, warm=3):
        """Device time of every launch of the plan IN THE FRAME: `frames` whole forwards are issued launch by launch on one
        stream (plan order) with the library's census at level 2, i.e. every kernel carries its own start/stop HIP event pair
        (hipExtLaunchKernelGGL: the dispatch's begin -> end interval, what rocprofv3's kernel trace reports).  Each launch sees
        the cache state the preceding launches of the frame left - not its own operands from 20 warm replays (profile()).
        Returns [dict(label, family, ms, flops, bytes, launches)] in plan order, ms = mean per frame."""
        from . import _lib
        lib = _lib.lib()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        n = len(self.calls)
        with torch.cuda.stream(side):
            st = ctypes.c_void_p(side.cuda_stream)
            for _ in range(warm):
                self._launch_all_on(st)
            side.synchronize()
            lib.fs_census_enable(2)
            try:
                for _ in range(frames):
                    for i, c in enumerate(self.calls):
                        lib.fs_census_tag(i)
                        call(c["fn"], st, *c["args"])
                lib.fs_census_tag(-1)
                side.synchronize()
            finally:
                lib.fs_census_enable(0)
            counts = (ctypes.c_longlong * n)()
            ms = (ctypes.c_double * n)()
            lib.fs_census_read_tags(n, ctypes.cast(counts, ctypes.c_void_p), ctypes.cast(ms, ctypes.c_void_p))
        torch.cuda.current_stream().wait_stream(side)
        return [dict(label=c["label"], family=c["family"], ms=ms[i] / frames, flops=c["flops"], bytes=c["bytes"],
                     launches=counts[i] / frames) for i, c in enumerate(self.calls)]

    def profile(self, repeats=20, rounds=3):
        """Device time of every launch of the plan, measured with HIP events on the launch stream.

        Each launch is captured `repeats` times back-to-back into its own small hipGraph and the replay is bracketed by
        two events (median of `rounds`), so the figure is kernel duration + one dependent-kernel boundary (~1.5 us) and
        is free of Python launch overhead; rocprofv3 --kernel-trace durations of the same kernels (profiles/) are the
        cross-check.  Returns [dict(label, family, ms, flops, bytes)] in plan order."""
        out = []
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            st = ctypes.c_void_p(side.cuda_stream)
            self._launch_all_on(st)
            side.synchronize()
            for c in self.calls:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=side):
                    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
                    for _ in range(repeats):
                        call(c["fn"], st, *c["args"])
                g.replay()
                times = []
                for _ in range(rounds):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    g.replay()
                    e1.record()
                    e1.synchronize()
                    times.append(e0.elapsed_time(e1) / repeats)
                times.sort()
                out.append(dict(label=c["label"], family=c["family"], ms=times[len(times) // 2], flops=c["flops"], bytes=c["bytes"]))
        torch.cuda.current_stream().wait_stream(side)
        return out

    def _launch_all_on(self, st):
        for c in self.calls:
            call(c["fn"], st, *c["args"])
