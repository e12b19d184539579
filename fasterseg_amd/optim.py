"""Optimizer step over the flat gradient buffer: `clip_grad_norm_` + `torch.optim.SGD(momentum, weight_decay).step()` for
every parameter in ONE kernel launch (fs_sgd_momentum_multi).

The supernet has ~40 k parameter tensors (252 M floats); torch's foreach SGD plus the clip cost ~105 ms of host and
launch time per step, a fifth of the whole pretrain step, for what is a single streaming pass over 3 GB.  The gradients
already sit in `FlatGradientSync.flat`; the momentum buffer mirrors it, and a static device-side table tells the kernel
where each parameter lives.  Semantics are those of the reference's optimiser calls (search/train_search.py:94-98,248-250;
train/train.py:173-176): global L2 norm over all gradients, scale = min(1, max_norm / (norm + 1e-6)), then
buf = momentum*buf + (g + wd*p), p -= lr*buf; a parameter that received no gradient in a step is skipped entirely (torch
skips grad=None parameters: no decay, no momentum update).
"""
import numpy as np
import torch

from . import _lib
from . import functional as FN
from . import kernels as K

_TENSOR = np.dtype([("p", "<u8"), ("g_off", "<i8"), ("numel", "<i8"), ("I", "<i4"), ("taps", "<i4"), ("pack_fwd", "<u8"),
                    ("pack_flip", "<u8")])     # fs_sgd_tensor


class FlatSGD:
    def __init__(self, sync, lr, momentum=0.9, weight_decay=0.0, max_norm=None, pack_dtype=None, unused="skip"):
        """pack_dtype (torch.float32 / torch.bfloat16): also keep RESIDENT packed copies of every conv filter in that dtype -
        the [O][R][S][I] pack the forward conv reads and the rotated [I][R][S][O] pack of the data-gradient conv - rewritten
        by the update kernel itself, so train-mode convs read (slices of) them in place instead of re-packing a filter on
        every forward and backward (two fs_pack_weight launches per conv per pass: ~10 % of a supernet step's GPU time)."""
        self.sync = sync
        self.lr, self.momentum, self.weight_decay, self.max_norm = lr, momentum, weight_decay, max_norm
        # unused: what happens to a parameter that receives no gradient in a step.  "skip" = torch >= 2.0 (zero_grad(set_to_none=True):
        # .grad is None, the optimizer passes over it).  "decay" = the reference's pinned torch 1.1 (search/train_search.py:244-250):
        # zero_grad() leaves a zero-filled .grad on every parameter that was touched ONCE, which therefore keeps decaying and coasting on
        # its momentum.  Pinned to a trajectory of the reference by tests/test_train_parity_gpu.py (fixture optimizer_trajectory.npz).
        assert unused in ("skip", "decay")
        self.unused = unused
        self._ever_touched = None
        params = sync.params
        dev = sync.flat.device
        if dev.type != "cuda":
            raise RuntimeError("FlatSGD runs on the HIP kernels only (no CPU path)")
        table = np.zeros(len(params), _TENSOR)
        chunks = []
        self.pack_dtype = pack_dtype
        packable = [i for i, p in enumerate(params) if pack_dtype is not None and p.dim() == 4 and p.shape[0] % 8 == 0
                    and p.shape[1] % 8 == 0 and p.shape[2] == p.shape[3] and p.shape[2] in (1, 3)]
        # packs follow the layout of the flat gradient buffer: filters whose gradient slices are adjacent (the fusable pairs of a
        # search MixedOp, fusion.flat_order) have adjacent packs too, so one two-segment descriptor addresses both
        packable.sort(key=lambda i: sync.offsets[i])
        total = sum(params[i].numel() for i in packable)
        self.pack_fwd = torch.empty(total, dtype=pack_dtype, device=dev) if packable else None
        self.pack_flip = torch.empty(total, dtype=pack_dtype, device=dev) if packable else None
        pack_off, esize = {}, (0 if pack_dtype is None else torch.empty(0, dtype=pack_dtype).element_size())
        off = 0
        for i in packable:
            pack_off[i] = off
            off += params[i].numel()
        for i, p in enumerate(params):
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise ValueError("FlatSGD needs contiguous fp32 parameters")
            I, taps = (p.shape[1], p.shape[2] * p.shape[3]) if p.dim() == 4 else (1, 1)
            fwd = flip = 0
            if i in pack_off:
                fwd = self.pack_fwd.data_ptr() + pack_off[i] * esize
                flip = self.pack_flip.data_ptr() + pack_off[i] * esize
            table[i] = (p.data_ptr(), sync.offsets[i], p.numel(), I, taps, fwd, flip)
            n = int(_lib.lib().fs_sgd_tensor_chunks(p.numel(), I, taps, int(fwd != 0)))      # blocks of the update kernel
            chunks.append(np.stack([np.full(n, i, np.int32), np.arange(n, dtype=np.int32)], 1))
        self._ptrs = [(i, params[i].data_ptr()) for i in range(0, len(params), 61)]
        self.table = torch.from_numpy(table.view(np.uint8).copy()).to(dev)
        self.chunks = torch.from_numpy(np.ascontiguousarray(np.concatenate(chunks))).to(dev)
        self.momentum_buf = torch.zeros_like(sync.flat)
        self.touched_dev = torch.ones(len(params), dtype=torch.uint8, device=dev)
        self._last_touched = None
        self.last_norm = None
        if packable:
            self._launch(None, pack_only=True)
            for i in packable:
                p = params[i]
                O, I, R, S = p.shape
                a = pack_off[i]
                FN.register_resident_pack(p, self.pack_fwd[a:a + p.numel()].view(O, R, S, I),
                                          self.pack_flip[a:a + p.numel()].view(I, R, S, O))

    def _launch(self, scale, pack_only=False):
        sync = self.sync
        K.call("fs_sgd_momentum_multi", K._stream(), self.table.data_ptr(), self.chunks.data_ptr(), self.chunks.shape[0],
               self.touched_dev.data_ptr(), sync.flat.data_ptr(), self.momentum_buf.data_ptr(),
               scale.data_ptr() if scale is not None else None, float(self.lr), float(self.momentum), float(self.weight_decay),
               K.dtype_code(self.pack_dtype) if self.pack_dtype is not None else 0, int(pack_only))

    def refresh_packs(self):
        """Rebuild the resident packs from the current parameters (after load_state_dict or any out-of-band update)."""
        if self.pack_fwd is not None:
            self._launch(None, pack_only=True)
            for p in self.sync.params:
                FN.revalidate_resident_pack(p)

    def step(self):
        """Call after `sync.sync()`.  Returns the global gradient norm (device scalar) when clipping, else None."""
        sync = self.sync
        for i, ptr in self._ptrs:          # parameters must not have been re-homed (.to(), .data = ...) since construction
            if sync.params[i].data_ptr() != ptr:
                raise RuntimeError("FlatSGD: parameter storage moved since the optimizer was built")
        scale = None
        self.last_norm = None
        gs = float(getattr(sync, "grad_scale", 1.0))          # average="defer": the buffer holds the SUM over ranks
        if self.max_norm is not None:
            self.last_norm = sync.flat.norm()                 # slices of untouched parameters are zero
            if gs != 1.0:
                self.last_norm = self.last_norm * gs          # norm of the average
            scale = torch.clamp(self.max_norm / (self.last_norm + 1e-6), max=1.0)
            if gs != 1.0:
                scale = scale * gs                            # ... and the kernel's gradient read applies clip and average at once
        elif gs != 1.0:
            scale = torch.full((), gs, dtype=torch.float32, device=sync.flat.device)
        touched = sync._touched
        if self.unused == "decay":
            ever = self._ever_touched or [False] * len(touched)
            touched = self._ever_touched = [a or b for a, b in zip(ever, touched)]
        if touched != self._last_touched:
            # pinned + non_blocking: a blocking copy from pageable memory synchronises the launch stream, i.e. the host would wait here for
            # the whole backward of the step instead of going on to issue the next one (the table changes every step under sampled widths)
            self.touched_dev.copy_(torch.tensor(touched, dtype=torch.uint8).pin_memory(), non_blocking=True)
            self._last_touched = list(touched)
        self._launch(scale)
        FN.bump_weights_epoch()            # parameters changed behind autograd's version counters: drop packed copies
        return self.last_norm

    def zero_grad(self, set_to_none=True):
        """FlatGradientSync.prepare() owns (and clears) the gradient buffer: nothing to zero here.  The .grad views are left in place -
        dropping them would need FlatGradientSync.invalidate() so that the next prepare() re-points every parameter."""
        return
