"""The train steps that surround the hot path, restated from the reference drivers so they can be benchmarked on
synthetic batches (the drivers themselves need Cityscapes + cv2 + tensorboardX and are out of scope):

  StudentDistillStep  train/train.py:219-271 — teacher (eval, no_grad) forward, student (train) forward with three heads,
                      OHEM-CE(pred8) + 0.2*OHEM-CE(pred16) + 0.2*OHEM-CE(pred32) + KLDiv(student || teacher), backward,
                      SGD(momentum .9, wd 5e-4).  Under DP the flat gradient buffer is all-reduced between backward and
                      the optimizer step (parallel.FlatGradientSync).
Every conv / BN / resize in both networks runs on the HIP kernels (operations.py -> functional.py); losses and the
optimizer are PyTorch ops.
"""
import torch

from . import archs
from .losses import ProbOhemCrossEntropy2d, distill_kl
from .optim import FlatSGD
from .parallel import FlatGradientSync, broadcast_parameters


class StudentDistillStep:
    def __init__(self, batch, height, width, lr=0.01, momentum=0.9, weight_decay=5e-4, teacher_engine_dtype=None, seed=12345,
                 device="cuda", compute_dtype=torch.float32, fused_loss=None):
        self.device = torch.device(device)
        self.compute_dtype = compute_dtype      # activation storage / MFMA operand type; master weights, BN statistics,
        # accumulators and gradients of parameters stay fp32
        self.teacher = archs.init_weight(archs.build_derived(0, training=True), seed).to(self.device).eval()
        self.student = archs.init_weight(archs.build_derived(1, training=True), seed + 1).to(self.device).train()
        broadcast_parameters(self.student)
        broadcast_parameters(self.teacher)
        min_kept = int(batch * height * width // 16)                       # train/train.py:62 with gt_down_sampling = 1
        self.ohem = ProbOhemCrossEntropy2d(ignore_label=255, thresh=0.7, min_kept=min_kept, use_weight=False)
        self.sync = FlatGradientSync(self.student.parameters(), bucket_mb=4, average="defer")      # 17.6 MB -> 5 buckets, overlapped with backward
        self.optimizer = FlatSGD(self.sync, lr, momentum, weight_decay, pack_dtype=compute_dtype)       # train/train.py:173-176
        self.lamb = 0.2
        # loss heads straight from the 1/8 - 1/32 resolution logits (loss_up.hip): the up-sampled (B, 19, H, W) tensors of
        # student and teacher are never materialised.  FS_FUSED_LOSS=0 keeps the up-sample + full-resolution criteria.
        import os
        self.fused_loss = bool(int(os.environ.get("FS_FUSED_LOSS", "1"))) if fused_loss is None else bool(fused_loss)
        self.size = (height, width)
        self.teacher_engine = None
        if teacher_engine_dtype is not None:        # frozen teacher through the static-plan engine (hipGraph)
            from .engine import InferenceEngine
            self.teacher_engine = InferenceEngine(self.teacher, (batch, 3, height, width), dtype=teacher_engine_dtype,
                                                  output="lowres" if self.fused_loss else "logits")

    def teacher_logits(self, imgs):
        with torch.no_grad():
            if self.teacher_engine is not None:
                return self.teacher_engine(imgs)
            if self.fused_loss:
                return self.teacher.forward_lowres(imgs)
            return self.teacher(imgs)

    def _loss(self, imgs, target, t_logits, ohem):
        from . import functional as FN
        from .losses import distill_kl_lowres, ohem_ce_lowres
        FN.set_compute_dtype(self.compute_dtype)
        try:
            p8, p16, p32 = self.student.forward_lowres(imgs) if self.fused_loss else self.student(imgs)
        finally:
            FN.set_compute_dtype(torch.float32)
        if self.fused_loss:      # train/train.py:254-260 with the x8 / x16 / x32 up-samples evaluated inside the criteria
            loss = (ohem_ce_lowres(ohem, p8, target) + self.lamb * ohem_ce_lowres(ohem, p16, target)
                    + self.lamb * ohem_ce_lowres(ohem, p32, target))
            return loss + distill_kl_lowres(p8, t_logits, self.size)
        loss = ohem(p8, target) + self.lamb * ohem(p16, target) + self.lamb * ohem(p32, target)
        return loss + distill_kl(p8, t_logits)

    def loss_only(self, imgs, target):
        """The step's loss on any number of images (train/train.py:246-262) through the same modules, criteria and compute dtype,
        without backward / update: the frozen teacher runs through its modules (the engine's plan is fixed to the step's batch), OHEM's
        min_kept follows the images given (train/train.py:62).  BatchNorm running statistics are put back afterwards.  Used by
        bench.py's parity gate against the CPU oracle."""
        from . import functional as FN
        buffers = list(self.student.buffers())
        saved = [b_.clone() for b_ in buffers]
        ohem = ProbOhemCrossEntropy2d(ignore_label=255, thresh=0.7, min_kept=int(imgs.shape[0] * self.size[0] * self.size[1] // 16),
                                      use_weight=False)
        with torch.no_grad():
            FN.set_compute_dtype(self.compute_dtype)
            try:
                t_logits = self.teacher.forward_lowres(imgs) if self.fused_loss else self.teacher(imgs)
            finally:
                FN.set_compute_dtype(torch.float32)
            loss = float(self._loss(imgs, target, t_logits, ohem))
            for b_, s_ in zip(buffers, saved):
                b_.copy_(s_)
        return loss

    def step(self, imgs, target):
        self.sync.prepare()
        t_logits = self.teacher_logits(imgs)
        loss = self._loss(imgs, target, t_logits, self.ohem)
        loss.backward()
        self.sync.sync()
        self.optimizer.step()
        return loss.detach()


import os as _os
_FAST_PHASE = bool(int(_os.environ.get("FS_FAST_PHASE", "1")))      # 0: flip requires_grad of all ~40 k weights at every phase change
# gradient all-reduce of the supernet step under the backward of its last (eager) pass: 1 with more than one rank (default), 2 also the
# bookkeeping on a single rank (tests), 0 every bucket in sync()
_DP_OVERLAP = int(_os.environ.get("FS_DP_OVERLAP", "1"))
# Adjacent passes of one `_loss` call evaluated together, layer by layer (model_search.Network_Multi_Path.forward_multi): the two
# fixed-width passes (max, min) become ONE captured graph, the two "random" passes of pretraining one eager joint pass - every kernel of
# a layer is then one grouped launch over both passes' problems.  FS_JOINT_PASSES=0: pass after pass (rounds 1-5).
_JOINT_PASSES = bool(int(_os.environ.get("FS_JOINT_PASSES", "1")))


class SearchConfig:
    """The fields of search/config_search.py the steps read (:57-59,78-107)."""
    lr = 2e-2
    momentum = 0.9
    weight_decay = 5e-4
    grad_clip = 5
    arch_learning_rate = 3e-4
    layers = 16
    Fch = 12
    width_mult_list = [4. / 12, 6. / 12, 8. / 12, 10. / 12, 1.]
    prun_modes = ['max', 'arch_ratio']
    stem_head_width = [(1, 1), (8. / 12, 8. / 12)]
    latency_weight = [0, 1e-2]


class SupernetStep:
    """search/train_search.py:215-251.  pretrain: `_loss(imgs, target, True)` (4 forwards: max, min, random, random) ->
    backward -> clip_grad_norm_(5) -> SGD.  search: Architect.step on the search batch (arch0, arch1 with Gumbel widths,
    max, min + latency penalty -> Adam on alpha/beta/ratio) followed by the weight step `_loss(imgs, target, False)`.
    Under DP the flat gradient buffer is all-reduced BEFORE the clip so the clip sees the global gradient; width sampling
    uses the host RNGs (np.random / torch CPU generator), seeded identically on every rank, so all ranks activate the
    same sub-network."""

    def __init__(self, pretrain=True, cfg=SearchConfig, seed=12345, device="cuda", lut=None, use_graphs=None,
                 compute_dtype=torch.float32, gc_freeze=None, bucket_mb=128):
        import os
        self.compute_dtype = compute_dtype
        self.use_graphs = bool(int(os.environ.get("FS_SUPERNET_GRAPHS", "1"))) if use_graphs is None else use_graphs
        self.graphs = None
        from . import model_search, operations
        from .architect import Architect
        import numpy as np
        self.pretrain = pretrain
        self.cfg = cfg
        crit = torch.nn.CrossEntropyLoss(ignore_index=255)                # train_search.py: ignore label 255
        self.model = model_search.Network_Multi_Path(19, cfg.layers, crit, cfg.Fch, cfg.width_mult_list, cfg.prun_modes,
                                                     cfg.stem_head_width)
        archs.init_weight(self.model, seed)
        self.model = self.model.to(device).train()
        broadcast_parameters(self.model)
        arch_ids = {id(p) for group in self.model._arch_parameters for p in group}
        self.weights = [p for p in self.model.parameters() if id(p) not in arch_ids]      # train_search.py:94-98
        # horizontal fusion inside every MixedOp ('conv' + 'conv_2x' first convs as one GEMM, the two zoomed primitives sharing their
        # down-sample and first conv): the pairs' BatchNorm state, gradient slices and resident packs become adjacent (fusion.py)
        from . import fusion
        self.fused_pairs = 0
        if bool(int(os.environ.get("FS_FUSE_MIXEDOP", "1"))):
            self.fused_pairs = fusion.colocate(self.model)
            self.weights = fusion.flat_order(self.model, self.weights)
        # The weight step's backward also reaches alpha/beta/ratio in the reference, but those gradients are zeroed by
        # the architect before it ever reads them (architect.py: optimizer.zero_grad() first); they are not computed here.
        self.arch_params = [p for group in self.model._arch_parameters for p in group]
        for p in self.arch_params:
            p.requires_grad_(False)
        self.sync = FlatGradientSync(self.weights, bucket_mb=bucket_mb, average="defer")      # FlatSGD folds 1 / world into its clip scale
        # train_search.py:94-98 SGD + :249 clip_grad_norm_(5), one launch over the flat buffers
        self.optimizer = FlatSGD(self.sync, cfg.lr, cfg.momentum, cfg.weight_decay, max_norm=cfg.grad_clip,
                                 pack_dtype=compute_dtype, unused=os.environ.get("FS_SGD_UNUSED", "skip"))
        # Width sampling (np.random.choice for "random", torch.rand for the Gumbel noise) runs on the host RNGs: every rank must
        # draw the SAME sub-network, or a parameter would be updated on one rank and skipped (grad=None) on another.  Rank 0's
        # seed wins, whatever the others were given.
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            box = [seed]
            dist.broadcast_object_list(box, src=0)
            seed = int(box[0])
        np.random.seed(seed)
        torch.manual_seed(seed)
        self._prewarmed = not bool(int(os.environ.get("FS_PREWARM_PROGRAMS", "1")))
        self.gc_freeze = bool(int(os.environ.get("FS_GC_FREEZE", "1"))) if gc_freeze is None else bool(gc_freeze)
        self.last_arch_ce = None
        self._phase_weights = None
        self._other_params = None          # weights outside the cells (stem, refinement, heads): autograd accumulates their gradients
        self.architect = None
        if not pretrain:
            if lut is not None:
                operations.latency_lookup_table.update(lut)
            args = type("Args", (), dict(momentum=cfg.momentum, weight_decay=cfg.weight_decay,
                                         arch_learning_rate=cfg.arch_learning_rate, latency_weight=cfg.latency_weight))
            self.architect = Architect(self.model, args, grad_sync=_allreduce_list)
            self.architect.loss_fn = self._loss_by_groups          # (eager architecture steps: the same pass groups as the graphed ones)

    # ---- the passes of one `_loss` call ---------------------------------------------------------------------------
    # A pass = (arch_idx to select or None to leave it, prun_mode); model_search.py `_loss` runs, in this order,
    #   pretrain:  (., max) (., min) (., random) (., random)
    #   search:    (0, None) (1, None) (., max) (., min)          None -> the arch's own mode: arch 0 "max", arch 1 "arch_ratio"
    # Passes whose widths are fixed ("max"/"min") have static shapes: forward + backward are captured once into a hipGraph
    # and replayed (~12 k launches -> one graph launch).  "random" / Gumbel "arch_ratio" passes change shape every step and
    # stay eager.
    def _specs(self):
        if self.pretrain:
            return [(None, "max"), (None, "min"), (None, "random"), (None, "random")]
        return [(0, None), (1, None), (None, "max"), (None, "min")]

    def _mode(self, spec):
        arch_idx, mode = spec
        return self.model._prun_modes[arch_idx] if mode is None else mode

    def _is_static(self, spec):
        return self._mode(spec) in ("max", "min")

    def _groups(self):
        """The passes of `_specs()` as tuples of ADJACENT passes that are evaluated together (forward_multi).  Two passes are joined when
        both replay from a graph or both are issued eagerly, the second keeps the architecture index of the first, and they are not the
        same sub-network (two fixed-width passes of one mode would meet in every BatchNorm's running statistics: nothing to group).
        Pairs only: a layer of two passes is ten MixedOp programs, a grouped launch carries twelve problems.  Adjacent, so that every
        BatchNorm sees its momentum updates in the reference's order (forward_multi / _run_tasks keep evaluations of one MixedOp at
        one output width in pass order); the bit-reproducible mode orders its gradient writes by stream and keeps single passes."""
        from . import kernels as K
        from . import model_search
        specs = self._specs()
        ok = (_JOINT_PASSES and model_search._PROGRAMS and model_search._GROUP_PROGRAMS and model_search._CAPTURE_PROGRAMS
              and model_search._GROUP_CAPTURE in (1, 2) and not K.deterministic_on())
        groups = []
        for spec in specs:
            last = groups[-1] if groups else None
            if (ok and last is not None and len(last) == 1 and spec[0] is None and self._is_static(last[0]) == self._is_static(spec)
                    and not (self._is_static(spec) and self._mode(last[0]) == self._mode(spec))):
                groups[-1] = last + (spec,)
            else:
                groups.append((spec,))
        return groups

    def _select(self, spec):
        arch_idx, mode = spec
        if arch_idx is not None:
            self.model.arch_idx = arch_idx
        self.model.prun_mode = mode

    def _run_pass(self, spec, imgs, target):
        self._select(spec)
        return sum(self.model._criterion(logit, target) for logit in self.model(imgs))

    def _run_group(self, group, imgs, target):
        """Loss of the passes of one group (a tuple of specs): the sum, pass by pass, of what _run_pass returns for each."""
        if len(group) == 1:
            return self._run_pass(group[0], imgs, target)
        out = self.model.forward_multi(imgs, list(group), batch_tails=True)
        from .model_search import JointLogits
        if isinstance(out, JointLogits):
            # the passes' logits along the batch: the criterion is a mean over the valid pixels, every pass sees the same labels, so
            # sum_p CE(logit_p, target) = passes * CE(logits of all passes, target repeated)
            t = torch.cat([target] * out.passes)
            return out.passes * sum(self.model._criterion(logit, t) for logit in out.logits)
        total = 0
        for logits in out:
            total = total + sum(self.model._criterion(logit, target) for logit in logits)
        return total

    def _pass_loss(self, mode, imgs, target):
        return self._run_pass((None, mode), imgs, target)

    def _set_phase(self, phase):
        """'w': network weights receive gradients, architecture parameters are frozen; 'a': the opposite.  The reference
        computes both sets in both phases and throws one away (train_search.py:245 optimizer.zero_grad / architect.py:38)."""
        if getattr(self, "_phase", None) == phase:          # (a pretrain run never leaves "w": 40 k requires_grad_ calls per step otherwise)
            return
        self._phase = phase
        weights = self.weights
        from . import model_search as _ms
        if self.use_graphs and getattr(self, "_capture_done", False) and _FAST_PHASE and _ms._PROGRAMS:
            # (only with launch programs: a MixedOp on the per-module path reads requires_grad of EVERY weight - ADVICE r5; _run_tasks
            # asserts that no MixedOp falls back to it while the subset is in charge)
            # A search iteration flips the phase twice: 80 k requires_grad_ calls = ~16 ms of a host-bound 160 ms step.  Once the
            # fixed-width passes are captured, the flags are only read (a) by autograd for the modules that run module by module in the
            # eager passes - stem, refine, heads - and (b) by MixedOp._program, which asks ONE weight per MixedOp whether the launch
            # programs should write weight gradients.  Those ~600 tensors are flipped; the other cell weights stay trainable (they are
            # only ever touched through the launch programs, which take their decision from (b)).
            if self._phase_weights is None:
                for p in self.weights:                       # leave the full set coherent before the subset takes over
                    p.requires_grad_(True)
                from . import model_search
                probe = {id(m._ops[1].conv1.weight) for m in self.model.modules() if isinstance(m, model_search.MixedOp)}
                cells = {id(p) for p in self.model.cells.parameters()}
                self._phase_weights = [p for p in self.weights if id(p) in probe or id(p) not in cells]
            weights = self._phase_weights
            _ms.FAST_PHASE_ACTIVE = True
        for p in weights:
            p.requires_grad_(phase == "w")
        for p in self.arch_params:
            p.requires_grad_(phase == "a")

    def _zero_arch_grads(self):
        for p in self.arch_params:          # persistent .grad tensors: captured AccumulateGrad nodes write into them
            if p.grad is None:
                p.grad = torch.zeros_like(p)
            else:
                p.grad.zero_()

    def _capture_pass(self, phase, group, side):
        imgs, target = self.static[phase]

        def fresh():
            if phase == "w":
                self.sync.prepare(passes=len(self._specs()))
            else:
                self._zero_arch_grads()
        fresh()
        # the warm-up pass is plumbing (allocator, pack caches), not a training step: BatchNorm running statistics and
        # num_batches_tracked are put back afterwards so that capture costs no extra momentum update (the reference runs
        # every pass exactly once per step)
        buffers = [b_ for b_ in self.model.buffers()]
        saved = [b_.clone() for b_ in buffers]
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                      # warm-up on the capture stream
            self._run_group(group, imgs, target).backward()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        with torch.no_grad():
            for b_, s_ in zip(buffers, saved):
                b_.copy_(s_)
        fresh()
        g = torch.cuda.CUDAGraph()
        from . import kernels as K
        with torch.cuda.graph(g, stream=side):
            K.zero_pool.begin_capture(imgs.device)          # one captured fill instead of two per conv-BN module
            try:
                loss = self._run_group(group, imgs, target)
                loss.backward()
            finally:
                arena = K.zero_pool.end_capture()
        touched = self.sync.touched_indices() if phase == "w" else None
        if phase == "w":
            self.sync.sync()
        self._graph_arenas.append(arena)
        return g, loss.detach(), touched

    def _capture(self, batches):
        """batches: {'w': (imgs, target)[, 'a': (imgs_search, target_search)]}."""
        try:        # the flat .grad views were created on the default stream; capture runs on a side stream by design
            torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
        except AttributeError:
            pass
        self.static = {ph: (i.clone(), t.clone()) for ph, (i, t) in batches.items()}
        self.graphs = {}
        self._graph_arenas = []
        from . import kernels as K_
        self._captured_deterministic = K_.deterministic_on()        # baked into the captured launches
        from . import model_search
        side = torch.cuda.Stream()
        lanes = model_search.branch_lanes(side)   # MixedOp forks its five primitives onto these inside the capture
        lanes = lanes + model_search.layer_lanes(side)            # ... and the MixedOps of a layer fork onto these
        from . import kernels as K
        for lane in [side] + lanes:               # every lane's conv workspace (scratch + zeroed arrival counters) exists before
            with torch.cuda.stream(lane):         # the capture: an allocation inside it would put the counter fill into the graph
                K.stream_workspace(self.static["w"][0].device)
        state = (self.model.arch_idx, self.model.prun_mode)
        self._pass_groups = self._groups()
        for phase in self.static:
            self._set_phase(phase)
            for group in self._pass_groups:
                for spec in group:
                    self._select(spec)      # also for the eager passes: later ones inherit arch_idx from them
                if self._is_static(group[0]) and (phase, group) not in self.graphs:
                    self.graphs[(phase, group)] = self._capture_pass(phase, group, side)
        self.model.arch_idx, self.model.prun_mode = state
        torch.cuda.synchronize()
        self._capture_done = True             # from here on the phase flips touch the ~600 tensors that are still consulted (_set_phase)

    def _phase_loss(self, phase, imgs, target):
        """All passes of `_loss(imgs, target)` with their backward; gradients accumulate pass by pass (d(sum)/dp = sum d/dp)."""
        s_imgs, s_target = self.static[phase]
        s_imgs.copy_(imgs)
        s_target.copy_(target)
        total = 0
        groups = self._pass_groups
        # Comm / compute overlap under DP: the gradient buckets are all-reduced under the backward of the LAST EAGER
        # group (FlatGradientSync.final_pass - a replayed graph runs no Python hooks and offers no point to launch a collective from).
        # When graphs are replayed after it (the search step: arch 0 | Gumbel pass | max + min graph), its FORWARD stays where the reference
        # has it - the BatchNorm running statistics see their updates in the reference's order - and its BACKWARD is issued after the
        # remaining replays: gradient accumulation commutes, and by then every other write of the step is enqueued.
        eager = [i for i, g_ in enumerate(groups) if not self._is_static(g_[0])]
        last_eager = eager[-1] if eager else None
        deferred = None
        for gi, group in enumerate(groups):
            if self._is_static(group[0]):
                g, loss, touched = self.graphs[(phase, group)]
                for spec in group:
                    self._select(spec)
                g.replay()
                if touched is not None:
                    self.sync.mark_touched(touched)
            else:
                # the LAST pass of the weight phase, issued eagerly, with more than one rank: its gradient buckets are all-reduced under
                # its own backward (FlatGradientSync.final_pass) - the launch programs of its forward say which writes
                # to expect, everything autograd accumulates itself (stem, refinement, heads) is expected once
                overlap = (phase == "w" and gi == last_eager and _DP_OVERLAP and (self.sync.world > 1 or _DP_OVERLAP > 1))
                from . import functional as FN
                if overlap:
                    FN._touch_log = []
                try:
                    loss = self._run_group(group, imgs, target)
                    log = FN._touch_log
                finally:
                    FN._touch_log = None
                usable = overlap and log is not None and all(p is not None for p in log)
                if usable and gi != len(groups) - 1:
                    deferred = (loss, log)
                else:
                    if usable:
                        self._declare_final_pass(log)
                    loss.backward()
                loss = loss.detach()
            total = total + loss
        if deferred is not None:
            loss, log = deferred
            self._declare_final_pass(log)
            loss.backward()
        return total

    def _declare_final_pass(self, log):
        if self._other_params is None:
            cells = {id(p) for p in self.model.cells.parameters()}
            self._other_params = [p for p in self.weights if id(p) not in cells]
        self.sync.final_pass(log, [p for p in self._other_params if p.requires_grad])

    def _graphed_pretrain_loss(self, imgs, target):
        return self._phase_loss("w", imgs, target)

    def step(self, imgs, target, imgs_search=None, target_search=None, force_eager=False):
        """One iteration.  force_eager=True issues every launch from the host (no hipGraph replay): same kernels, used by
        bench.py to take the launch census of a step."""
        from . import functional as FN
        from . import model_search
        FN.set_compute_dtype(self.compute_dtype)
        model_search.MIMIC_CAPTURE = bool(force_eager and self.use_graphs)
        try:
            if self.use_graphs and not force_eager:
                return self._step_graphed(imgs, target, imgs_search, target_search)
            return self._step_eager(imgs, target, imgs_search, target_search)
        finally:
            model_search.MIMIC_CAPTURE = False
            FN.set_compute_dtype(torch.float32)

    def _step_graphed(self, imgs, target, imgs_search, target_search):
        from . import kernels as K
        if self.graphs is None:
            batches = {"w": (imgs, target)}
            if self.architect is not None:
                batches["a"] = (imgs_search, target_search)
            self._capture(batches)
        if K.deterministic_on() != self._captured_deterministic:
            raise RuntimeError("fasterseg_amd: the bit-reproducible mode was switched %s after this SupernetStep captured its hipGraphs; the "
                               "captured launches keep the mode of the capture - set it (kernels.deterministic / FS_DETERMINISTIC=1) "
                               "BEFORE the first step, or build a new SupernetStep" % ("on" if K.deterministic_on() else "off"))
        loss_arch = None
        if self.architect is not None:                       # Architect.step (architect.py:34-47), pass by pass
            self._set_phase("a")
            self._zero_arch_grads()
            K.zero_pool.reset(imgs.device)
            try:
                loss_arch = self._phase_loss("a", imgs_search, target_search)
                self.last_arch_ce = loss_arch                 # `_loss` of the architecture step, before the latency penalty
                loss_latency = self.architect._latency_loss()
                if torch.is_tensor(loss_latency):
                    loss_latency.backward()
                    loss_arch = loss_arch + loss_latency.detach()
            finally:
                K.zero_pool.stop()
            if self.architect.grad_sync is not None:
                self.architect.grad_sync(self.arch_params)
            for optimizer in self.architect.optimizers:
                optimizer.step()
            self.model.note_arch_update()
        self._set_phase("w")
        self.sync.prepare(passes=len(self._specs()))
        loss = self._phase_loss("w", imgs, target)
        self.sync.sync()
        self.optimizer.step()
        if not self._prewarmed:               # after the first step every MixedOp has seen its call sites
            self._prewarmed = True
            self.prewarm_programs()
        return loss, loss_arch

    def prewarm_programs(self):
        """One-time setup, like the hipGraph capture of the fixed-width passes: lower the launch programs of every width
        combination the eager ("random" / Gumbel) passes can draw, then move the (now static) Python object graph - 40 k
        parameter tensors, ~2 k programs - out of the cyclic garbage collector's reach (gc.freeze): cProfile showed a
        supernet step spending more host time in first-use lowering and in generation-2 collections than in issuing its
        launches.  FS_PREWARM_PROGRAMS=0 keeps lowering on first use."""
        import gc
        from . import functional as FN
        from . import kernels as K
        from . import model_search
        mixed = [m for m in self.model.modules() if isinstance(m, model_search.MixedOp)]
        built = 0
        for phase in (("a", "w") if self.architect is not None else ("w",)):
            self._set_phase(phase)
            if phase == "w":
                self.sync.prepare(passes=len(self._specs()))
            try:
                for m in mixed:
                    built += m.prewarm_programs()
            finally:
                if phase == "w":          # leave the sink state without a (collective) sync
                    FN._grad_sink = None
                    K.zero_pool.stop()
        self._set_phase("w")
        self.programs_prewarmed = built
        # process-wide side effect, so it is the application's choice: FS_GC_FREEZE=0 (or freeze=False) leaves the collector alone
        if self.gc_freeze:
            gc.collect()
            gc.freeze()
        return built

    def describe(self):
        """How the passes of a step are executed (bench.py prints it)."""
        specs = self._specs()
        return {"passes_per_phase": len(specs), "pass_groups": [[self._mode(s_) for s_ in g] for g in self._groups()],
                "graphed": sum(1 for s_ in specs if self._is_static(s_)) if self.use_graphs else 0,
                "eager": sum(1 for s_ in specs if not (self.use_graphs and self._is_static(s_))),
                "programs_prewarmed": getattr(self, "programs_prewarmed", 0), "gc_frozen": bool(self.gc_freeze and self._prewarmed),
                "fused_bn_banks": self.fused_pairs}

    def _step_eager(self, imgs, target, imgs_search=None, target_search=None):
        loss_arch = None
        if self.architect is not None:
            # The reference's architect backward also produces (and then discards: optimizer.zero_grad, train_search.py:245)
            # gradients for every network weight.  They are never used, so the weights are frozen for the architecture
            # step: no wgrad / BN-parameter gradient kernels run, the alpha/beta/ratio gradients are unchanged.
            for p in self.weights:
                p.grad = None
            self.sync._hidden = None              # every .grad was dropped: the next prepare() re-points all of them
            self._set_phase("a")
            try:
                loss_arch = self.architect.step(imgs, target, imgs_search, target_search)
                self.last_arch_ce = self.architect.last_loss
            finally:
                self._set_phase("w")
        # `_loss` sums its four forwards and ONE backward runs over all of them - but a weight shared by the four forwards is
        # written by four wgrad launches, and the sink's hook fires on the first: buckets must not go out before sync()
        self.sync.prepare(passes=len(self._specs()))
        loss = self._loss_by_groups(imgs, target)
        loss.backward()
        self.sync.sync()
        self.optimizer.step()
        return loss.detach(), loss_arch

    def _loss_by_groups(self, imgs, target):
        """`model._loss(imgs, target, pretrain)` with the passes evaluated in the groups the graphed step uses (joint passes issue other -
        fewer, larger - launches than pass-by-pass evaluation, and the census step of bench.py must time the kernels of the timed steps)."""
        groups = self._groups()
        if all(len(g) == 1 for g in groups):
            return self.model._loss(imgs, target, self.pretrain)
        loss = 0
        for group in groups:
            loss = loss + self._run_group(group, imgs, target)
        return loss


def _allreduce_list(params):
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    grads = [p.grad for p in params if p.grad is not None]
    if grads:
        flat = torch.cat([g.reshape(-1) for g in grads])
        dist.all_reduce(flat)
        flat.div_(dist.get_world_size())
        off = 0
        for g in grads:
            g.copy_(flat[off:off + g.numel()].view_as(g)); off += g.numel()


def synthetic_batch(batch, height, width, rank, device, num_classes=19):
    """Images ~N(0,1) (post-normalisation Cityscapes pixels), labels uniform over classes with ~5 % ignore=255."""
    g = torch.Generator().manual_seed(1000 + rank)
    imgs = torch.randn(batch, 3, height, width, generator=g)
    target = torch.randint(0, num_classes, (batch, height, width), generator=g)
    target[torch.rand(batch, height, width, generator=g) < 0.05] = 255
    return imgs.to(device), target.to(device)
