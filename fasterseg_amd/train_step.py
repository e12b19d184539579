"""The train steps that surround the hot path, restated from the reference drivers so they can be benchmarked on
synthetic batches (the drivers themselves need Cityscapes + cv2 + tensorboardX and are out of scope):

  StudentDistillStep  train/train.py:219-271 — teacher (eval, no_grad) forward, student (train) forward with three heads,
                      OHEM-CE(pred8) + 0.2*OHEM-CE(pred16) + 0.2*OHEM-CE(pred32) + KLDiv(student || teacher), backward,
                      SGD(momentum .9, wd 5e-4).  Under DP the flat gradient buffer is all-reduced between backward and
                      the optimizer step (parallel.FlatGradientSync).
Every conv / BN / resize in both networks runs on the HIP kernels (operations.py -> functional.py); losses and the
optimizer are PyTorch ops.
"""
import json
import time

import torch

from . import archs
from .losses import ProbOhemCrossEntropy2d, distill_kl
from .parallel import FlatGradientSync, broadcast_parameters


class StudentDistillStep:
    def __init__(self, batch, height, width, lr=0.01, momentum=0.9, weight_decay=5e-4, teacher_engine_dtype=None, seed=12345,
                 device="cuda"):
        self.device = torch.device(device)
        self.teacher = archs.init_weight(archs.build_derived(0, training=True), seed).to(self.device).eval()
        self.student = archs.init_weight(archs.build_derived(1, training=True), seed + 1).to(self.device).train()
        broadcast_parameters(self.student)
        broadcast_parameters(self.teacher)
        min_kept = int(batch * height * width // 16)                       # train/train.py:62 with gt_down_sampling = 1
        self.ohem = ProbOhemCrossEntropy2d(ignore_label=255, thresh=0.7, min_kept=min_kept, use_weight=False)
        self.optimizer = torch.optim.SGD(self.student.parameters(), lr=lr, momentum=momentum, weight_decay=weight_decay)
        self.sync = FlatGradientSync(self.student.parameters(), bucket_mb=256)
        self.lamb = 0.2
        self.teacher_engine = None
        if teacher_engine_dtype is not None:        # frozen teacher through the static-plan engine (hipGraph)
            from .engine import InferenceEngine
            self.teacher_engine = InferenceEngine(self.teacher, (batch, 3, height, width), dtype=teacher_engine_dtype)

    def teacher_logits(self, imgs):
        with torch.no_grad():
            if self.teacher_engine is not None:
                return self.teacher_engine(imgs)
            return self.teacher(imgs)

    def step(self, imgs, target):
        self.sync.prepare()
        t_logits = self.teacher_logits(imgs)
        p8, p16, p32 = self.student(imgs)
        loss = self.ohem(p8, target) + self.lamb * self.ohem(p16, target) + self.lamb * self.ohem(p32, target)
        loss = loss + distill_kl(p8, t_logits)
        loss.backward()
        self.sync.sync()
        self.optimizer.step()
        return loss.detach()


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

    def __init__(self, pretrain=True, cfg=SearchConfig, seed=12345, device="cuda", lut=None, use_graphs=None):
        import os
        self.use_graphs = bool(int(os.environ.get("FS_SUPERNET_GRAPHS", "1"))) if use_graphs is None else use_graphs
        self.graph_modes = ("max", "min")
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
        self.optimizer = torch.optim.SGD(self.weights, lr=cfg.lr, momentum=cfg.momentum, weight_decay=cfg.weight_decay)
        self.sync = FlatGradientSync(self.weights, bucket_mb=128)
        np.random.seed(seed)
        torch.manual_seed(seed)
        self.architect = None
        if not pretrain:
            if lut is not None:
                operations.latency_lookup_table.update(lut)
            args = type("Args", (), dict(momentum=cfg.momentum, weight_decay=cfg.weight_decay,
                                         arch_learning_rate=cfg.arch_learning_rate, latency_weight=cfg.latency_weight))
            self.architect = Architect(self.model, args, grad_sync=_allreduce_list)

    # ---- hipGraph capture of the shape-static passes ("max" and "min" widths) ----------------------------------
    def _capture(self, imgs, target):
        """Forward + backward of one fixed-width supernet pass as a single hipGraph (gradients accumulate into the flat
        buffer, BN running statistics update on replay).  ~25 k launches per pass collapse into one graph launch; the
        two "random"-width passes of a pretrain step change shape every step and stay eager."""
        self.static_imgs, self.static_target = imgs.clone(), target.clone()
        self.graphs = {}
        try:        # the flat .grad views were created on the default stream; capture runs on a side stream by design
            torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
        except AttributeError:
            pass
        side = torch.cuda.Stream()
        for mode in self.graph_modes:
            self.sync.prepare()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):                      # warm-up on the capture stream (allocator, pack caches)
                self._pass_loss(mode, self.static_imgs, self.static_target).backward()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            self.sync.prepare()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                loss = self._pass_loss(mode, self.static_imgs, self.static_target)
                loss.backward()
            self.graphs[mode] = (g, loss.detach(), self.sync.touched_indices())
            self.sync.sync()
        torch.cuda.synchronize()

    def _pass_loss(self, mode, imgs, target):
        self.model.prun_mode = mode
        return sum(self.model._criterion(logit, target) for logit in self.model(imgs))

    def _graphed_pretrain_loss(self, imgs, target):
        """`_loss(imgs, target, pretrain=True)` with the static passes replayed from graphs; same pass order as the reference
        (max, min, random, random), gradients accumulated pass by pass (d(sum)/dw = sum of d/dw)."""
        if self.graphs is None:
            self._capture(imgs, target)
            self.sync.prepare()
        self.static_imgs.copy_(imgs)
        self.static_target.copy_(target)
        total = 0
        for mode in ("max", "min"):
            g, loss, touched = self.graphs[mode]
            g.replay()
            self.sync.mark_touched(touched)
            total = total + loss
        for _ in range(2):
            loss = self._pass_loss("random", imgs, target)
            loss.backward()
            total = total + loss.detach()
        return total

    def step(self, imgs, target, imgs_search=None, target_search=None):
        if self.use_graphs and self.pretrain and self.architect is None:
            self.sync.prepare()
            loss = self._graphed_pretrain_loss(imgs, target)
            self.sync.sync()
            torch.nn.utils.clip_grad_norm_(self.weights, self.cfg.grad_clip)
            self.optimizer.step()
            return loss.detach() if torch.is_tensor(loss) else loss, None
        loss_arch = None
        if self.architect is not None:
            # The reference's architect backward also produces (and then discards: optimizer.zero_grad, train_search.py:245)
            # gradients for every network weight.  They are never used, so the weights are frozen for the architecture
            # step: no wgrad / BN-parameter gradient kernels run, the alpha/beta/ratio gradients are unchanged.
            for p in self.weights:
                p.grad = None
                p.requires_grad_(False)
            try:
                loss_arch = self.architect.step(imgs, target, imgs_search, target_search)
            finally:
                for p in self.weights:
                    p.requires_grad_(True)
        self.sync.prepare()
        loss = self.model._loss(imgs, target, self.pretrain)
        loss.backward()
        self.sync.sync()
        torch.nn.utils.clip_grad_norm_(self.weights, self.cfg.grad_clip)
        self.optimizer.step()
        return loss.detach(), loss_arch


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


def bench_student_train(args, world, rank, barrier, max_over_ranks):
    batch = args.batch or 12
    H, W = (args.height, args.width) if (args.height, args.width) != (1024, 2048) else (512, 1024)   # config C4 crop
    eng_dtype = {"bf16": torch.bfloat16, "fp32": torch.float32}[args.dtype]
    stepper = StudentDistillStep(batch, H, W, teacher_engine_dtype=eng_dtype)
    imgs, target = synthetic_batch(batch, H, W, rank, "cuda")
    for _ in range(args.warmup):
        stepper.step(imgs, target)
    barrier(world)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = stepper.step(imgs, target)
    barrier(world)
    elapsed = max_over_ranks(time.perf_counter() - t0, world)
    ips = world * batch * args.steps / elapsed
    return {
        "metric": "supernet train-step images/sec @1024x2048 (1/2/4/8 GPU) + student fps",
        "value": round(ips, 3), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "fp32 student (exact-fp32 MFMA); teacher engine %s" % args.dtype, "data": "synthetic",
        "config": {"workload": "student KL-distillation train step (BASELINE configs[3]): %d x 3x%dx%d per GPU, teacher arch_0 eval "
                               "+ student arch_1 train (3 heads), OHEM-CE + KLDiv, SGD" % (batch, H, W),
                   "global_batch": world * batch, "parallelism": "dp%d, flat fp32 gradient bucket all-reduce (RCCL)" % world},
        "final_loss": float(loss),
    }


def bench_supernet(args, world, rank, barrier, max_over_ranks, pretrain):
    """BASELINE configs[2] (pretrain, 3 x 3x256x512) / configs[4] (search, 2 x 3x224x448 per GPU); labels at 1/8 resolution."""
    import os
    batch = args.batch or (3 if pretrain else 2)
    default_hw = (256, 512) if pretrain else (224, 448)
    H, W = (args.height, args.width) if (args.height, args.width) != (1024, 2048) else default_hw
    lut = None
    if not pretrain:
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "latency_lut_1080ti.json")
        with open(path) as f:
            lut = json.load(f)      # the reference's shipped table until fasterseg_amd.latency_lookup_table regenerates it
    stepper = SupernetStep(pretrain=pretrain, lut=lut)
    g = torch.Generator().manual_seed(2000 + rank)

    def make():
        imgs = torch.randn(batch, 3, H, W, generator=g).cuda()
        tgt = torch.randint(0, 19, (batch, H // 8, W // 8), generator=g)
        tgt[torch.rand(batch, H // 8, W // 8, generator=g) < 0.05] = 255
        return imgs, tgt.cuda()
    imgs, target = make()
    imgs_s, target_s = make()
    for _ in range(args.warmup):
        stepper.step(imgs, target, imgs_s, target_s)
    barrier(world)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, loss_arch = stepper.step(imgs, target, imgs_s, target_s)
    barrier(world)
    elapsed = max_over_ranks(time.perf_counter() - t0, world)
    ips = world * batch * args.steps / elapsed
    name = "supernet pretrain step (BASELINE configs[2])" if pretrain else "architecture-search step: arch update + weight update (BASELINE configs[4])"
    return {
        "metric": "supernet train-step images/sec @1024x2048 (1/2/4/8 GPU) + student fps",
        "value": round(ips, 4), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "fp32 (exact-fp32 MFMA)", "data": "synthetic",
        "config": {"workload": "%s: %d x 3x%dx%d per GPU, F12.L16, widths {4,6,8,10,12}/12, all 5 primitives per MixedOp, "
                               "fwd+bwd, clip 5, SGD" % (name, batch, H, W),
                   "global_batch": world * batch, "parallelism": "dp%d, flat fp32 gradient buckets all-reduced over RCCL" % world},
        "final_loss": float(loss), "arch_loss": None if loss_arch is None else float(loss_arch),
    }
