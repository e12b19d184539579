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
