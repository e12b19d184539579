"""Runs only one train workload for a few steps (for rocprofv3 --kernel-trace --stats): python tools/profile_step.py c3|c4|c5 [steps]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fasterseg_amd import latency_lookup_table, train_step
which = sys.argv[1] if len(sys.argv) > 1 else "c4"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dt = torch.bfloat16 if os.environ.get("FS_DTYPE", "bf16") == "bf16" else torch.float32
if which == "c4":
    st = train_step.StudentDistillStep(12, 512, 1024, teacher_engine_dtype=dt, compute_dtype=dt)
    imgs, target = train_step.synthetic_batch(12, 512, 1024, 0, "cuda")
    run = lambda: st.step(imgs, target)
else:
    pre = which == "c3"
    b, h, w = (3, 256, 512) if pre else (2, 224, 448)
    st = train_step.SupernetStep(pretrain=pre, lut=None if pre else latency_lookup_table.load_shipped("bf16"), compute_dtype=dt)
    g = torch.Generator().manual_seed(1)
    mk = lambda: (torch.randn(b, 3, h, w, generator=g).cuda(), torch.randint(0, 19, (b, h // 8, w // 8), generator=g).cuda())
    (imgs, target), (imgs_s, target_s) = mk(), mk()
    run = lambda: st.step(imgs, target, imgs_s, target_s)
for _ in range(int(os.environ.get("FS_PROFILE_WARMUP", "2"))):
    run()
torch.cuda.synchronize()
import time
print("PROFILE_STEPS_BEGIN", steps, time.monotonic_ns(), flush=True)
for _ in range(steps):
    run()
torch.cuda.synchronize()
print("PROFILE_STEPS_END", steps, time.monotonic_ns(), flush=True)
