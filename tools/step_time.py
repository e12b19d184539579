"""Wall time of one train workload's step (no profiler): python tools/step_time.py c3|c4|c5 [steps] [fp32]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fasterseg_amd import latency_lookup_table, train_step
which = sys.argv[1] if len(sys.argv) > 1 else "c3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dt = torch.float32 if "fp32" in sys.argv[3:] else torch.bfloat16
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
for _ in range(3):
    out = run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    out = run()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / steps * 1e3
print("STEP_TIME %s %s lanes=%s: %.2f ms/step (loss %s)" % (which, "fp32" if dt == torch.float32 else "bf16", os.environ.get("FS_EAGER_LANES", "default"),
                                                        ms, float(out[0] if isinstance(out, tuple) else out)), flush=True)
