"""A few hundred consecutive steps of a supernet workload on ONE batch (graph replays + eager passes, default switches): every loss finite,
the trajectory falling, gradients / weights finite at the end.  python tools/long_run.py c3|c5 [steps] [fp32]
(bench.py's post-timed check looks at ~15 steps, the graphed-vs-eager test at 24: this is the same question over a longer horizon.)"""
import os, sys, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fasterseg_amd import latency_lookup_table, train_step
which = sys.argv[1] if len(sys.argv) > 1 else "c3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
dt = torch.float32 if "fp32" in sys.argv[3:] else torch.bfloat16
pre = which == "c3"
b, h, w = (3, 256, 512) if pre else (2, 224, 448)
st = train_step.SupernetStep(pretrain=pre, lut=None if pre else latency_lookup_table.load_shipped("bf16"), compute_dtype=dt)
g = torch.Generator().manual_seed(1)
mk = lambda: (torch.randn(b, 3, h, w, generator=g).cuda(), torch.randint(0, 19, (b, h // 8, w // 8), generator=g).cuda())
(imgs, target), (imgs_s, target_s) = mk(), mk()
losses = []
for i in range(steps):
    out = st.step(imgs, target, imgs_s, target_s)
    if i % 25 == 0 or i == steps - 1:
        losses.append((i, float(out[0]), None if out[1] is None else float(out[1])))
        print("LONG_RUN %s %s step %4d loss %.4f%s" % (which, "fp32" if dt == torch.float32 else "bf16", i, losses[-1][1],
                                                       "" if losses[-1][2] is None else " arch %.4f" % losses[-1][2]), flush=True)
torch.cuda.synchronize()
ok = all(math.isfinite(l[1]) and (l[2] is None or math.isfinite(l[2])) for l in losses)
wf = all(bool(torch.isfinite(p).all()) for p in st.weights[::97])
gf = bool(torch.isfinite(st.sync.flat).all())
print("LONG_RUN %s: %d steps, finite losses %s, loss %.3f -> %.3f (falling: %s), sampled weights finite %s, flat gradient finite %s" %
      (which, steps, ok, losses[0][1], losses[-1][1], losses[-1][1] < losses[0][1], wf, gf))
