"""Per-geometry timing of one train step's conv launches (census level 2): python tools/census_shapes.py c3|c4|c5 [fp32] [--json out]
Run it under FS_IGEMM2=0 and with the default to compare the two implicit-GEMM kernels geometry by geometry inside a real step."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fasterseg_amd import census, latency_lookup_table, train_step
which = sys.argv[1] if len(sys.argv) > 1 else "c4"
dt = torch.float32 if "fp32" in sys.argv[2:] else torch.bfloat16
if which == "c4":
    st = train_step.StudentDistillStep(12, 512, 1024, teacher_engine_dtype=dt, compute_dtype=dt)
    imgs, target = train_step.synthetic_batch(12, 512, 1024, 0, "cuda")
    run = lambda: st.step(imgs, target)
else:
    pre = which == "c3"
    b, h, w = (3, 256, 512) if pre else (2, 224, 448)
    st = train_step.SupernetStep(pretrain=pre, lut=None if pre else latency_lookup_table.load_shipped("bf16"), compute_dtype=dt, use_graphs=False)
    g = torch.Generator().manual_seed(1)
    mk = lambda: (torch.randn(b, 3, h, w, generator=g).cuda(), torch.randint(0, 19, (b, h // 8, w // 8), generator=g).cuda())
    (imgs, target), (imgs_s, target_s) = mk(), mk()
    run = lambda: st.step(imgs, target, imgs_s, target_s)
for _ in range(3):
    run()
torch.cuda.synchronize()
with census.recording(2) as rec:
    run()
rows = {}
for family, d, count, ms in rec.entries:
    if (family & 0xff) != census.IGEMM or ms <= 0:
        continue
    key = "N%d %dx%d %d->%d k%d s%d fl%x g%d" % (d.N, d.H, d.W, d.Cin, d.Cout, d.R, d.stride, d.flags, d.bn_groups)
    r = rows.setdefault(key, [0, 0.0, 0.0])
    r[0] += count
    r[1] += ms
    r[2] += census.conv_flops(d) * count
tot = sum(r[1] for r in rows.values())
print("CENSUS %s igemm family: %.3f ms in %d launches, FS_IGEMM2=%s" % (which, tot, sum(r[0] for r in rows.values()), os.environ.get("FS_IGEMM2", "default")))
for k, r in sorted(rows.items(), key=lambda kv: -kv[1][1])[:40]:
    print("  %-44s x%-3d %8.2f us avg %7.1f TF/s  %6.3f ms" % (k, r[0], r[1] / r[0] * 1e3, r[2] / (r[1] * 1e-3) / 1e12, r[1]))
if "--json" in sys.argv:
    json.dump({k: {"launches": r[0], "ms": r[1], "flops": r[2]} for k, r in rows.items()}, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)
