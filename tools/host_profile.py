"""cProfile of the HOST side of a supernet step (the step is host-bound: tools/host_vs_device.py).  python tools/host_profile.py c3|c5 [steps]
Prints the heaviest functions by own time and by cumulative time over `steps` steps (no device synchronisation inside the profiled region)."""
import cProfile, io, os, pstats, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fasterseg_amd import latency_lookup_table, train_step
which = sys.argv[1] if len(sys.argv) > 1 else "c3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
pre = which == "c3"
b, h, w = (3, 256, 512) if pre else (2, 224, 448)
st = train_step.SupernetStep(pretrain=pre, lut=None if pre else latency_lookup_table.load_shipped("bf16"), compute_dtype=torch.bfloat16)
g = torch.Generator().manual_seed(1)
mk = lambda: (torch.randn(b, 3, h, w, generator=g).cuda(), torch.randint(0, 19, (b, h // 8, w // 8), generator=g).cuda())
(imgs, target), (imgs_s, target_s) = mk(), mk()
for _ in range(3):
    st.step(imgs, target, imgs_s, target_s)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    st.step(imgs, target, imgs_s, target_s)
pr.disable()
torch.cuda.synchronize()
for key, n in (("tottime", 45), ("cumulative", 60)):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).strip_dirs().sort_stats(key).print_stats(n)
    text = s.getvalue()
    print("HOST_PROFILE %s x%d sorted by %s" % (which, steps, key))
    print("\n".join(line[:200] for line in text.splitlines()[4:]))
