"""Is a supernet step bound by the host (issuing launches) or by the device (executing them)?   python tools/host_vs_device.py c3|c5 [steps]

Per step, without any profiler: `enqueue` = wall time until step() has returned (everything issued: graph replays, the eager passes' launch
programs, the optimizer), `done` = until the device has drained.  enqueue ~ done: the host is the critical path of the eager passes (fewer /
cheaper FFI crossings and launches matter, kernel time does not); enqueue << done: the device is (kernel count and duration matter).
Also times the host side of the parts: graph replays, eager forward, eager backward (no synchronisation inside the step)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fasterseg_amd import latency_lookup_table, train_step
which = sys.argv[1] if len(sys.argv) > 1 else "c3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
if which == "c2":            # the student frame: hipGraph replays back to back
    from fasterseg_amd import archs, engine
    net = archs.init_weight(archs.build_derived(1, training=False), seed=12345).cuda().eval()
    eng = engine.InferenceEngine(net, (1, 3, 1024, 2048), dtype=torch.bfloat16, logits_dtype=torch.float32)
    eng.input.copy_(torch.randn(1, 3, 1024, 2048, device="cuda"))
    for _ in range(200):
        eng.run()
    torch.cuda.synchronize()
    n = max(steps, 2000)
    t0 = time.perf_counter()
    for _ in range(n):
        eng.run()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("HOST_VS_DEVICE c2: enqueue %.4f ms/frame, done %.4f ms/frame over %d frames (device tail %.2f ms); lanes %s" %
          ((t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3, n, (t2 - t1) * 1e3, getattr(eng, "graph_lanes", "?")))
    # one frame at a time: the host cost of ONE replay with an idle queue
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        eng.run()
        torch.cuda.synchronize()
    print("   one frame at a time (replay + drain): %.4f ms" % ((time.perf_counter() - t0) / 200 * 1e3))
    sys.exit(0)
pre = which == "c3"
b, h, w = (3, 256, 512) if pre else (2, 224, 448)
st = train_step.SupernetStep(pretrain=pre, lut=None if pre else latency_lookup_table.load_shipped("bf16"), compute_dtype=torch.float32 if "fp32" in sys.argv[3:] else torch.bfloat16)
g = torch.Generator().manual_seed(1)
mk = lambda: (torch.randn(b, 3, h, w, generator=g).cuda(), torch.randint(0, 19, (b, h // 8, w // 8), generator=g).cuda())
(imgs, target), (imgs_s, target_s) = mk(), mk()
for _ in range(3):
    st.step(imgs, target, imgs_s, target_s)
torch.cuda.synchronize()
# host-side split of the weight phase: wrap the pass runner
parts = {}
orig_run, orig_replay, orig_group = st._run_pass, torch.cuda.CUDAGraph.replay, st._run_group


def run_pass(spec, im, tg):
    t0 = time.perf_counter()
    out = orig_run(spec, im, tg)
    parts["eager fwd (host)"] = parts.get("eager fwd (host)", 0.0) + time.perf_counter() - t0
    return out


def replay(self):
    t0 = time.perf_counter()
    orig_replay(self)
    parts["graph replays (host)"] = parts.get("graph replays (host)", 0.0) + time.perf_counter() - t0


def run_group(group, im, tg):
    if len(group) == 1:
        return orig_group(group, im, tg)          # -> run_pass
    t0 = time.perf_counter()
    out = orig_group(group, im, tg)
    parts["eager fwd (host)"] = parts.get("eager fwd (host)", 0.0) + time.perf_counter() - t0
    return out


st._run_pass = run_pass
st._run_group = run_group
torch.cuda.CUDAGraph.replay = replay
enq = done = 0.0
for _ in range(steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    st.step(imgs, target, imgs_s, target_s)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    enq += t1 - t0
    done += t2 - t0
print("HOST_VS_DEVICE %s: enqueue %.2f ms/step, done %.2f ms/step (device tail after the host finished: %.2f ms)" %
      (which, enq / steps * 1e3, done / steps * 1e3, (done - enq) / steps * 1e3))
for k, v in parts.items():
    print("   %-24s %.2f ms/step" % (k, v / steps * 1e3))
print("   (synchronised per step: one step no longer overlaps the next one's host work; the free-running figure is tools/step_time.py)")
