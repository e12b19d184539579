"""Host time of one hipGraph replay of the student inference plan vs its device time (is the frame loop host-bound?).
Run on an MI355X:  python tools/replay_host_cost.py"""
import os
import sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fasterseg_amd import archs, engine
net = archs.build_derived(1, training=False); archs.init_weight(net, seed=12345); net = net.cuda().eval()
eng = engine.InferenceEngine(net, (1, 3, 1024, 2048), dtype=torch.bfloat16, logits_dtype=torch.float32)
print("candidates", eng.capture_log)
for _ in range(50): eng.run()
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(300): eng.run()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("enqueue %.4f ms/replay, complete %.4f ms/replay" % ((t1 - t0) / 300 * 1e3, (t2 - t0) / 300 * 1e3))
# latency of a single replay (no pipelining)
ts = []
for _ in range(50):
    torch.cuda.synchronize(); t0 = time.perf_counter(); eng.run(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
ts.sort(); print("single replay latency median %.4f ms" % (ts[len(ts)//2] * 1e3))
