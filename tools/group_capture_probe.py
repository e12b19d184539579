"""Does a hipGraph capture survive grouped launches (kernel arguments of 1-2 KB)?  python tools/group_capture_probe.py K"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fasterseg_amd import fusion, kernels as K, model_search  # noqa: E402
from fasterseg_amd.parallel import FlatGradientSync  # noqa: E402

WIDTHS = [4. / 12, 6. / 12, 8. / 12, 10. / 12, 1.]
k = int(sys.argv[1])
backward = len(sys.argv) > 2 and sys.argv[2] == "bwd"
dtype = torch.bfloat16
torch.manual_seed(5)
ops = torch.nn.ModuleList([model_search.MixedOp(48, 48, stride=1, width_mult_list=WIDTHS) for _ in range(k)]).cuda().train()
for m in ops:
    fusion.colocate(m)
sync = FlatGradientSync(fusion.flat_order(ops, ops.parameters()))
ratios = [(WIDTHS[2], WIDTHS[3])] * k
xs, coefs = [], []
for m in ops:
    m.set_prun_ratio(ratios[0])
    cout, cin = m._ops[1].conv1.active_channels()
    xs.append(K.to_nhwc(torch.randn(2, cin, 16, 24, device="cuda"), dtype).requires_grad_(True))
    coefs.append(torch.softmax(torch.randn(5, device="cuda"), 0))
sync.prepare()


def run():
    outs = model_search._run_tasks([(m, x, c, r, 1) for m, x, c, r in zip(ops, xs, coefs, ratios)])
    if backward:
        torch.autograd.backward(outs, [torch.ones_like(o) for o in outs])
    return outs


for _ in range(2):
    run()
torch.cuda.synchronize()
print("eager ok", flush=True)
s = torch.cuda.Stream()
model_search.layer_lanes(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(s):
    K.stream_workspace("cuda")
    for lane in model_search.layer_lanes(s):
        with torch.cuda.stream(lane):
            K.stream_workspace("cuda")
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        outs = run()
print("captured", flush=True)
g.replay()
torch.cuda.synchronize()
print("replayed k=%d %s ok" % (k, "bwd" if backward else "fwd"), flush=True)
