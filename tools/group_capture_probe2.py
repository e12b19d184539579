"""Two layers of MixedOps, several lockstep buckets per layer on side lanes, forward + backward inside ONE hipGraph capture."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fasterseg_amd import functional as FN, fusion, kernels as K, model_search  # noqa: E402
from fasterseg_amd.parallel import FlatGradientSync  # noqa: E402

WIDTHS = [4. / 12, 6. / 12, 8. / 12, 10. / 12, 1.]
variant = sys.argv[1] if len(sys.argv) > 1 else "all"
dtype = torch.bfloat16
torch.manual_seed(5)
kinds1 = {"all": (1, 1, 1, 2, 2), "one_bucket": (1, 1, 1), "k1": (1, 2), "two_buckets": (1, 1, 2, 2)}[variant]
layer1 = torch.nn.ModuleList([model_search.MixedOp(48, 48 * s, stride=s, width_mult_list=WIDTHS) for s in kinds1]).cuda().train()
layer2 = torch.nn.ModuleList([model_search.MixedOp(48, 48, stride=1, width_mult_list=WIDTHS) for _ in range(2)]).cuda().train()
for m in list(layer1) + list(layer2):
    fusion.colocate(m)
    m.set_prun_ratio((1., 1.))
allp = torch.nn.ModuleList([layer1, layer2])
sync = FlatGradientSync(fusion.flat_order(allp, allp.parameters()))
x0 = K.to_nhwc(torch.randn(2, 48, 16, 24, device="cuda"), dtype).requires_grad_(True)
coef = torch.softmax(torch.randn(5, device="cuda"), 0)
r = (1., 1.)


def run():
    outs1 = model_search._run_tasks([(m, x0, coef, r, 1) for m in layer1])
    s1 = [o for o, k in zip(outs1, kinds1) if k == 1]
    h = FN.weighted_sum(s1, torch.ones(len(s1), device="cuda") / len(s1)) if len(s1) > 1 else s1[0]
    outs2 = model_search._run_tasks([(m, h, coef, r, 1) for m in layer2])
    loss = sum(o.float().square().mean() for o in outs2) + sum(o.float().mean() for o, k in zip(outs1, kinds1) if k == 2)
    loss.backward()
    return loss


sync.prepare(passes=4)
for _ in range(2):
    run()
torch.cuda.synchronize()
print("eager ok", flush=True)
s = torch.cuda.Stream()
lanes = model_search.branch_lanes(s) + model_search.layer_lanes(s)
for lane in [s] + lanes:
    with torch.cuda.stream(lane):
        K.stream_workspace("cuda")
torch.cuda.synchronize()
try:
    torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
except AttributeError:
    pass
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    run()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    K.zero_pool.begin_capture("cuda")
    try:
        loss = run()
    finally:
        arena = K.zero_pool.end_capture()
print("captured", flush=True)
g.replay()
torch.cuda.synchronize()
print("replayed %s ok loss %.4f" % (variant, float(loss)), flush=True)
