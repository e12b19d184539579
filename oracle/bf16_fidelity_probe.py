"""TEST INFRASTRUCTURE / measurement (build container only: imports the unmodified reference) - where does the bf16 train path lose its
deep-layer gradients?  Runs the reference's F12.L16 supernet `_loss` + backward (model_search.py:478-505) on the supernet_l16 fixture's
weights and batch, in fp32 arithmetic, with emulated storage rounding:

    fp32            nothing rounded (the reference's own fp32-vs-fp64 gap)
    fwd             every module output (conv, BN, ReLU, resample, MixedOp sum) rounded to bf16 in FORWARD only; gradients stay fp32
                    - the "fp32-gradient backward for bf16 storage" design
    fwd+bwd         the same, and every gradient crossing those points rounded to bf16 (what the HIP bf16 path stores)
    fwd+w           forward rounding plus bf16 filter banks
    only-conv-out / only-bn-out / only-relu-out / only-resample / all-but-conv-out     forward rounding at one kind of tensor only

and prints, per sampled parameter gradient, cosine and relative L2 against the fp64 run.  python -m oracle.bf16_fidelity_probe [pretrain|search]
"""
import json
import sys

import torch
import torch.nn.functional as F

from . import ref_loader
from .make_golden import L16_CASES, L16_CFG
from .seeded import seeded_input, seeded_state


class _RoundSTE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, bwd):
        ctx.bwd = bwd
        return x.to(torch.bfloat16).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return (g.to(torch.bfloat16).to(g.dtype) if ctx.bwd else g), None


def run(mode_name, rounding, which="pretrain"):
    shape = dict(L16_CASES)[which]
    with ref_loader.reference("search"):
        import model_search
        real_cuda = torch.Tensor.cuda
        torch.Tensor.cuda = lambda self, *a, **k: self
        real_interp = F.interpolate
        try:
            crit = torch.nn.CrossEntropyLoss(ignore_index=255)
            net = model_search.Network_Multi_Path(criterion=crit, **L16_CFG)
            sd = seeded_state(net.state_dict(), 778)
            for k in list(sd):
                if k.split("_")[0] in ("alpha", "beta", "ratio"):
                    sd[k] = sd[k] * 5.0
            dt = torch.float64 if rounding == "fp64" else torch.float32
            net.load_state_dict(sd)
            net = net.to(dt).train()
            fwd = rounding in ("fwd", "fwd+bwd", "fwd+w")
            bwd = rounding == "fwd+bwd"
            only = {"only-conv-out": torch.nn.Conv2d, "only-bn-out": torch.nn.modules.batchnorm._BatchNorm, "only-relu-out": torch.nn.ReLU}.get(rounding)
            if fwd or only is not None or rounding in ("only-resample", "all-but-conv-out"):
                def hook(mod, inp, out):
                    return _RoundSTE.apply(out, bwd) if torch.is_tensor(out) and out.is_floating_point() and out.dim() == 4 else out
                for m in net.modules():
                    if rounding == "all-but-conv-out":
                        if not isinstance(m, torch.nn.Conv2d):
                            m.register_forward_hook(hook)
                    elif fwd or (only is not None and isinstance(m, only)):
                        m.register_forward_hook(hook)
                if fwd or rounding in ("only-resample", "all-but-conv-out"):
                    F.interpolate = lambda *a, **k: _RoundSTE.apply(real_interp(*a, **k), bwd)
                    torch.nn.functional.interpolate = F.interpolate
            if rounding == "fwd+w":
                for m in net.modules():
                    if isinstance(m, torch.nn.Conv2d):
                        m.register_forward_pre_hook(lambda mod, inp: None)
                        w = m.weight
                        m.weight.data = w.data.to(torch.bfloat16).to(w.dtype)
            x = seeded_input(shape, 41).to(dt)
            g = torch.Generator().manual_seed(42)
            target = torch.randint(0, 19, (shape[0], shape[2] // 8, shape[3] // 8), generator=g)
            target[torch.rand(target.shape, generator=g) < 0.05] = 255
            net.zero_grad()
            import numpy as np
            np.random.seed(5)
            torch.manual_seed(6)
            net.arch_idx = 0
            loss = net._loss(x, target, which == "pretrain")
            loss.backward()
            grads = {k: p.grad.detach().to(torch.float64).clone() for k, p in net.named_parameters() if p.grad is not None}
            return float(loss.detach()), grads
        finally:
            torch.Tensor.cuda = real_cuda
            F.interpolate = real_interp
            torch.nn.functional.interpolate = real_interp


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "pretrain"
    loss64, g64 = run("fp64", "fp64", which)
    import zlib
    names = [k for k in sorted(g64) if k.split("_")[0] not in ("alpha", "beta", "ratio") and g64[k].numel() >= 64]
    pick = [k for k in names if zlib.crc32(k.encode()) % 40 == 0]
    out = {"loss_fp64": loss64, "tensors": len(pick)}
    for rounding in (sys.argv[2].split(",") if len(sys.argv) > 2 else ("fp32", "fwd", "fwd+bwd", "fwd+w")):
        loss, g = run(rounding, rounding, which)
        cos, rel = [], []
        for k in pick:
            a, b = g64[k].flatten(), g[k].flatten()
            cos.append(float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-300)))
            rel.append(float((a - b).norm() / (a.norm() + 1e-300)))
        depth = lambda k: int(k.split(".")[1]) if k.startswith("cells.") else 99
        deep = [c for k, c in zip(pick, cos) if depth(k) <= 5]
        cs = sorted(cos)
        out[rounding] = {"loss_rel_err": abs(loss - loss64) / abs(loss64), "cos_min": cs[0], "cos_p10": cs[len(cs) // 10], "cos_median": cs[len(cs) // 2],
                         "cos_deep_layers_0_5_median": sorted(deep)[len(deep) // 2] if deep else None,
                         "rel_l2_median": sorted(rel)[len(rel) // 2], "rel_l2_max": max(rel)}
        print(which, rounding, json.dumps(out[rounding]), flush=True)
    with open("profiles/r04_bf16_fidelity_probe_%s%s.json" % (which, "_selective" if len(sys.argv) > 2 else ""), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
