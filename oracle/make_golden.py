"""TEST INFRASTRUCTURE — generates tests/golden/* from the UNMODIFIED reference (build container only).

    python -m oracle.make_golden            # rewrites every fixture

The reference ships no tests or golden vectors; these fixtures are outputs of
the reference's own modules (train/operations.py, train/seg_oprs.py, train/model_seg.py,
search/model_search.py) run on CPU with seeded parameters (oracle/seeded.py) and are what pins both
oracle/ref_ops.py and the HIP path.  Nothing here is imported by the product.
"""
import json
import os
import sys

import numpy as np
import torch

from . import ref_loader
from .seeded import seeded_input, seeded_state

GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
WML = [4. / 12, 6. / 12, 8. / 12, 10. / 12, 1.]
OPS_NAME = ["FactorizedReduce", "BasicResidual1x", "BasicResidual_downup_1x", "BasicResidual2x",
            "BasicResidual_downup_2x"]   # search/operations.py:546


def _np(t):
    return t.detach().cpu().numpy()


BIG, STEP = 30000, 5
SHAPES = {}


def _put(store, key, t):
    """Arrays above BIG elements are stored as flat[::STEP] under key+'@STEP' (fixture size)."""
    a = _np(t)
    if a.size > BIG:
        store["%s@%d" % (key, STEP)] = a.reshape(-1)[::STEP].copy()
    else:
        store[key] = a


def _load_seeded(module, seed):
    sd = seeded_state(module.state_dict(), seed)
    module.load_state_dict(sd)
    return sd


# ------------------------------------------------------------------------------------------------
# 1. shipped architectures: raw arch params + decoded structure
# ------------------------------------------------------------------------------------------------
def build_ref_net(model_seg, state, idx, lasts, training):
    teacher = (idx == 0)
    net = model_seg.Network_Multi_Path_Infer(
        [state["alpha_%d_0" % idx].detach().clone(), state["alpha_%d_1" % idx].detach().clone(),
         state["alpha_%d_2" % idx].detach().clone()],
        [None, state["beta_%d_1" % idx].detach().clone(), state["beta_%d_2" % idx].detach().clone()],
        [state["ratio_%d_0" % idx].detach().clone(), state["ratio_%d_1" % idx].detach().clone(),
         state["ratio_%d_2" % idx].detach().clone()],
        num_classes=19, layers=16, Fch=12, width_mult_list=WML,
        stem_head_width=(1., 1.) if teacher else (8. / 12, 8. / 12), ignore_skip=teacher)
    net.train(training)
    net.build_structure(list(lasts))
    return net


def net_meta(net):
    cells = {}
    for key, cell in net.cells.items():
        cells[key] = {"op": OPS_NAME.index(cell._op._op.__class__.__name__),
                      "down": int(bool(cell._down)), "C_in": int(cell._C_in), "C_out": int(cell._C_out)}
    return {
        "lasts": [int(v) for v in net.lasts],
        "ops": [[int(o) for o in ops] for ops in net.ops],
        "paths": [[int(v) for v in p] for p in net.paths],
        "downs": [[int(v) for v in p] for p in net.downs],
        "widths": [[float(v) for v in p] for p in net.widths],
        "branch_groups": [[[int(b) for b in g] for g in groups] for groups in net.branch_groups],
        "cells": cells,
        "ch_16": int(net.ch_16), "ch_8_2": int(net.ch_8_2), "ch_8_1": int(net.ch_8_1),
        "state_shapes": {k: list(v.shape) for k, v in net.state_dict().items()},
        "num_params": int(sum(p.numel() for p in net.parameters())),
    }


def gen_arch():
    with ref_loader.reference("train") as wd:
        import model_seg
        from utils.darts_utils import objective_acc_lat
        for idx in (0, 1):
            state = torch.load(os.path.join(wd, "fasterseg", "arch_%d.pt" % idx), map_location="cpu",
                               weights_only=False)
            raw = {}
            for k, v in state.items():
                raw[k] = _np(v) if torch.is_tensor(v) else np.asarray(v)
            np.savez_compressed(os.path.join(GOLD, "arch_%d.npz" % idx), **raw)
            # the searched architectures are data the reference ships (train/fasterseg/arch_{0,1}.pt); the package
            # carries the same numbers as .npz so the derived networks can be built without the reference tree
            pkg = os.path.join(os.path.dirname(GOLD), "..", "fasterseg_amd", "fasterseg")
            os.makedirs(pkg, exist_ok=True)
            np.savez_compressed(os.path.join(pkg, "arch_%d.npz" % idx), **raw)
            lat = lambda k: float(state[k])
            meta = {"objective02": float(objective_acc_lat(float(state["mIoU02"]), lat("latency02"))),
                    "objective12": float(objective_acc_lat(float(state["mIoU12"]), lat("latency12")))}
            for training in (False, True):
                for lasts in ([2, 1], [2, 0], [1], [2]):
                    net = build_ref_net(model_seg, state, idx, lasts, training)
                    meta["%s_%s" % ("train" if training else "eval", "".join(map(str, lasts)))] = net_meta(net)
            with open(os.path.join(GOLD, "arch_%d.json" % idx), "w") as f:
                json.dump(meta, f)
            print("arch_%d: eval_21 params" % idx, meta["eval_21"]["num_params"])


# ------------------------------------------------------------------------------------------------
# 2. decode property cases: random arch params through network_metas (with its in-place mutation)
# ------------------------------------------------------------------------------------------------
def gen_decode_cases(n_cases=48):
    out = []
    with ref_loader.reference("train"):
        import model_seg
        rng = np.random.RandomState(7)
        for case in range(n_cases):
            layers = 16 if case % 3 else int(rng.choice([9, 12, 16]))
            nw = 5 if case % 4 else 1
            sharp = float(rng.choice([0.5, 2.0, 5.0]))
            alphas = [torch.tensor(rng.randn(layers - s, 5) * sharp, dtype=torch.float32) for s in range(3)]
            betas = [None, torch.tensor(rng.randn(layers - 2, 2) * sharp, dtype=torch.float32),
                     torch.tensor(rng.randn(layers - 3, 2) * sharp, dtype=torch.float32)]
            ratios = [torch.tensor(rng.randn(layers - 1, nw), dtype=torch.float32),
                      torch.tensor(rng.randn(layers - 1, nw), dtype=torch.float32),
                      torch.tensor(rng.randn(layers - 2, nw), dtype=torch.float32)]
            ignore_skip = bool(case % 2)
            rec = {"layers": layers, "ignore_skip": ignore_skip,
                   "alphas": [a.tolist() for a in alphas], "betas": [None] + [b.tolist() for b in betas[1:]],
                   "ratios": [r.tolist() for r in ratios]}
            wml = WML if nw == 5 else ([1.] if ignore_skip else [4. / 12])
            a = [t.clone() for t in alphas]
            b = [None] + [t.clone() for t in betas[1:]]
            metas = []
            try:
                for last in (0, 1, 2):    # same order + shared mutation as model_seg.py:198-200
                    ops, path, downs, widths = model_seg.network_metas(a, b, ratios, wml, layers, last,
                                                                       ignore_skip=ignore_skip)
                    metas.append({"ops": [int(o) for o in ops], "path": [int(v) for v in path],
                                  "downs": [int(v) for v in downs], "widths": [float(w) for w in widths]})
                rec["metas"] = metas
            except AssertionError:
                rec["metas"] = None
                rec["partial"] = metas
            out.append(rec)
    with open(os.path.join(GOLD, "decode_cases.json"), "w") as f:
        json.dump(out, f)
    print("decode cases:", len(out), "asserting:", sum(1 for r in out if r["metas"] is None))


# ------------------------------------------------------------------------------------------------
# 3. per-operator goldens (eval fwd, train fwd+bwd, slimmable)
# ------------------------------------------------------------------------------------------------
def _run_case(module, x, seed, training, store, name):
    sd = _load_seeded(module, seed)
    SHAPES[name] = {k: list(v.shape) for k, v in sd.items()}
    module.train(training)
    x = x.clone().requires_grad_(training)
    y = module(x)
    store[name + "/y"] = _np(y)
    if training:
        gy = seeded_input(tuple(y.shape), seed + 17)
        (y * gy).sum().backward()
        store[name + "/gx"] = _np(x.grad)
        for k, p in module.named_parameters():
            if p.grad is not None:
                _put(store, name + "/g/" + k, p.grad)
        after = module.state_dict()
        for k in after:
            if k.endswith("running_mean") or k.endswith("running_var"):
                if not torch.equal(after[k], sd[k]):
                    store[name + "/s/" + k] = _np(after[k])


def gen_ops():
    store = {}
    index = []
    with ref_loader.reference("train"):
        import operations
        import seg_oprs
        import slimmable_ops
        from genotypes import PRIMITIVES
        seed = 100
        # non-slimmable primitives
        for kind in PRIMITIVES:
            for stride in (1, 2):
                for (n, cin, cout, h, w) in ((2, 16, 32, 10, 14), (1, 32, 32, 7, 14)):
                    if stride == 2 and (h % 2 or w % 2):
                        continue
                    if kind == "skip" and stride == 1 and cin != cout:
                        continue
                    for training in (False, True):
                        seed += 1
                        name = "%s_s%d_n%dc%dk%dh%dw%d_%s" % (kind, stride, n, cin, cout, h, w,
                                                                "train" if training else "eval")
                        m = operations.OPS[kind](cin, cout, stride, False, [1.])
                        _run_case(m, seeded_input((n, cin, h, w), seed), seed, training, store, name)
                        index.append({"name": name, "type": "primitive", "kind": kind, "stride": stride,
                                      "shape": [n, cin, h, w], "cout": cout, "training": training, "seed": seed,
                                      "slimmable": False})
        # slimmable primitives (train mode only: USBatchNorm2d has track_running_stats on the bank)
        for kind in PRIMITIVES:
            for stride in (1, 2):
                for ratio in ((8. / 12, 6. / 12), (4. / 12, 1.)):
                    seed += 1
                    cin, cout = 48, 48 * stride
                    m = operations.OPS[kind](cin, cout, stride, True, WML)
                    m.set_ratio(ratio)
                    cin_eff = slimmable_ops.make_divisible(cin * ratio[0])
                    name = "slim_%s_s%d_r%d_%d" % (kind, stride, round(ratio[0] * 12), round(ratio[1] * 12))
                    _run_case(m, seeded_input((2, cin_eff, 8, 12), seed), seed, True, store, name)
                    index.append({"name": name, "type": "primitive", "kind": kind, "stride": stride,
                                  "shape": [2, cin_eff, 8, 12], "cin_max": cin, "cout": cout, "training": True,
                                  "seed": seed, "slimmable": True, "ratio": list(ratio)})
        # ConvNorm / Head / FeatureFusion
        for (k, stride, pad, cin, cout, h, w) in ((3, 2, 1, 3, 16, 16, 24), (3, 1, 1, 48, 32, 9, 12),
                                                  (1, 1, 0, 64, 32, 6, 10), (3, 2, None, 16, 32, 12, 16)):
            for training in (False, True):
                seed += 1
                name = "convnorm_k%ds%d_c%dk%dh%dw%d_%s" % (k, stride, cin, cout, h, w, "train" if training else "eval")
                m = operations.ConvNorm(cin, cout, k, stride, pad, slimmable=False)
                _run_case(m, seeded_input((2, cin, h, w), seed), seed, training, store, name)
                index.append({"name": name, "type": "convnorm", "k": k, "stride": stride, "pad": pad,
                              "shape": [2, cin, h, w], "cout": cout, "training": training, "seed": seed})
        for (cin, h, w) in ((32, 8, 12), (320, 4, 6)):
            for training in ((False, True) if cin <= 256 else (False,)):
                seed += 1
                name = "head_c%dh%dw%d_%s" % (cin, h, w, "train" if training else "eval")
                m = seg_oprs.Head(cin, 19, True)
                _run_case(m, seeded_input((2, cin, h, w), seed), seed, training, store, name)
                index.append({"name": name, "type": "head", "shape": [2, cin, h, w], "training": training, "seed": seed})
        for training in (False, True):
            seed += 1
            name = "ffm_c64_%s" % ("train" if training else "eval")
            m = seg_oprs.FeatureFusion(64, 64, reduction=1, Fch=12, scale=8, branch=2)
            _run_case(m, seeded_input((2, 64, 8, 12), seed), seed, training, store, name)
            index.append({"name": name, "type": "ffm", "shape": [2, 64, 8, 12], "training": training, "seed": seed})
    np.savez_compressed(os.path.join(GOLD, "ops.npz"), **store)
    for rec in index:
        rec["state_shapes"] = SHAPES[rec["name"]]
    with open(os.path.join(GOLD, "ops_index.json"), "w") as f:
        json.dump(index, f)
    print("op cases:", len(index), "arrays:", len(store))


# ------------------------------------------------------------------------------------------------
# 4. whole-network goldens (BASELINE config C1 shape and a train-mode step)
# ------------------------------------------------------------------------------------------------
def gen_nets():
    store = {}
    with ref_loader.reference("train") as wd:
        import model_seg
        for idx, shape in ((1, (1, 3, 128, 256)), (0, (1, 3, 64, 128))):
            state = torch.load(os.path.join(wd, "fasterseg", "arch_%d.pt" % idx), map_location="cpu",
                               weights_only=False)
            net = build_ref_net(model_seg, state, idx, [2, 1], False)
            _load_seeded(net, 12345)
            net.eval()
            x = seeded_input(shape, 5)
            with torch.no_grad():
                y = net(x)
            store["arch%d_eval/logits_sub" % idx] = _np(y[:, :, ::4, ::4])
            store["arch%d_eval/stats" % idx] = np.array([float(y.mean()), float(y.std()), float(y.abs().max()),
                                                         float(y.double().sum())])
            store["arch%d_eval/argmax_sub" % idx] = _np(y.argmax(1)[:, ::2, ::2]).astype(np.uint8)
            print("arch%d eval logits" % idx, tuple(y.shape), store["arch%d_eval/stats" % idx])
        # student train-mode forward/backward (3 heads) at a small crop.  Run in fp32 (what the reference does) AND
        # in fp64 with the same reference modules: batch-statistics BN on maps as small as 1x2 makes individual
        # gradient elements sensitive to 1e-7 rounding (ReLU-mask flips), so the fp64 run is the pin for gradients.
        state = torch.load(os.path.join(wd, "fasterseg", "arch_1.pt"), map_location="cpu", weights_only=False)
        for tag, dt in (("arch1_train", torch.float32), ("arch1_train64", torch.float64)):
            net = build_ref_net(model_seg, state, 1, [2, 1], True)
            _load_seeded(net, 12345)
            net = net.to(dt)
            net.train()
            x = seeded_input((2, 3, 128, 256), 6).to(dt).requires_grad_(True)
            p8, p16, p32 = net(x)
            loss = (p8 * seeded_input(tuple(p8.shape), 7).to(dt)).sum() \
                + 0.2 * (p16 * seeded_input(tuple(p16.shape), 8).to(dt)).sum() \
                + 0.2 * (p32 * seeded_input(tuple(p32.shape), 9).to(dt)).sum()
            loss.backward()
            store[tag + "/p8_sub"] = _np(p8[:, :, ::4, ::4]).astype(np.float32)
            store[tag + "/p16_sub"] = _np(p16[:, :, ::4, ::4]).astype(np.float32)
            store[tag + "/p32_sub"] = _np(p32[:, :, ::4, ::4]).astype(np.float32)
            store[tag + "/loss"] = np.array([float(loss.detach())])
            _put(store, tag + "/gx", x.grad.float())
            norms = {k: float(p.grad.norm()) for k, p in net.named_parameters() if p.grad is not None}
            for k in ("stem.0.conv.0.weight", "cells.3-0._op._op.conv1.weight", "heads8.conv_1x1.weight",
                      "heads8.conv_1x1.bias", "cells.9-0._op._op.bn2.weight", "refines32.0.conv.0.weight",
                      "ffm.conv_1x1.bn.bias", "heads16.conv_3x3.bn.bias", "stem.2.bn2.weight"):
                _put(store, tag + "/g/" + k, dict(net.named_parameters())[k].grad.float())
            with open(os.path.join(GOLD, tag + "_gradnorms.json"), "w") as f:
                json.dump(norms, f)
            print(tag, "loss", float(loss.detach()))
    np.savez_compressed(os.path.join(GOLD, "nets.npz"), **store)


# 4b. the student's KL-distillation train step as train/train.py:219-271 runs it: teacher (arch_0, eval) forward, student (arch_1, train,
#     3 heads) forward, ProbOhemCrossEntropy2d x3 + KLDivLoss, backward - reference modules and criteria in fp64, 2 x 3x256x512 (maps down
#     to 4x8 x 2 images = 64 samples per BatchNorm channel; the 128x256 fixture above normalises over 4 samples, where bf16 storage
#     noise is amplified without bound).  Pins the bf16 student step (tests/test_train_parity_gpu.py).
def gen_student_step():
    store = {}
    B, H, W = 2, 256, 512
    with ref_loader.reference("train") as wd:
        import model_seg
        tools = os.path.join(os.path.dirname(wd), "tools")
        sys.path.insert(0, tools)
        for m in [k for k in sys.modules if k.split(".")[0] in ("seg_opr", "engine")]:
            sys.modules.pop(m)
        from seg_opr.loss_opr import ProbOhemCrossEntropy2d
        nets = []
        for idx in (0, 1):
            state = torch.load(os.path.join(wd, "fasterseg", "arch_%d.pt" % idx), map_location="cpu", weights_only=False)
            net = build_ref_net(model_seg, state, idx, [2, 1], True)
            _load_seeded(net, 12345 + idx)
            nets.append(net.to(torch.float64))
        teacher, student = nets
        teacher.eval()
        student.train()
        x = seeded_input((B, 3, H, W), 61).to(torch.float64)
        g = torch.Generator().manual_seed(62)
        target = torch.randint(0, 19, (B, H, W), generator=g)
        target[torch.rand(B, H, W, generator=g) < 0.05] = 255
        crit = ProbOhemCrossEntropy2d(ignore_label=255, thresh=0.7, min_kept=int(B * H * W // 16), use_weight=False)   # train.py:62
        kl = torch.nn.KLDivLoss()                                                                                         # train.py:64
        with torch.no_grad():
            t_logits = teacher(x)
        p8, p16, p32 = student(x)
        loss = crit(p8, target.clone()) + 0.2 * crit(p16, target.clone()) + 0.2 * crit(p32, target.clone())
        loss = loss + kl(torch.nn.functional.softmax(p8, dim=1).log(), torch.nn.functional.softmax(t_logits, dim=1))
        loss.backward()
        store["target"] = _np(target).astype(np.uint8)
        store["loss"] = np.array([float(loss.detach())])
        store["p8_sub"] = _np(p8[:, :, ::8, ::8]).astype(np.float32)
        store["teacher_sub"] = _np(t_logits[:, :, ::8, ::8]).astype(np.float32)
        named = dict(student.named_parameters())
        norms = {k: float(p.grad.norm()) for k, p in named.items() if p.grad is not None}
        for k in ("stem.0.conv.0.weight", "cells.3-0._op._op.conv1.weight", "heads8.conv_1x1.weight", "heads8.conv_1x1.bias",
                  "cells.9-0._op._op.bn2.weight", "refines32.0.conv.0.weight", "ffm.conv_1x1.bn.bias", "heads16.conv_3x3.bn.bias",
                  "stem.2.bn2.weight", "cells.5-1._op._op.conv1.weight", "heads32.conv_3x3.conv.weight"):
            _put(store, "g/" + k, named[k].grad.float())
        with open(os.path.join(GOLD, "student_step_gradnorms.json"), "w") as f:
            json.dump(norms, f)
        print("student step loss", float(loss.detach()), "params with grad:", len(norms))
    np.savez_compressed(os.path.join(GOLD, "student_step.npz"), **store)


# ------------------------------------------------------------------------------------------------
# 5. the shipped latency LUT (key grammar + published per-op numbers)
# ------------------------------------------------------------------------------------------------
def gen_lut():
    table = np.load(os.path.join(ref_loader.REFERENCE_ROOT, "search", "latency_lookup_table.npy"),
                    allow_pickle=True).item()
    with open(os.path.join(GOLD, "latency_lut_1080ti.json"), "w") as f:
        json.dump({k: float(v) for k, v in sorted(table.items())}, f)
    print("LUT entries:", len(table))


# ------------------------------------------------------------------------------------------------
# 6. step glue: OHEM cross-entropy (tools/seg_opr/loss_opr.py) on seeded logits
# ------------------------------------------------------------------------------------------------
def gen_loss():
    store = {}
    with ref_loader.reference("train") as wd:
        tools = os.path.join(os.path.dirname(wd), "tools")       # train/train.py reaches it through config's add_path
        sys.path.insert(0, tools)
        for m in [k for k in sys.modules if k.split(".")[0] in ("seg_opr", "engine")]:
            sys.modules.pop(m)
        from seg_opr.loss_opr import ProbOhemCrossEntropy2d
        for i, (shape, thresh, min_kept) in enumerate((((2, 19, 16, 24), 0.7, 48), ((1, 19, 12, 20), 0.7, 200),
                                                       ((2, 19, 8, 8), 0.3, 16), ((1, 19, 8, 8), 0.7, 1000))):
            g = torch.Generator().manual_seed(50 + i)
            pred = (torch.randn(shape, generator=g) * 2).requires_grad_(True)
            target = torch.randint(0, 19, (shape[0], shape[2], shape[3]), generator=g)
            target[torch.rand(target.shape, generator=g) < 0.1] = 255
            crit = ProbOhemCrossEntropy2d(ignore_label=255, thresh=thresh, min_kept=min_kept, use_weight=False)
            loss = crit(pred, target.clone())
            loss.backward()
            store["ohem%d/pred" % i] = _np(pred)
            store["ohem%d/target" % i] = _np(target).astype(np.int64)
            store["ohem%d/cfg" % i] = np.array([thresh, min_kept])
            store["ohem%d/loss" % i] = np.array([float(loss.detach())])
            store["ohem%d/grad" % i] = _np(pred.grad)
    np.savez_compressed(os.path.join(GOLD, "loss.npz"), **store)
    print("loss cases:", [float(store["ohem%d/loss" % i][0]) for i in range(4)])


# ------------------------------------------------------------------------------------------------
# 7. supernet (search/model_search.py): eval forward, pretrain/search losses + gradients, latency model
# ------------------------------------------------------------------------------------------------
SUPERNET_CFG = dict(num_classes=19, layers=6, Fch=12, width_mult_list=WML, prun_modes=['max', 'arch_ratio'],
                    stem_head_width=[(1, 1), (8. / 12, 8. / 12)])


def gen_supernet():
    store, meta = {}, {}
    with ref_loader.reference("search"):
        import model_search
        real_cuda = torch.Tensor.cuda
        torch.Tensor.cuda = lambda self, *a, **k: self          # the reference hard-codes .cuda() (model_search.py:16,373-384)
        try:
            crit = torch.nn.CrossEntropyLoss(ignore_index=255)
            for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
                net = model_search.Network_Multi_Path(criterion=crit, **SUPERNET_CFG)
                sd = seeded_state(net.state_dict(), 777)
                for k in list(sd):                              # arch params: small random logits instead of the 1e-3 init
                    if k.split("_")[0] in ("alpha", "beta", "ratio"):
                        sd[k] = sd[k] * 5.0
                net.load_state_dict(sd)
                net = net.to(dt)
                if tag == "f32":
                    meta["state_shapes"] = {k: list(v.shape) for k, v in net.state_dict().items()}
                    meta["num_params"] = int(sum(p.numel() for p in net.parameters()))
                x = seeded_input((2, 3, 128, 256), 31).to(dt)
                g = torch.Generator().manual_seed(32)
                target = torch.randint(0, 19, (2, 16, 32), generator=g)
                target[torch.rand(2, 16, 32, generator=g) < 0.05] = 255
                if tag == "f32":
                    store["target"] = _np(target)
                    net.eval()
                    for idx in (0, 1):
                        net.arch_idx = idx
                        net.prun_mode = "max"
                        with torch.no_grad():
                            preds = net(x)
                        for i, pr in enumerate(preds):
                            store["eval_arch%d/pred%d_sub" % (idx, i)] = _np(pr[:, :, ::4, ::4]).astype(np.float32)
                net.train()
                for mode in ("pretrain", "search"):
                    net.zero_grad()
                    np.random.seed(5)
                    torch.manual_seed(6)
                    net.arch_idx = 0
                    loss = net._loss(x, target, mode == "pretrain")
                    loss.backward()
                    store["%s_%s/loss" % (mode, tag)] = np.array([float(loss.detach())])
                    norms = {k: float(p.grad.norm()) for k, p in net.named_parameters() if p.grad is not None}
                    with open(os.path.join(GOLD, "supernet_%s_%s_gradnorms.json" % (mode, tag)), "w") as f:
                        json.dump(norms, f)
                    for k in ("alpha_0_0", "alpha_1_1", "beta_1_1", "beta_1_2", "ratio_1_0", "ratio_1_2",
                              "cells.1.0._op._ops.3.conv1.weight", "cells.2.1.downsample._ops.0.conv2.weight",
                              "cells.3.2._op._ops.4.bn2.bn.4.weight", "stem.0.0.conv.0.weight", "head02.0.conv_1x1.bias"):
                        p_ = dict(net.named_parameters())[k]
                        if p_.grad is not None:
                            _put(store, "%s_%s/g/%s" % (mode, tag, k), p_.grad.float())
                    print("supernet", mode, tag, float(loss.detach()), "params with grad:", len(norms))
                if tag == "f32":
                    # differentiable latency model (reads search/latency_lookup_table.npy = the shipped 1080Ti LUT)
                    net.arch_idx = 1
                    net.prun_mode = None
                    for a, b, r in ((True, False, False), (False, True, False), (False, False, True), (True, True, True)):
                        net.zero_grad()
                        torch.manual_seed(9)
                        lat = net.forward_latency((3, 1024, 2048), alpha=a, beta=b, ratio=r)
                        lat.backward()
                        key = "lat_%d%d%d" % (a, b, r)
                        store[key + "/value"] = np.array([float(lat.detach())])
                        for k in ("alpha_1_0", "alpha_1_2", "beta_1_1", "ratio_1_1"):
                            gr = getattr(net, k).grad
                            store[key + "/g/" + k] = _np(gr) if gr is not None else np.zeros(1, np.float32)
                        print("supernet latency", key, float(lat.detach()))
        finally:
            torch.Tensor.cuda = real_cuda
    np.savez_compressed(os.path.join(GOLD, "supernet.npz"), **store)
    with open(os.path.join(GOLD, "supernet_meta.json"), "w") as f:
        json.dump(meta, f)


# 7b. the FULL-SIZE supernet of the benchmarked steps: F12.L16 (search/config_search.py:81-84) at the C3 map size (256x512:
#     32x64 ... 8x16, zoomed 4x8) and at the C5 map size (224x448: 28x56 / 14x28 / 7x14, zoomed 3x7 - odd sizes through H//2),
#     one image, fp64: loss, the set of parameters that receive gradients, gradient norms of a sample of them, a few gradients.
L16_CFG = dict(num_classes=19, layers=16, Fch=12, width_mult_list=WML, prun_modes=['max', 'arch_ratio'],
               stem_head_width=[(1, 1), (8. / 12, 8. / 12)])
L16_CASES = (("pretrain", (1, 3, 256, 512)), ("search", (1, 3, 224, 448)))
L16_GRADS = ("alpha_0_0", "alpha_1_1", "beta_1_1", "beta_1_2", "ratio_1_0", "ratio_1_2", "cells.1.0._op._ops.3.conv1.weight",
             "cells.9.1.downsample._ops.0.conv2.weight", "cells.13.2._op._ops.4.bn2.bn.4.weight", "cells.15.2._op._ops.2.conv1.weight",
             "cells.7.1._op._ops.1.conv1.weight", "stem.0.0.conv.0.weight", "head02.0.conv_1x1.bias", "refine32.0.1.conv.0.weight")


def l16_sampled(key):
    """Which parameters have their gradient norm in the fixture (all 40 k would be 2.5 MB of names): 1 in 16 by name hash, plus
    every parameter outside the cells."""
    import zlib
    return (not key.startswith("cells.")) or zlib.crc32(key.encode()) % 16 == 0


def gen_supernet_l16():
    import hashlib
    store, meta = {}, {}
    with ref_loader.reference("search"):
        import model_search
        real_cuda = torch.Tensor.cuda
        torch.Tensor.cuda = lambda self, *a, **k: self
        try:
            crit = torch.nn.CrossEntropyLoss(ignore_index=255)
            net = model_search.Network_Multi_Path(criterion=crit, **L16_CFG)
            sd = seeded_state(net.state_dict(), 778)
            for k in list(sd):
                if k.split("_")[0] in ("alpha", "beta", "ratio"):
                    sd[k] = sd[k] * 5.0
            net.load_state_dict(sd)
            net = net.to(torch.float64).train()
            meta["num_params"] = int(sum(p.numel() for p in net.parameters()))
            for mode, shape in L16_CASES:
                x = seeded_input(shape, 41).to(torch.float64)
                g = torch.Generator().manual_seed(42)
                target = torch.randint(0, 19, (shape[0], shape[2] // 8, shape[3] // 8), generator=g)
                target[torch.rand(target.shape, generator=g) < 0.05] = 255
                store[mode + "/target"] = _np(target)
                net.load_state_dict({k: v.to(torch.float64) if v.is_floating_point() else v for k, v in sd.items()})   # BN buffers back
                net.zero_grad()
                np.random.seed(5)
                torch.manual_seed(6)
                net.arch_idx = 0
                loss = net._loss(x, target, mode == "pretrain")
                loss.backward()
                store[mode + "/loss"] = np.array([float(loss.detach())])
                named = dict(net.named_parameters())
                with_grad = sorted(k for k, p in named.items() if p.grad is not None)
                is_arch = lambda k: k.split("_")[0] in ("alpha", "beta", "ratio")
                weights = [k for k in with_grad if not is_arch(k)]
                meta[mode] = {"shape": list(shape), "params_with_grad": len(with_grad),
                              "params_with_grad_sha1": hashlib.sha1("\n".join(with_grad).encode()).hexdigest(),
                              "weights_with_grad": len(weights),
                              "weights_with_grad_sha1": hashlib.sha1("\n".join(weights).encode()).hexdigest(),
                              "gradnorms": {k: float(named[k].grad.norm()) for k in with_grad if l16_sampled(k)}}
                for k in L16_GRADS:
                    if named[k].grad is not None:
                        _put(store, "%s/g/%s" % (mode, k), named[k].grad.float())
                print("supernet L16", mode, shape, float(loss.detach()), "params with grad:", len(with_grad), "norms kept:", len(meta[mode]["gradnorms"]))
        finally:
            torch.Tensor.cuda = real_cuda
    np.savez_compressed(os.path.join(GOLD, "supernet_l16.npz"), **store)
    with open(os.path.join(GOLD, "supernet_l16_meta.json"), "w") as f:
        json.dump(meta, f)


# 8. evaluation metrics (tools/seg_opr/metric.py is pure numpy and imports unmodified)
def gen_eval():
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_metric", os.path.join(ref_loader.REFERENCE_ROOT, "tools", "seg_opr", "metric.py"))
    metric = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(metric)
    rng = np.random.RandomState(11)
    store = {}
    for name, shape in (("a", (64, 96)), ("b", (128, 256))):
        gt = rng.randint(0, 19, size=shape).astype(np.int64)
        gt[rng.rand(*shape) < 0.07] = 255                       # Cityscapes ignore label
        pred = np.where(rng.rand(*shape) < 0.6, gt.clip(0, 18), rng.randint(0, 19, size=shape)).astype(np.uint8)
        hist, labeled, correct = metric.hist_info(19, pred, gt)
        iu, miou, miou_nb, acc = metric.compute_score(hist, correct, labeled)
        store[name + "/gt"], store[name + "/pred"] = gt.astype(np.uint8), pred
        store[name + "/hist"], store[name + "/counts"] = hist.astype(np.int64), np.array([labeled, correct], dtype=np.int64)
        store[name + "/iu"], store[name + "/scores"] = iu, np.array([miou, miou_nb, acc])
    np.savez_compressed(os.path.join(GOLD, "eval.npz"), **store)
    print("eval fixtures:", sorted(store))


# 7c. the weight-update loop of the search (search/train_search.py:244-250: zero_grad -> `_loss` -> backward ->
#     clip_grad_norm_(model.parameters(), 5) -> SGD(lr, momentum 0.9, weight decay 5e-4) -> zero_grad), three steps on one batch.
#     Two trajectories: "zeros" = the reference's pinned torch 1.1, whose zero_grad() leaves zero-filled .grad tensors behind, so a
#     parameter that was touched once keeps decaying / coasting on its momentum in steps that do not use it; "none" = today's torch
#     (zero_grad(set_to_none=True)): unused parameters are skipped.  The clip norm covers model.parameters(), i.e. alpha / beta /
#     ratio too; the norm over the network weights alone is stored beside it.
TRAJ_KEYS = ("stem.0.0.conv.0.weight", "stem.0.1.conv1.weight", "cells.0.0._op._ops.1.conv1.weight", "cells.1.0._op._ops.3.conv2.weight",
             "cells.1.1._op._ops.2.bn1.bn.4.weight", "cells.2.1.downsample._ops.0.conv2.weight", "cells.3.2._op._ops.4.bn2.bn.4.bias",
             "cells.4.1._op._ops.4.conv1.weight", "cells.5.0._op._ops.1.bn1.bn.0.weight", "refine32.0.0.conv.0.weight",
             "head0.0.conv_1x1.weight", "head02.0.conv_1x1.bias", "head12.0.conv_3x3.conv.weight")


def gen_optimizer_trajectory(steps=3):
    store = {}
    with ref_loader.reference("search"):
        import model_search
        real_cuda = torch.Tensor.cuda
        torch.Tensor.cuda = lambda self, *a, **k: self
        try:
            crit = torch.nn.CrossEntropyLoss(ignore_index=255)
            x = seeded_input((2, 3, 128, 256), 31)
            g = torch.Generator().manual_seed(32)
            target = torch.randint(0, 19, (2, 16, 32), generator=g)
            target[torch.rand(2, 16, 32, generator=g) < 0.05] = 255
            finals = {}
            for sem in ("zeros", "none", "none_fp32"):          # none_fp32: the reference's OWN fp32 run (its distance from fp64 = the yardstick)
                net = model_search.Network_Multi_Path(criterion=crit, **SUPERNET_CFG)
                sd = seeded_state(net.state_dict(), 777)
                for k in list(sd):
                    if k.split("_")[0] in ("alpha", "beta", "ratio"):
                        sd[k] = sd[k] * 5.0
                net.load_state_dict(sd)
                dt = torch.float32 if sem.endswith("fp32") else torch.float64
                net = net.to(dt).train()                   # fp64: the trajectory itself, not a rounding study
                net.arch_idx = 0
                parameters = []
                for part in (net.stem, net.cells, net.refine32, net.refine16, net.head0, net.head1, net.head2, net.head02, net.head12):
                    parameters += list(part.parameters())                                  # train_search.py:84-94
                optimizer = torch.optim.SGD(parameters, lr=2e-2, momentum=0.9, weight_decay=5e-4)   # config_search.py:57-58,90
                wid = {id(p) for p in parameters}
                names = dict(net.named_parameters())
                assert all(k in names for k in TRAJ_KEYS), [k for k in TRAJ_KEYS if k not in names]
                for step in range(steps):
                    np.random.seed(100 + step)
                    torch.manual_seed(200 + step)
                    optimizer.zero_grad(set_to_none=(sem != "zeros"))
                    loss = net._loss(x.to(dt), target, True)
                    loss.backward()
                    w_norm = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in net.parameters() if p.grad is not None and id(p) in wid))
                    total = torch.nn.utils.clip_grad_norm_(net.parameters(), 5)             # train_search.py:249
                    optimizer.step()
                    store["%s/loss%d" % (sem, step)] = np.array([float(loss.detach())])
                    store["%s/norm_all%d" % (sem, step)] = np.array([float(total)])
                    store["%s/norm_weights%d" % (sem, step)] = np.array([float(w_norm)])
                    store["%s/touched%d" % (sem, step)] = np.array([sum(1 for p in parameters if p.grad is not None and float(p.grad.abs().sum()) > 0)])
                    print("trajectory", sem, step, float(loss.detach()), float(total), float(w_norm))
                finals[sem] = {k: names[k].detach().double().clone() for k in TRAJ_KEYS}
                if not sem.endswith("fp32"):
                    for k in TRAJ_KEYS:
                        _put(store, "%s/final/%s" % (sem, k), names[k].detach().float())
            for k in TRAJ_KEYS:       # relative error of the reference's fp32 UPDATE (final - initial) against its fp64 one
                init = sd[k].double()
                du64, du32 = finals["none"][k] - init, finals["none_fp32"][k] - init
                store["ref_fp32_update_err/" + k] = np.array([float((du32 - du64).norm() / (du64.norm() + 1e-30))])
                print("reference fp32 vs fp64 update", k, float(store["ref_fp32_update_err/" + k][0]))
        finally:
            torch.Tensor.cuda = real_cuda
    np.savez_compressed(os.path.join(GOLD, "optimizer_trajectory.npz"), **store)


def main():
    os.makedirs(GOLD, exist_ok=True)
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    which = sys.argv[1:] or ["arch", "decode", "ops", "nets", "student_step", "lut", "loss", "supernet", "supernet_l16", "eval", "trajectory"]
    for w in which:
        {"arch": gen_arch, "decode": gen_decode_cases, "ops": gen_ops, "nets": gen_nets, "lut": gen_lut, "loss": gen_loss, "supernet": gen_supernet, "supernet_l16": gen_supernet_l16, "eval": gen_eval, "student_step": gen_student_step, "trajectory": gen_optimizer_trajectory}[w]()


if __name__ == "__main__":
    main()
