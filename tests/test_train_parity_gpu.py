"""Parity of the train paths bench.py times, at the benchmarked size and in the benchmarked dtype.

* F12.L16 supernet (search/config_search.py:81-84) `_loss` forward + backward at the C3 map size (1x3x256x512, pretrain passes)
  and the C5 map size (1x3x224x448, search passes: 28x56 / 14x28 / 7x14 maps and the zoomed 3x7 ones) against fixtures the
  UNMODIFIED reference produced in fp64 (tests/golden/supernet_l16*, oracle/make_golden.py gen_supernet_l16) - through the bare
  modules (fp32) and through train_step.SupernetStep exactly as bench.py builds it (hipGraph replay of the fixed-width passes,
  MixedOp launch programs, pair batching, flat-gradient sink), fp32 AND bf16.
* the student's KL-distillation step (train/train.py:246-262: frozen teacher, 3 student heads, OHEM-CE + KLDiv) through
  train_step.StudentDistillStep as bench.py builds it, fp32 AND bf16, against the reference's fp64 run at 2x3x256x512
  (tests/golden/student_step*, oracle/make_golden.py gen_student_step).

Bars.  fp32 (exact-fp32 MFMA, fp32 storage): loss 2e-3 relative, gradient norms and per-tensor relative L2 5e-2 (measured: loss exact to
7 digits, per-tensor relative L2 <= 2e-2, i.e. the reference's own fp32-vs-fp64 difference: batch-statistics BN over a few dozen
samples amplifies a 1e-7 rounding to 1e-2 in the gradients of deep layers - an amplification of ~1e5).
bf16 (bf16 storage of every activation and activation gradient, bf16 MFMA operands, fp32 accumulation / statistics / master weights /
parameter gradients): the LOSS is within 1e-3 of the fp64 reference (bar 5e-3) and the same parameters receive gradients, but the
same ~1e5 amplification acts on the 4e-3 storage rounding: gradients of tensors next to the heads agree (cosine >= 0.99), those of
deep layers only in direction (measured cosine 0.5 - 0.9, norms within 40 %) - on these batch-1 fixtures a bf16 gradient is as far
from the fp64 one as a gradient of another mini-batch would be.  That is a property of bf16 storage under tiny-batch BatchNorm, not
of a kernel (every kernel is pinned per operator in bf16 in tests/test_kernels_gpu.py / test_ops_gpu.py); it is why bench.py prints
an fp32 leg for every train workload and why the bars below for bf16 gradients are sanity bars (cosine >= 0.25 - three runs measured
0.50 - 0.60 for the worst tensor - and norms within 2x for 95 % of the sampled tensors), not parity bars.  Measured values are written to gpurun_out/parity_metrics.json and quoted in DESIGN.md.
"""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

from oracle.seeded import seeded_input, seeded_state
from tests._util import golden_get, load_json, load_npz

pytestmark = pytest.mark.gpu
WML = [4. / 12, 6. / 12, 8. / 12, 10. / 12, 1.]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = {"pretrain": (1, 3, 256, 512), "search": (1, 3, 224, 448)}


def _record(name, metrics):
    out = os.path.join(ROOT, "gpurun_out")
    if not os.path.isdir(out):
        return
    path = os.path.join(out, "parity_metrics.json")
    try:
        with open(path) as f:
            data = json.load(f)
    except Exception:
        data = {}
    data[name] = metrics
    with open(path, "w") as f:
        json.dump(data, f, indent=1)


def _l16_state(net):
    sd = seeded_state(net.state_dict(), 778)
    for k in list(sd):
        if k.split("_")[0] in ("alpha", "beta", "ratio"):
            sd[k] = sd[k] * 5.0
    return sd


def _batch(mode):
    shape = CASES[mode]
    store = load_npz("supernet_l16.npz")
    return seeded_input(shape, 41).cuda(), torch.tensor(store[mode + "/target"]).cuda()


def _cos_rel(got, store, key):
    want, step = golden_get(store, key)
    g = got.detach().float().cpu().numpy().reshape(-1)[::step].astype(np.float64)
    w = want.reshape(-1).astype(np.float64)
    cos = float((g * w).sum() / (np.sqrt((g * g).sum() * (w * w).sum()) + 1e-300))
    rel = float(np.sqrt(((g - w) ** 2).sum()) / (np.sqrt((w * w).sum()) + 1e-300))
    return cos, rel


def _check(name, loss, params, mode, dtype, weights_only):
    store = load_npz("supernet_l16.npz")
    meta = load_json("supernet_l16_meta.json")[mode]
    bf16 = dtype == torch.bfloat16
    want = float(store[mode + "/loss"][0])
    loss_rel = abs(loss - want) / abs(want)
    is_arch = lambda k: k.split("_")[0] in ("alpha", "beta", "ratio")
    got_names = sorted(k for k, p in params.items() if p.grad is not None and not (weights_only and is_arch(k)))
    tag = "weights_with_grad" if weights_only else "params_with_grad"
    norm_bar = 1.0 if bf16 else 5e-2
    bad, worst = [], 0.0
    sampled = {k: w for k, w in meta["gradnorms"].items() if not (weights_only and is_arch(k))}
    for k, w in sampled.items():
        g = float(params[k].grad.float().norm())
        err = abs(g - w) / (w + 1e-12)
        worst = max(worst, err)
        if abs(g - w) > norm_bar * w + 1e-6:
            bad.append((k, g, w))
    cos_min, rel_max, per = 1.0, 0.0, {}
    for key in store:
        if key.startswith(mode + "/g/"):
            pname = key[len(mode + "/g/"):].split("@")[0]
            if params[pname].grad is None:
                continue
            cos, rel = _cos_rel(params[pname].grad, store, mode + "/g/" + pname)
            per[pname] = (round(cos, 5), round(rel, 5))
            cos_min, rel_max = min(cos_min, cos), max(rel_max, rel)
    _record(name, dict(loss=loss, want=want, loss_rel=loss_rel, n_with_grad=len(got_names), norms_checked=len(sampled), norms_missed=len(bad),
                       worst_norm_err=worst, cos_min=cos_min, rel_l2_max=rel_max, per_tensor=per))
    assert loss_rel <= (5e-3 if bf16 else 2e-3), (loss, want)
    assert len(got_names) == meta[tag] and hashlib.sha1("\n".join(got_names).encode()).hexdigest() == meta[tag + "_sha1"], \
        "a different set of parameters received gradients (%d vs %d)" % (len(got_names), meta[tag])
    assert len(bad) <= (len(sampled) // 20 if bf16 else len(sampled) // 100), bad[:8]
    if bf16:
        near_head = [v for k, v in per.items() if k.startswith("head")]
        assert cos_min >= 0.25 and all(c >= 0.99 for c, _ in near_head), per
    else:
        assert cos_min >= 0.999 and rel_max <= 5e-2, per


@pytest.mark.parametrize("mode", ["pretrain", "search"])
def test_l16_supernet_modules_fp32(mode):
    """The bare modules (per-module autograd path, architecture parameters differentiated too) at full depth / real map sizes."""
    from fasterseg_amd import model_search
    net = model_search.Network_Multi_Path(19, 16, torch.nn.CrossEntropyLoss(ignore_index=255), 12, WML, ['max', 'arch_ratio'],
                                          [(1, 1), (8. / 12, 8. / 12)])
    net.load_state_dict(_l16_state(net))
    net = net.cuda().train()
    x, target = _batch(mode)
    np.random.seed(5)
    torch.manual_seed(6)
    net.arch_idx = 0
    loss = net._loss(x, target, mode == "pretrain")
    loss.backward()
    _check("l16_modules_fp32_" + mode, float(loss.detach()), dict(net.named_parameters()), mode, torch.float32, weights_only=False)


class _FrozenLR:
    """train_step.SearchConfig with lr = 0: the step runs its optimizer kernel but leaves the weights where the fixture has them."""
    lr = 0.0
    momentum = 0.9
    weight_decay = 5e-4
    grad_clip = 5
    arch_learning_rate = 3e-4
    layers = 16
    Fch = 12
    width_mult_list = WML
    prun_modes = ['max', 'arch_ratio']
    stem_head_width = [(1, 1), (8. / 12, 8. / 12)]
    latency_weight = [0, 1e-2]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("mode", ["pretrain", "search"])
def test_l16_supernet_step_as_benchmarked(mode, dtype):
    """train_step.SupernetStep as bench.py runs it (graphs + programs + pair batching + flat-gradient sink, compute dtype), one
    weight step of `_loss` on the fixture's weights and batch: loss and gradients vs the reference's fp64 run."""
    from fasterseg_amd.train_step import SupernetStep
    st = SupernetStep(pretrain=(mode == "pretrain"), cfg=_FrozenLR, compute_dtype=dtype)
    st.architect = None                                   # the fixture is `_loss` of the weight step (train_search.py:246-250)
    st._prewarmed = True                                  # (the program prewarm after a first step re-arms the gradient buffer)
    st.model.load_state_dict({k: v.cuda() for k, v in _l16_state(st.model).items()})
    st.optimizer.refresh_packs()
    x, target = _batch(mode)
    np.random.seed(5)
    torch.manual_seed(6)
    st.model.arch_idx = 0
    loss, _ = st.step(x, target)
    torch.cuda.synchronize()
    assert st.graphs, "the fixed-width passes were not captured"
    _check("l16_step_%s_%s" % ("bf16" if dtype == torch.bfloat16 else "fp32", mode), float(loss), dict(st.model.named_parameters()), mode,
           dtype, weights_only=True)


def test_l16_supernet_step_is_reproducible_in_loss():
    """Two fresh builds of the bf16 pretrain step on the same weights / batch / seeds give the same loss to 1e-3 relative (the
    remaining run-to-run differences are float atomics in BN statistics of maps > 512 px and in the weight-gradient slabs)."""
    from fasterseg_amd.train_step import SupernetStep
    out = []
    for _ in range(2):
        st = SupernetStep(pretrain=True, cfg=_FrozenLR, compute_dtype=torch.bfloat16)
        st._prewarmed = True
        st.model.load_state_dict({k: v.cuda() for k, v in _l16_state(st.model).items()})
        st.optimizer.refresh_packs()
        x, target = _batch("pretrain")
        np.random.seed(5)
        torch.manual_seed(6)
        out.append(float(st.step(x, target)[0]))
        del st
        torch.cuda.empty_cache()
    assert abs(out[0] - out[1]) <= 1e-3 * abs(out[0]), out


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_student_distill_step_as_benchmarked(dtype):
    """train_step.StudentDistillStep as bench.py's C4 runs it (frozen teacher through the engine, student train forward with three
    heads, OHEM-CE + KLDiv evaluated from the low-resolution logits, flat-gradient sink, compute dtype) on the fixture's weights and
    batch, 2 x 3x256x512: loss and gradients against the reference's fp64 run of train/train.py:246-262 (tests/golden/student_step*).
    (The 128x256 fixture of test_ops_gpu normalises 4 samples per channel in its coarsest maps: fine for fp32, meaningless for bf16.)"""
    from fasterseg_amd.engine import InferenceEngine
    from fasterseg_amd.train_step import StudentDistillStep
    store = load_npz("student_step.npz")
    norms = load_json("student_step_gradnorms.json")
    bf16 = dtype == torch.bfloat16
    B, H, W = 2, 256, 512
    st = StudentDistillStep(B, H, W, lr=0.0, teacher_engine_dtype=None, compute_dtype=dtype)
    st.teacher.load_state_dict({k: v.cuda() for k, v in seeded_state(st.teacher.state_dict(), 12345).items()})
    st.student.load_state_dict({k: v.cuda() for k, v in seeded_state(st.student.state_dict(), 12346).items()})
    st.optimizer.refresh_packs()
    st.teacher_engine = InferenceEngine(st.teacher, (B, 3, H, W), dtype=dtype, output="lowres" if st.fused_loss else "logits")
    x = seeded_input((B, 3, H, W), 61).cuda()
    target = torch.tensor(store["target"].astype(np.int64)).cuda()
    loss = float(st.step(x, target))
    torch.cuda.synchronize()
    want = float(store["loss"][0])
    params = dict(st.student.named_parameters())
    assert set(norms) == {k for k, p in params.items() if p.grad is not None}, "same set of parameters receives gradients"
    errs = {k: abs(float(params[k].grad.float().norm()) - w) / (w + 1e-12) for k, w in norms.items()}
    cos_min, rel_max, per = 1.0, 0.0, {}
    for key in store:
        if key.startswith("g/"):
            pname = key[2:].split("@")[0]
            cos, rel = _cos_rel(params[pname].grad, store, "g/" + pname)
            per[pname] = (round(cos, 5), round(rel, 5))
            cos_min, rel_max = min(cos_min, cos), max(rel_max, rel)
    bar = 1.0 if bf16 else 3e-2
    missed = sorted(((e, k) for k, e in errs.items() if e > bar), reverse=True)
    _record("student_step_" + ("bf16" if bf16 else "fp32"), dict(loss=loss, want=want, loss_rel=abs(loss - want) / want, worst_norm_err=max(errs.values()),
                                                              norms_missed=len(missed), cos_min=cos_min, rel_l2_max=rel_max, per_tensor=per))
    assert abs(loss - want) <= (5e-3 if bf16 else 2e-3) * want, (loss, want)
    assert len(missed) <= (len(norms) // 20 if bf16 else 0), missed[:8]
    if bf16:
        assert cos_min >= 0.25 and all(c >= 0.99 for k, (c, _) in per.items() if k.startswith("heads8.conv_1x1")), per
    else:
        assert cos_min >= 0.999 and rel_max <= bar, per


class _SmallSearch:
    """The 6-layer supernet of the reference fixtures with the weight step's settings (config_search.py:57-58,71,90)."""
    lr = 2e-2
    momentum = 0.9
    weight_decay = 5e-4
    grad_clip = 5
    arch_learning_rate = 3e-4
    layers = 6
    Fch = 12
    width_mult_list = WML
    prun_modes = ['max', 'arch_ratio']
    stem_head_width = [(1, 1), (8. / 12, 8. / 12)]
    latency_weight = [0, 1e-2]


@pytest.mark.parametrize("unused", ["decay", "skip"])
def test_weight_update_trajectory_matches_reference(unused):
    """Three iterations of the reference's weight-update loop (search/train_search.py:244-250: `_loss` -> backward ->
    clip_grad_norm_(model.parameters(), 5) -> SGD) on one batch, run by the UNMODIFIED reference in fp64 (oracle/make_golden.py
    `trajectory`), against SupernetStep + FlatSGD in fp32.  "decay" = the reference's pinned torch 1.1 (zero-filled gradients of unused
    parameters keep decaying them), "skip" = today's torch.  Checked: the three losses, the clip norm of every step, the final value of
    13 tensors.  The clip norm here covers the network weights only (architecture parameters are frozen in the weight step): the
    fixture holds both norms and the difference is asserted to be below 1e-3 relative - the deviation stated in DESIGN.md."""
    from oracle.seeded import seeded_input, seeded_state
    from fasterseg_amd.optim import FlatSGD
    from fasterseg_amd.train_step import SupernetStep
    store = load_npz("optimizer_trajectory.npz")
    sem = "zeros" if unused == "decay" else "none"
    st = SupernetStep(pretrain=True, cfg=_SmallSearch, compute_dtype=torch.float32, use_graphs=False)
    sd = seeded_state(st.model.state_dict(), 777)
    for k in list(sd):
        if k.split("_")[0] in ("alpha", "beta", "ratio"):
            sd[k] = sd[k] * 5.0
    st.model.load_state_dict({k: v.cuda() for k, v in sd.items()})
    st.optimizer = FlatSGD(st.sync, _SmallSearch.lr, _SmallSearch.momentum, _SmallSearch.weight_decay, max_norm=_SmallSearch.grad_clip,
                           pack_dtype=torch.float32, unused=unused)
    st._prewarmed = True
    x = seeded_input((2, 3, 128, 256), 31).cuda()
    g = torch.Generator().manual_seed(32)
    target = torch.randint(0, 19, (2, 16, 32), generator=g)
    target[torch.rand(2, 16, 32, generator=g) < 0.05] = 255
    target = target.cuda()
    st.model.arch_idx = 0
    for step in range(3):
        np.random.seed(100 + step)
        torch.manual_seed(200 + step)
        loss, _ = st.step(x, target)
        want = float(store["%s/loss%d" % (sem, step)][0])
        assert abs(float(loss) - want) <= 2e-3 * abs(want), (step, float(loss), want)
        norm_w, norm_all = float(store["%s/norm_weights%d" % (sem, step)][0]), float(store["%s/norm_all%d" % (sem, step)][0])
        assert abs(norm_all - norm_w) <= 1e-3 * norm_all              # what excluding alpha / beta / ratio from the clip norm changes
        assert abs(float(st.optimizer.last_norm) - norm_w) <= 2e-2 * norm_w, (step, float(st.optimizer.last_norm), norm_w)
    params = dict(st.model.named_parameters())
    worst = 0.0
    names = sorted({k[len(sem) + 7:].split("@")[0] for k in store.keys() if k.startswith(sem + "/final/")})
    assert len(names) == 13
    for name in names:
        want_np, stride = golden_get(store, "%s/final/%s" % (sem, name))            # big tensors are stored as flat[::stride]
        want = torch.from_numpy(np.asarray(want_np)).float().reshape(-1)
        got = params[name].detach().float().cpu().reshape(-1)[::stride]
        # compare the UPDATE (final - initial), which is what the optimizer computed
        init = sd[name].float().reshape(-1)[::stride]
        du_want, du_got = want - init, got - init
        rel = float((du_got - du_want).norm() / (du_want.norm() + 1e-12))
        worst = max(worst, rel)
        # yardstick: what the reference's own fp32 run of the same three steps loses against its fp64 run on this tensor (7 % for
        # the stem, 1e-3 next to the heads: tiny-batch BatchNorm amplifies fp32 rounding)
        ref32 = float(store["ref_fp32_update_err/" + name][0])
        assert rel <= 3.0 * ref32 + 2e-2, (name, rel, ref32)
    print("trajectory %s: worst relative error of an update vs the reference's fp64 run %.3e" % (unused, worst))
