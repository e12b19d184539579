"""Pins oracle/ref_ops.py (the CPU restatement) against fixtures produced by the UNMODIFIED reference
(oracle/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import ref_ops
from oracle.seeded import resolve_aliases, seeded_input, seeded_state
from tests._util import assert_close_golden, load_json, load_npz, shapes_template

OPS_INDEX = load_json("ops_index.json")
WML = [4. / 12, 6. / 12, 8. / 12, 10. / 12, 1.]


def run_oracle_case(rec, params, x):
    c = ref_ops.Ctx(params, rec["training"], WML)
    if rec["type"] == "primitive":
        ratio = tuple(rec["ratio"]) if rec.get("slimmable") else None
        return c.primitive(rec["kind"], x, "", rec["stride"], ratio)
    if rec["type"] == "convnorm":
        return c.conv_norm(x, "", rec["k"], rec["stride"], rec["pad"])
    if rec["type"] == "head":
        return c.head(x, "")
    if rec["type"] == "ffm":
        return c.ffm(x, "")
    raise ValueError(rec["type"])


def strip(params):
    # oracle prefixes are "" + ".conv1.weight" -> keys with a leading dot
    return {"." + k: v for k, v in params.items()}


@pytest.mark.parametrize("rec", OPS_INDEX, ids=[r["name"] for r in OPS_INDEX])
def test_oracle_op_matches_reference(rec):
    store = load_npz("ops.npz")
    name = rec["name"]
    params = seeded_state(shapes_template(rec["state_shapes"]), rec["seed"])
    params = {k: v.clone().requires_grad_(rec["training"] and v.is_floating_point() and "running" not in k)
              for k, v in params.items()}
    x = seeded_input(tuple(rec["shape"]), rec["seed"]).requires_grad_(rec["training"])
    y = run_oracle_case(rec, strip(params), x)
    assert_close_golden(y, store, name + "/y", 1e-5, 1e-5, name)
    if rec["training"]:
        (y * seeded_input(tuple(y.shape), rec["seed"] + 17)).sum().backward()
        assert_close_golden(x.grad, store, name + "/gx", 2e-5, 1e-4, name)
        for k, p in params.items():
            gkey = name + "/g/" + k
            has = any(s == gkey or s.startswith(gkey + "@") for s in store)
            if has:
                assert p.grad is not None, k
                assert_close_golden(p.grad, store, gkey, 1e-4, 1e-4, name)
            skey = name + "/s/" + k
            if skey in store:
                assert_close_golden(p, store, skey, 1e-6, 1e-5, name)


@pytest.mark.parametrize("idx,shape,tag", [(1, (1, 3, 128, 256), "eval_21"), (0, (1, 3, 64, 128), "eval_21")])
def test_oracle_derived_net_eval(idx, shape, tag):
    meta = load_json("arch_%d.json" % idx)[tag]
    store = load_npz("nets.npz")
    params = resolve_aliases(seeded_state(shapes_template(meta["state_shapes"]), 12345), meta)
    x = seeded_input(shape, 5)
    with torch.no_grad():
        y = ref_ops.derived_forward(params, meta, x, training=False)
    assert tuple(y.shape) == (shape[0], 19, shape[2], shape[3])
    assert_close_golden(y[:, :, ::4, ::4], store, "arch%d_eval/logits_sub" % idx, 2e-4, 1e-4)
    stats = store["arch%d_eval/stats" % idx]
    assert abs(float(y.double().sum()) - stats[3]) <= 1e-4 * abs(stats[3]) + 1.0
    am = y.argmax(1)[:, ::2, ::2].numpy().astype(np.uint8)
    assert (am == store["arch%d_eval/argmax_sub" % idx]).mean() > 0.999


def test_oracle_student_train_step():
    meta = load_json("arch_1.json")["train_21"]
    store = load_npz("nets.npz")
    params = resolve_aliases(seeded_state(shapes_template(meta["state_shapes"]), 12345), meta)
    params = {k: v.requires_grad_(v.is_floating_point() and "running" not in k) for k, v in params.items()}
    x = seeded_input((2, 3, 128, 256), 6).requires_grad_(True)
    p8, p16, p32 = ref_ops.derived_forward(params, meta, x, training=True)
    loss = (p8 * seeded_input(tuple(p8.shape), 7)).sum() + 0.2 * (p16 * seeded_input(tuple(p16.shape), 8)).sum() \
        + 0.2 * (p32 * seeded_input(tuple(p32.shape), 9)).sum()
    loss.backward()
    assert abs(float(loss) - float(store["arch1_train/loss"][0])) < 2e-2
    assert_close_golden(p8[:, :, ::4, ::4], store, "arch1_train/p8_sub", 1e-3, 1e-3)
    assert_close_golden(x.grad, store, "arch1_train/gx", 2e-3 * float(x.grad.abs().max()), 2e-3)
    for k in store:
        if k.startswith("arch1_train/g/"):
            pname = k[len("arch1_train/g/"):].split("@")[0]
            assert_close_golden(params[pname].grad, store, "arch1_train/g/" + pname, 5e-3, 5e-3)


def test_oracle_ohem_matches_reference():
    """oracle/ref_loss.ohem_ce (the criterion of bench.py's C4 parity gate / cpu_baseline) vs the reference's ProbOhemCrossEntropy2d."""
    from oracle import ref_loss
    store = load_npz("loss.npz")
    for i in range(4):
        pred = torch.tensor(store["ohem%d/pred" % i]).requires_grad_(True)
        target = torch.tensor(store["ohem%d/target" % i])
        thresh, min_kept = store["ohem%d/cfg" % i]
        loss = ref_loss.ohem_ce(pred, target, 255, float(thresh), int(min_kept))
        loss.backward()
        assert abs(float(loss.detach()) - float(store["ohem%d/loss" % i][0])) < 1e-6, i
        np.testing.assert_allclose(pred.grad.numpy(), store["ohem%d/grad" % i], atol=1e-6)


def test_oracle_student_distill_step_matches_reference():
    """oracle/ref_ops (teacher eval + student train forward) + oracle/ref_loss.student_step_loss - the checker of bench.py's C4 parity gate
    and cpu_baseline - against the reference's own fp64 run of train/train.py:246-262 (fixture tests/golden/student_step.npz)."""
    from oracle import ref_loss
    store = load_npz("student_step.npz")
    metas = [load_json("arch_%d.json" % i)["train_21"] for i in (0, 1)]
    pt = resolve_aliases(seeded_state(shapes_template(metas[0]["state_shapes"]), 12345), metas[0])
    ps = resolve_aliases(seeded_state(shapes_template(metas[1]["state_shapes"]), 12346), metas[1])
    pt = {k: v.double() if v.is_floating_point() else v for k, v in pt.items()}
    ps = {k: v.double() if v.is_floating_point() else v for k, v in ps.items()}
    x = seeded_input((2, 3, 256, 512), 61).double()
    target = torch.tensor(store["target"].astype(np.int64))
    with torch.no_grad():
        t_logits = ref_ops.derived_forward(pt, metas[0], x, training=False)
        p8, p16, p32 = ref_ops.derived_forward(ps, metas[1], x, training=True)
        loss = float(ref_loss.student_step_loss(p8, p16, p32, t_logits, target, min_kept=2 * 256 * 512 // 16))
    assert_close_golden(t_logits[:, :, ::8, ::8], store, "teacher_sub", 1e-5, 1e-5)
    assert_close_golden(p8[:, :, ::8, ::8], store, "p8_sub", 1e-5, 1e-5)
    assert abs(loss - float(store["loss"][0])) <= 1e-7 * float(store["loss"][0]), (loss, float(store["loss"][0]))
