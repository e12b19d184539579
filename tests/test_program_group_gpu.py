"""Layer execution of MixedOp launch programs (fs_exec_program_group): the MixedOps of one layer replayed together - whatever their
structure (stride 1 / 2), map size and BatchNorm grouping - with the commands of one kind (conv -> BN units, convolutions, weight / data
gradients, BatchNorm passes, resamples, weighted sums) as ONE grouped launch each, against the same MixedOps replayed one program at a
time (fs_exec_program) - outputs, input gradients, coefficient gradients, the flat weight gradient and the BN running statistics."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu

WIDTHS = [4. / 12, 6. / 12, 8. / 12, 10. / 12, 1.]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("phase", ["w", "a"])
@pytest.mark.parametrize("kinds,sizes,groups", [
    ((1, 1, 1, 1), None, None),
    ((1, 2, 1, 2, 1), None, None),
    # three scales of a supernet layer at once: grid-wide BatchNorm passes (3072 / 768 px), the one-launch column kernels (192 / 48 px),
    # pair-batched cells (two independently normalised halves) beside plain ones, nine programs (more than one grouped launch holds)
    ((1, 2, 1, 2, 1, 1, 2, 1, 1), [(32, 48), (32, 48), (16, 24), (16, 24), (8, 12), (8, 16), (8, 16), (16, 24), (32, 48)], [1, 1, 2, 2, 2, 1, 1, 1, 2]),
], ids=["s1x4", "mixed", "scales"])
def test_lockstep_group_equals_single_programs(kinds, sizes, groups, phase, dtype):
    from fasterseg_amd import fusion, kernels as K, model_search
    from fasterseg_amd.parallel import FlatGradientSync
    torch.manual_seed(5)
    ops = torch.nn.ModuleList([model_search.MixedOp(48, 48 * s, stride=s, width_mult_list=WIDTHS) for s in kinds]).cuda().train()
    for p in ops.parameters():
        if p.dim() == 1:
            p.data.uniform_(0.5, 1.5)
    for m in ops:
        fusion.colocate(m)
    state0 = copy.deepcopy(ops.state_dict())
    sync = FlatGradientSync(fusion.flat_order(ops, ops.parameters()))
    ratios = [(WIDTHS[(k + 2) % 5], WIDTHS[(2 * k + 1) % 5]) for k in range(len(ops))]
    cins = [int(48 * r[0]) // 8 * 8 for r in ratios]
    sizes = sizes or [(16, 24)] * len(kinds)
    groups = groups or [1] * len(kinds)
    xs0, dys0, coefs0 = [], [], []
    for k, m in enumerate(ops):
        m.set_prun_ratio(ratios[k])
        cout, cin = m._ops[1].conv1.active_channels()
        s = kinds[k]
        h, w = sizes[k]
        xs0.append(K.to_nhwc(torch.randn(2, cin, h, w, device="cuda"), dtype))
        dys0.append(K.to_nhwc(torch.randn(2, cout, h // s, w // s, device="cuda"), dtype))
        coefs0.append(torch.softmax(torch.randn(5, device="cuda"), 0))
    got = []
    flag = model_search._GROUP_PROGRAMS
    try:
        for group in (False, True):
            model_search._GROUP_PROGRAMS = group
            ops.load_state_dict(state0)
            for p in ops.parameters():
                p.requires_grad_(phase == "w")
            if phase == "w":
                sync.prepare()
            xs = [x.clone().requires_grad_(True) for x in xs0]
            coefs = [c.clone().requires_grad_(phase == "a") for c in coefs0]
            outs = model_search._run_tasks([(m, x, c, r, g) for m, x, c, r, g in zip(ops, xs, coefs, ratios, groups)])
            names = {type(o.grad_fn).__name__ for o in outs}
            assert names == ({"_MixedOpProgramGroupBackward"} if group else {"_MixedOpProgramBackward"}), names
            torch.autograd.backward(outs, dys0)
            rec = {"out": torch.cat([o.detach().float().reshape(-1) for o in outs]), "dx": torch.cat([x.grad.float().reshape(-1) for x in xs])}
            if phase == "a":
                rec["dcoef"] = torch.cat([c.grad for c in coefs])
            else:
                sync.sync()
                rec["flat"] = sync.flat.clone()
                rec["touched"] = list(sync._touched)
            rec["running"] = torch.cat([b.float().reshape(-1) for n, b in ops.named_buffers()])
            got.append(rec)
    finally:
        model_search._GROUP_PROGRAMS = flag
        for p in ops.parameters():
            p.requires_grad_(True)
    ref, new = got
    tol = 3e-3 if dtype == torch.float32 else 3e-2
    for k in ref:
        if k == "touched":
            assert ref[k] == new[k]
            continue
        rel = float((ref[k] - new[k]).norm() / (ref[k].norm() + 1e-12))
        assert rel < tol, (k, rel)
