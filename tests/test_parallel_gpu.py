"""The data-parallel path with the PRODUCT modules (no CPU path exists for them), two ranks on one GPU over gloo
(FS_DIST_BACKEND=gloo: RCCL refuses two ranks on one device; the collective semantics - all-reduce of the flat gradient buffer
before clip / SGD, architecture-gradient all-reduce inside Architect.step - are backend independent).

  * student distillation step (FlatGradientSync + FlatSGD, [O][R][S][I] gradient views, wgrad kernels accumulating into the
    flat buffer, 5 overlapped buckets): replicas stay bit-identical over 3 steps; the all-reduced step-0 gradient equals the
    average of the two ranks' local gradients computed without any collective.
  * supernet search step (hipGraph-replayed passes + mark_touched, 4 accumulating backward passes, Architect(grad_sync)):
    network weights and architecture parameters identical on both ranks after 2 iterations."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _student_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from fasterseg_amd import _lib, train_step
    _lib.lib().fs_set_deterministic(1)       # ordered reductions: the collective-free reference below can be compared tightly
    st = train_step.StudentDistillStep(2, 128, 256, seed=12345 + 7 * rank)      # different init per rank: broadcast must fix it
    assert len(st.sync.buckets) > 1
    imgs, target = train_step.synthetic_batch(2, 128, 256, rank, "cuda")          # each rank its own shard
    # local gradient of this rank's shard, no collective: run the step's forward/backward by hand on a detached copy
    import copy
    from fasterseg_amd.losses import distill_kl_lowres, ohem_ce_lowres
    ref = copy.deepcopy(st.student)
    with torch.no_grad():
        t_logits = st.teacher_logits(imgs)
    p8, p16, p32 = ref.forward_lowres(imgs)
    loss = ohem_ce_lowres(st.ohem, p8, target) + 0.2 * ohem_ce_lowres(st.ohem, p16, target) + 0.2 * ohem_ce_lowres(st.ohem, p32, target)
    (loss + distill_kl_lowres(p8, t_logits, (128, 256))).backward()
    local = {k: (None if p.grad is None else p.grad.detach().float().cpu().clone()) for k, p in ref.named_parameters()}
    # BN running statistics were updated by the hand-run: give the real student the same history before the real step
    st.student.load_state_dict(ref.state_dict(), strict=True)
    for k, p in ref.named_parameters():
        dict(st.student.named_parameters())[k].data.copy_(p.data)
    st.sync.prepare()
    t_logits = st.teacher_logits(imgs)
    p8, p16, p32 = st.student.forward_lowres(imgs)
    loss = ohem_ce_lowres(st.ohem, p8, target) + 0.2 * ohem_ce_lowres(st.ohem, p16, target) + 0.2 * ohem_ce_lowres(st.ohem, p32, target)
    (loss + distill_kl_lowres(p8, t_logits, (128, 256))).backward()
    st.sync.sync()
    # average="defer": the buffer holds the SUM over ranks and sync.grad_scale = 1 / world, which FlatSGD folds into its gradient read
    assert st.sync.grad_scale == 0.5
    synced = {k: (None if p.grad is None else (p.grad.detach().float() * st.sync.grad_scale).cpu().clone()) for k, p in st.student.named_parameters()}
    probe = [k for k, p in st.student.named_parameters() if p.grad is not None and p.dim() in (1, 4)][:40:13]
    before = {k: dict(st.student.named_parameters())[k].detach().float().cpu().clone() for k in probe}
    st.optimizer.step()
    after = {k: dict(st.student.named_parameters())[k].detach().float().cpu().clone() for k in probe}
    for _ in range(2):
        st.step(imgs, target)
    torch.cuda.synchronize()
    torch.save({"local": local, "synced": synced, "state": {k: v.detach().cpu() for k, v in st.student.state_dict().items()},
                "before": before, "after": after, "hyper": (st.optimizer.lr, st.optimizer.weight_decay)},
               "%s.r%d" % (out, rank))
    dist.destroy_process_group()


def test_student_step_data_parallel_two_ranks_one_gpu(tmp_path):
    out = str(tmp_path / "student")
    mp.spawn(_student_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r0, r1 = torch.load(out + ".r0"), torch.load(out + ".r1")
    n, worst = 0, []
    for k in r0["synced"]:
        a, b = r0["local"][k], r1["local"][k]
        if a is None and b is None:
            assert r0["synced"][k] is None
            continue
        want = (a + b) / 2
        got = r0["synced"][k]
        assert torch.equal(got, r1["synced"][k]), "ranks disagree on the all-reduced gradient of " + k
        rel = float((got - want).abs().max()) / (float(want.abs().max()) + 1e-12)
        worst.append((rel, k))
        n += 1
    worst.sort(reverse=True)
    # value check against a collective-free computation, in the library's bit-reproducible mode (round 2 ran this with float atomics
    # in the BN statistics / wgrad slabs: two runs of one step then differ at the 1e-4 level, which batch-2 BatchNorm on 4x8 maps
    # amplifies to several per cent in a few tensors - the bar had to be 8e-2): a sum-instead-of-mean or a missed bucket is off by 50-100 %.
    assert worst[0][0] <= 1e-4, worst[:8]
    assert n > 100
    # the first SGD step applied the AVERAGE (momentum buffer starts at zero: w1 = w0 - lr * (g_avg + wd * w0)); a sum instead of the mean
    # would double the update
    lr, wd = r0["hyper"]
    assert len(r0["before"]) >= 2
    for k in r0["before"]:
        w0, w1 = r0["before"][k], r0["after"][k]
        want = w0 - lr * (r0["synced"][k] + wd * w0)
        assert float((w1 - want).abs().max()) <= 1e-6 + 1e-4 * float((w1 - w0).abs().max()), k
    for k in r0["state"]:
        if k.endswith(("running_mean", "running_var", "num_batches_tracked")):
            continue                                  # BatchNorm statistics stay per rank by design (no SyncBN in the reference)
        assert torch.equal(r0["state"][k], r1["state"][k]), "replicas diverged: " + k


def _search_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from fasterseg_amd import latency_lookup_table, train_step

    class Cfg(train_step.SearchConfig):
        layers = 5                                    # same code path as F12.L16, a quarter of the cells
    st = train_step.SupernetStep(pretrain=False, cfg=Cfg, lut=latency_lookup_table.load_shipped("bf16"), seed=999 + rank)
    g = torch.Generator().manual_seed(50 + rank)

    def make():
        imgs = torch.randn(2, 3, 64, 128, generator=g).cuda()
        return imgs, torch.randint(0, 19, (2, 8, 16), generator=g).cuda()
    (imgs, target), (imgs_s, target_s) = make(), make()
    for _ in range(2):
        st.step(imgs, target, imgs_s, target_s)
    torch.cuda.synchronize()
    torch.save({k: v.detach().cpu() for k, v in st.model.state_dict().items()}, "%s.r%d" % (out, rank))
    dist.destroy_process_group()


def test_search_step_data_parallel_two_ranks_one_gpu(tmp_path):
    out = str(tmp_path / "search")
    mp.spawn(_search_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r0, r1 = torch.load(out + ".r0"), torch.load(out + ".r1")
    checked = 0
    for k in r0:
        if k.endswith(("running_mean", "running_var", "num_batches_tracked")):
            continue
        assert torch.equal(r0[k], r1[k]), "replicas diverged: " + k
        checked += 1
    assert checked > 1000 and any(k.startswith("alpha_") for k in r0)


def test_bench_two_ranks_over_gloo_runs_the_supernet_step(tmp_path):
    """The same launch for the supernet pretrain step: F12.L16 on two ranks - one flat broadcast per dtype of the ~70 k
    parameter / buffer tensors, rank 0's width-sampling seed on both ranks, four accumulating passes (two replayed from hipGraphs:
    mark_touched), every bucket all-reduced in sync(), the average folded into the SGD kernel's clip scale, the post-timed check on every
    rank - gated on rank 0 against the CPU oracle like the single-GPU run."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FS_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--workloads", "c3", "--steps", "50", "--warmup", "5", "--train-steps", "2",
           "--train-warmup", "1", "--min-seconds", "0.05", "--no-cpu-baseline", "--no-class-map", "--no-fp32-leg", "--detail", str(tmp_path / "detail.json")]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1 and len(lines[0]) < 4096
    d = json.loads(lines[0])
    w = d["workloads"]["C3_supernet_pretrain"]
    assert d["n_gpus"] == 2 and w["parity"]["pass"] and w["parity"]["rel_err"] <= 1e-2 and w["parity"]["after_timed"] is True
    with open(tmp_path / "detail.json") as f:
        c3 = json.load(f)["C3_supernet_pretrain"]
    assert c3["global_batch"] == 6 and c3["per_gpu_batch"] == 3
    assert "dp2" in c3["config"]["parallelism"] and "gloo" in c3["config"]["parallelism"]
    assert c3["post_timed_check"]["grads_finite"] and c3["post_timed_check"]["weights_finite"]
    assert c3["roofline"]["frac"] > 0 and c3["execution"]["graphed"] == 2


def test_bench_two_ranks_over_gloo_runs_the_collective_path(tmp_path):
    """`python bench.py --gpus 2` as the driver launches it (self-spawned torch.distributed.run, one rank per process), with both ranks
    on this box's one GPU over gloo (RCCL refuses two ranks on a device): the data-parallel train path, the parity gate and the timed
    census (every rank must issue the census step - it contains the gradient all-reduce) all run; the line says what it is."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FS_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--workloads", "c4", "--steps", "50", "--warmup", "5", "--train-steps", "2",
           "--train-warmup", "1", "--min-seconds", "0.05", "--no-cpu-baseline", "--no-class-map", "--no-fp32-leg", "--detail", str(tmp_path / "detail.json")]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, "exactly one JSON line (rank 0)"
    assert len(lines[0]) < 4096                      # the driver could not parse round 3's 24 KB line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 50
    assert d["workloads"]["C4_student_train"]["parity"]["pass"] and d["workloads"]["C4_student_train"]["roofline"]["frac"] > 0
    with open(tmp_path / "detail.json") as f:        # the full objects (method strings, per-kernel tables) live in the detail file
        c4 = json.load(f)["C4_student_train"]
    assert c4["global_batch"] == 24 and c4["parity"]["pass"]
    assert "functional test only" in c4["config"]["parallelism"] and "gloo" in c4["config"]["parallelism"]
    assert c4["roofline"]["flops_coverage"] == 1.0 and c4["roofline"]["frac"] > 0
