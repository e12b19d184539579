"""N>1 path on CPU: world_size-2 gloo processes check that FlatGradientSync reproduces the full-batch gradient, leaves
untouched parameters at grad=None, and keeps replicas identical through optimizer steps."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


class Toy(nn.Module):
    def __init__(self):
        super().__init__()
        self.a = nn.Linear(6, 8)
        self.b = nn.Linear(8, 3)
        self.unused = nn.Linear(4, 4)        # never part of the graph, like USBN's dead affine

    def forward(self, x):
        return self.b(torch.relu(self.a(x)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from fasterseg_amd.parallel import FlatGradientSync, broadcast_parameters
    torch.manual_seed(100 + rank)                       # different init per rank on purpose
    model = Toy()
    broadcast_parameters(model)
    sync = FlatGradientSync(model.parameters(), bucket_mb=0.0001)     # tiny buckets: exercise the multi-bucket path
    assert len(sync.buckets) > 1
    opt = torch.optim.SGD(model.parameters(), lr=0.1, momentum=0.9, weight_decay=5e-4)
    g = torch.Generator().manual_seed(7)
    X, Y = torch.randn(8, 6, generator=g), torch.randn(8, 3, generator=g)
    for step in range(3):
        sync.prepare()
        xs, ys = X[rank::world], Y[rank::world]            # shard the batch
        ((model(xs) - ys) ** 2).mean().backward()
        sync.sync()
        if step == 0:
            grads = {k: (None if p.grad is None else p.grad.clone()) for k, p in model.named_parameters()}
        opt.step()
    if rank == 0:
        torch.save({"grads": grads, "state": model.state_dict()}, out)
    else:
        torch.save(model.state_dict(), out + ".r1")
    dist.destroy_process_group()


def test_flat_gradient_sync_world2(tmp_path):
    out = str(tmp_path / "r0.pt")
    port = _free_port()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out)
    other = torch.load(out + ".r1")
    for k in got["state"]:
        assert torch.equal(got["state"][k], other[k]), "replicas diverged: " + k
    # single-process reference on the full batch
    torch.manual_seed(100)
    ref = Toy()
    g = torch.Generator().manual_seed(7)
    X, Y = torch.randn(8, 6, generator=g), torch.randn(8, 3, generator=g)
    ((ref(X) - Y) ** 2).mean().backward()
    for k, p in ref.named_parameters():
        if k.startswith("unused"):
            assert got["grads"][k] is None
        else:
            assert torch.allclose(got["grads"][k], p.grad, atol=1e-6), k


def _worker_two_pass(rank, world, port, out, comm_bf16):
    """Two accumulating backward passes before sync() (the supernet's `_loss`: several passes into one buffer).  Every
    bucket is completed by the FIRST pass here, which is exactly the case where an early all-reduce would be wrong."""
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from fasterseg_amd.parallel import FlatGradientSync, broadcast_parameters
    torch.manual_seed(100 + rank)
    model = Toy()
    broadcast_parameters(model)
    sync = FlatGradientSync(model.parameters(), bucket_mb=0.0001, comm_dtype=torch.bfloat16 if comm_bf16 else None)
    g = torch.Generator().manual_seed(7)
    X, Y = torch.randn(8, 6, generator=g), torch.randn(8, 3, generator=g)
    xs, ys = X[rank::world], Y[rank::world]
    sync.prepare(passes=2)
    ((model(xs) - ys) ** 2).mean().backward()
    (3.0 * (model(xs * 0.5) - ys) ** 2).mean().backward()
    assert not sync.handles, "a bucket was all-reduced before the last backward pass"
    sync.sync()
    if rank == 0:
        torch.save({k: (None if p.grad is None else p.grad.clone()) for k, p in model.named_parameters()}, out)
    dist.destroy_process_group()


def _two_pass_reference():
    torch.manual_seed(100)
    ref = Toy()
    g = torch.Generator().manual_seed(7)
    X, Y = torch.randn(8, 6, generator=g), torch.randn(8, 3, generator=g)
    (((ref(X) - Y) ** 2).mean() + (3.0 * (ref(X * 0.5) - Y) ** 2).mean()).backward()
    return ref


def test_flat_gradient_sync_two_backward_passes_world2(tmp_path):
    out = str(tmp_path / "g.pt")
    mp.spawn(_worker_two_pass, args=(2, _free_port(), out, False), nprocs=2, join=True)
    got = torch.load(out)
    for k, p in _two_pass_reference().named_parameters():
        if k.startswith("unused"):
            assert got[k] is None
        else:
            assert torch.allclose(got[k], p.grad, atol=1e-6), (k, float((got[k] - p.grad).abs().max()))


def test_flat_gradient_sync_bf16_buckets_world2(tmp_path):
    """comm_dtype=bf16: the all-reduced gradient equals the fp32 one up to bf16 rounding of each rank's contribution."""
    out = str(tmp_path / "g16.pt")
    mp.spawn(_worker_two_pass, args=(2, _free_port(), out, True), nprocs=2, join=True)
    got = torch.load(out)
    for k, p in _two_pass_reference().named_parameters():
        if not k.startswith("unused"):
            assert got[k].dtype == torch.float32
            assert float((got[k] - p.grad).abs().max()) <= 2 ** -7 * float(p.grad.abs().max()) + 1e-6, k
