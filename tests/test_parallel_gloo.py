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
