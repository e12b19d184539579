"""Do small independent kernels on different HIP streams overlap on this device?  K convs (each on its own output) issued round-robin on
S streams, wall time per conv; eager launches and one captured multi-branch hipGraph.   python tools/stream_overlap.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fasterseg_amd import kernels as K  # noqa: E402

dtype = torch.bfloat16
NCONV = 48


def make(N, cin, cout, H, W):
    x = K.to_nhwc(torch.randn(N, cin, H, W, device="cuda"), dtype)
    w = K.pack_weight(torch.randn(cout, cin, 3, 3, device="cuda") * 0.05, dtype)
    outs = [K.empty_nhwc(N, cout, H, W, dtype, "cuda") for _ in range(NCONV)]
    return lambda i: K.conv2d(x, w, cout, 3, 3, 1, 1, out=outs[i], workspace=False)


def run(fn, nstreams, graph):
    main = torch.cuda.current_stream()
    lanes = [torch.cuda.Stream() for _ in range(nstreams)]

    def issue():
        for l in lanes:
            l.wait_stream(main)
        for i in range(NCONV):
            with torch.cuda.stream(lanes[i % nstreams]):
                fn(i)
        for l in lanes:
            main.wait_stream(l)
    for _ in range(3):
        issue()
    torch.cuda.synchronize()
    if graph:
        g = torch.cuda.CUDAGraph()
        cap = torch.cuda.Stream()
        with torch.cuda.stream(cap):
            main2 = torch.cuda.current_stream()
            with torch.cuda.graph(g, stream=cap):
                for l in lanes:
                    l.wait_stream(cap)
                for i in range(NCONV):
                    with torch.cuda.stream(lanes[i % nstreams]):
                        fn(i)
                for l in lanes:
                    cap.wait_stream(l)
        call = g.replay
    else:
        call = issue
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    R = 20
    for _ in range(R):
        call()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (R * NCONV) * 1e6


for shape in [(6, 192, 192, 16, 32), (6, 96, 96, 32, 64), (6, 384, 384, 8, 16), (6, 32, 32, 16, 32)]:
    fn = make(*shape)
    row = []
    for graph in (False, True):
        for ns in (1, 2, 4, 8):
            row.append("%s s%d %.2f" % ("graph" if graph else "eager", ns, run(fn, ns, graph)))
    print(shape, "us per conv:", " | ".join(row), flush=True)
