"""Latency harness: replaces the reference's TensorRT/PyCUDA timer (tools/utils/darts_utils.py:96-177) and its
PyTorch fallback (darts_utils.py:182-223) with hipEvent timing of the HIP kernels on the stream they run on.

Contract kept: `compute_latency(model, (1, C, H, W)) -> milliseconds per forward`, eval mode, random input,
10 warm-up runs, iteration count calibrated from a trial run, then one timed batch of iterations.  The forward is
captured once into a hipGraph (the analogue of building a TensorRT engine) so the number is device time, not Python
dispatch time; `graph=False` times eager launches like the reference's PyTorch fallback.
"""
import torch


def _time_ms(fn, iters):
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    for _ in range(iters):
        fn()
    stop.record()
    stop.synchronize()
    return start.elapsed_time(stop)


def compute_latency_ms_hip(model, input_size, iterations=None, device=None, graph=True, min_calib_ms=200.0, budget_ms=600.0):
    if not torch.cuda.is_available():
        raise RuntimeError("compute_latency_ms_hip needs an MI355X (no CPU fallback)")
    device = device or "cuda"
    was_training = model.training
    model.eval()
    model = model.to(device)
    x = torch.randn(*input_size, device=device)
    with torch.no_grad():
        for _ in range(3):
            model(x)
        run = lambda: model(x)
        if graph:
            # A table must not mix device-time entries with entries that include Python dispatch: if the forward cannot be
            # captured the measurement FAILS (ask for graph=False explicitly to get the eager number).
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            try:
                with torch.cuda.stream(side):
                    model(x)
                    with torch.cuda.graph(g, stream=side):
                        model(x)
            except Exception as e:
                torch.cuda.synchronize()
                raise RuntimeError("compute_latency_ms_hip: forward of %s is not hipGraph-capturable (%s); pass graph=False for an "
                                   "eager, dispatch-inclusive timing" % (type(model).__name__, e)) from e
            torch.cuda.current_stream().wait_stream(side)
            run = g.replay
        for _ in range(10):        # warm-up, darts_utils.py:141-142,194-195
            run()
        torch.cuda.synchronize()
        if iterations is None:     # calibrate like darts_utils.py:144-154 (grow until the trial is long enough)
            iterations, elapsed = 20, 0.0
            while True:
                elapsed = _time_ms(run, iterations)
                if elapsed >= min_calib_ms or iterations >= (1 << 16):
                    break
                iterations *= 2
            per = elapsed / iterations
            iterations = max(10, int(budget_ms / max(per, 1e-4)))
        latency = _time_ms(run, iterations) / iterations
    model.train(was_training)
    return latency


# names the reference imports (operations.py:24-29)
compute_latency_ms_tensorrt = compute_latency_ms_hip
compute_latency_ms_pytorch = compute_latency_ms_hip
