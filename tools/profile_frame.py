"""C2 frames issued launch by launch in plan order on ONE stream (no hipGraph), for rocprofv3 --kernel-trace: the k-th kernel of a frame
in the trace is the k-th launch of the plan, so tools/roofline_from_profile.py can price every kernel family from the trace alone.

    FS_ENGINE_PLAN=<file> rocprofv3 --kernel-trace --output-format csv -d <dir> -o run -- python tools/profile_frame.py [frames] [plan.json]
"""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fasterseg_amd import archs, engine
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 50
net = archs.init_weight(archs.build_derived(1, training=False), seed=12345).cuda().eval()
eng = engine.InferenceEngine(net, (1, 3, 1024, 2048), dtype=torch.bfloat16, logits_dtype=torch.float32, use_graph=False)
eng.input.copy_(torch.randn(1, 3, 1024, 2048, device="cuda"))
for _ in range(5):
    eng.run()
torch.cuda.synchronize()
for _ in range(frames):
    eng.run()
torch.cuda.synchronize()
if len(sys.argv) > 2:
    with open(sys.argv[2], "w") as f:
        json.dump([dict(label=c["label"], family=c["family"], fn=c["fn"], flops=c["flops"], bytes=c["bytes"]) for c in eng.calls], f, indent=1)
print("PROFILE_FRAMES", frames, len(eng.calls))
