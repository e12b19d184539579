"""Times fs_conv2d_wgrad_strided on the nine commonest C3 geometries (weighted by their launch counts per step); the data behind
the slab heuristic in csrc/wgrad.hip (FS_WGRAD_BLOCKS / FS_WGRAD_MIN_CHUNKS).  Run on an MI355X."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fasterseg_amd import kernels as K
from fasterseg_amd._lib import call, ConvDesc
from fasterseg_amd.census import _graph_time_ms
dt = torch.bfloat16
print("BLOCKS=%s MIN_CHUNKS=%s STORE=%s" % (os.environ.get("FS_WGRAD_BLOCKS"), os.environ.get("FS_WGRAD_MIN_CHUNKS"), os.environ.get("FS_WGRAD_DEBUG_STORE")))
tot = tot2 = 0
for (N, H, W, Ci, Co, k, s, cnt) in [(6, 16, 32, 192, 192, 3, 1, 300), (6, 8, 16, 384, 384, 3, 1, 250), (3, 32, 64, 96, 96, 3, 1, 150), (6, 8, 16, 192, 192, 3, 1, 400),
                                (6, 4, 8, 384, 384, 3, 1, 400), (3, 16, 32, 96, 96, 3, 1, 250), (3, 32, 64, 96, 192, 3, 2, 60), (6, 16, 32, 128, 160, 3, 1, 300),
                                (6, 16, 32, 192, 192, 1, 1, 60)]:
    pad = 1 if k == 3 else 0
    Ho, Wo = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
    x = K.empty_nhwc(N, Ci, H, W, dt, 'cuda'); x.normal_()
    dy = K.empty_nhwc(N, Co, Ho, Wo, dt, 'cuda'); dy.normal_()
    dw = torch.zeros(Co, k, k, Ci, device='cuda')
    d = ConvDesc(N, H, W, Ci, Co, k, k, s, pad, Ho, Wo, Ci, Co, 1, 0)
    t = _graph_time_ms(lambda st: call("fs_conv2d_wgrad_strided", st, ctypes.byref(d), K._p(x), K._p(dy), K._p(dw), k * k * Ci, 1, Ci))
    ws = torch.empty(K.WORKSPACE_BYTES, dtype=torch.uint8, device="cuda"); ws[-K.WS_COUNTER_BYTES:].zero_()
    t2 = _graph_time_ms(lambda st: call("fs_conv2d_wgrad_ws", st, ctypes.byref(d), K._p(x), K._p(dy), K._p(dw), k * k * Ci, 1, Ci, K._p(ws), K.WORKSPACE_BYTES))
    tot += t * cnt
    tot2 += t2 * cnt
    print("N=%d %3dx%-3d %3d->%3d k%d s%d | M=%5d | atomics %7.2f us | slab reduction %7.2f us" % (N, H, W, Ci, Co, k, s, N * Ho * Wo, t * 1e3, t2 * 1e3))
print("weighted total: atomics %.2f ms, deterministic slab reduction %.2f ms" % (tot, tot2))
