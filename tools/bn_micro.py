"""Times fs_bn_group_fwd / _bwd (bn_col.hip) on the supernet's small-map geometries; FS_BN_SMALL=0 selects the generic
kernels, so two runs give the register-resident vs generic comparison quoted in bn_col.hip.  Run on an MI355X."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fasterseg_amd import kernels as K
from fasterseg_amd._lib import call
from fasterseg_amd.census import _graph_time_ms
dt = torch.bfloat16
print("FS_BN_SMALL=%s" % os.environ.get("FS_BN_SMALL", "1"))
for (N, C, H, W, G, splits) in [(3, 192, 8, 16, 1, 0), (3, 384, 8, 16, 1, 0), (3, 384, 4, 8, 1, 0), (3, 64, 8, 16, 1, 0), (6, 192, 8, 16, 2, 0),
                                (6, 384, 4, 8, 2, 0), (3, 192, 8, 16, 1, 4), (6, 192, 8, 16, 2, 4), (3, 384, 4, 8, 1, 8)]:
    M = N * H * W
    z = K.empty_nhwc(N, C, H, W, dt, 'cuda'); z.normal_()
    y = K.empty_nhwc(N, C, H, W, dt, 'cuda'); y.normal_()
    dy = K.empty_nhwc(N, C, H, W, dt, 'cuda'); dy.normal_()
    dz = K.empty_nhwc(N, C, H, W, dt, 'cuda')
    out = K.empty_nhwc(N, C, H, W, dt, 'cuda')
    saved = torch.rand(G * 4 * C, device='cuda') + 0.5
    gamma = torch.rand(C, device='cuda'); beta = torch.rand(C, device='cuda')
    red = torch.zeros(2 * C, device='cuda')
    parts = torch.randn(max(splits, 1), M, C, device='cuda')
    cs = K.channel_stride(z)
    tb = _graph_time_ms(lambda st: call("fs_bn_group_bwd", st, M, C, G, K._p(z), cs, K._p(dy), cs, K._p(y), cs, K._p(saved), K._p(gamma), 1, 1, K._p(dz), cs, K._p(red), None, None))
    tf = _graph_time_ms(lambda st: call("fs_bn_group_fwd", st, M, C, G, K._p(z), cs, K._p(parts) if splits else None, splits, K._p(gamma), K._p(beta), 1e-5, 0.1, None, None, None, K._p(saved), K._p(out), cs, 1, 1))
    print("px/group=%4d C=%3d G=%d splits=%d | fwd %6.2f us  bwd %6.2f us" % (M // G, C, G, splits, tf * 1e3, tb * 1e3))
