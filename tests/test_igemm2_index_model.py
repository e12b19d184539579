"""The index algebra of csrc/conv_igemm2.hip, pinned without a GPU: tools/emulate_igemm2.py restates the kernel's addressing in numpy -
XCD-aware block renumbering (must be a bijection), tile / K-slice decomposition, the per-block (row, tap) byte-offset table with the
range-check sentinel for padding, the LDS-DMA lane map with its source-side XOR swizzle and the matching fragment reads, and the
stride-2 data gradient by output-parity classes (1 / 2 / 2 / 4 of 9 taps) - and is compared with F.conv2d / its autograd input gradient
(what the reference's nn.Conv2d computes: search/operations.py:149,298,467-473).  The hardware semantics the model cannot see (zeros
from out-of-range buffer loads, M0 addressing) are covered on the device by tests/test_kernels_gpu.py."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import emulate_igemm2 as em  # noqa: E402


@pytest.mark.parametrize("case", em.CASES, ids=["%s%dx%d-%d-%d-k%d-s%d-p%d-%dx%d" % ("dgrad-" if c.get("dgrad") else "", c["H"], c["W"], c["Cin"], c["Cout"],
                                                                                   c["k"], c["stride"], c["pad"], c["BM"], c["BN"]) for c in em.CASES])
def test_index_model_matches_conv2d(case):
    assert em.run_case(**case) < 1e-4


def test_stride2_parity_classes_cover_every_output_pixel_once():
    """The four parity classes of a stride-2 data gradient partition the output rows: tiles per class from host_args add up, and the
    class tile ranges are disjoint and ordered (igemm2_configure's cls_start)."""
    for (N, Ho, Wo, BM) in ((1, 9, 13, 64), (2, 8, 12, 32), (3, 7, 5, 32), (1, 1, 1, 32)):
        a = em.host_args(N, (Ho + 1) // 2, (Wo + 1) // 2, 16, 16, 3, 3, 1, 1, Ho, Wo, 16, True, BM, 32)
        cs = a["cls_start"]
        assert cs[0] == 0 and all(cs[i] <= cs[i + 1] for i in range(4)) and cs[4] == a["tiles_m"]
        rows = sum(N * ((Ho - (c >> 1) + 1) // 2) * ((Wo - (c & 1) + 1) // 2) for c in range(4))
        assert rows == N * Ho * Wo
