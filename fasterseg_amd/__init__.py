"""fasterseg_amd — MI355X-native implementation of FasterSeg's multi-resolution conv hot path.

Python host side mirroring the reference operator API (search/operations.py, slimmable_ops.py, seg_oprs.py,
genotypes.py, train/model_seg.py, search/model_search.py) on top of hand-written gfx950 HIP kernels reached through
the C ABI in include/fasterseg_hip.h.  See DESIGN.md.
"""
__version__ = "0.1.0"
