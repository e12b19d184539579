"""TEST INFRASTRUCTURE — not product code.

CPU restatement ("oracle") of the FasterSeg multi-resolution conv hot path.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
anything from this package; fasterseg_amd/ never does (tests/test_no_oracle_in_product.py
enforces it).

Parity status: the reference ships no tests or golden vectors, so the
oracle is pinned against outputs of the *reference itself*, imported unmodified in the
build container by oracle/make_golden.py (fixtures under tests/golden/, generating script
committed).  See oracle/README.md.
"""
