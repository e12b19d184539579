"""Host-side checks of the MixedOp launch programs (fasterseg_amd/program.py): the lowering runs without a GPU (it only
needs shapes, strides and addresses), so the command stream can be decoded here and checked against the executor's
contract (csrc/program.hip): argument counts per op, every arena reference inside its arena, zero regions first.  The
numerical equivalence with the per-module autograd path is a GPU test (tests/test_train_steps_gpu.py)."""
import pytest
import torch

from fasterseg_amd import model_search, program
from fasterseg_amd.parallel import FlatGradientSync

NARGS = {program.OP_MEMSET: 2, program.OP_PACK_WEIGHT: 10, program.OP_CONV_FWD: 9, program.OP_UNIT_FWD: 16, program.OP_UNIT_BWD: 23,
         program.OP_WGRAD_STRIDED: 9, program.OP_CHANNEL_STATS: 6, program.OP_BN_FINALIZE: 14, program.OP_AFFINE_ACT: 10,
         program.OP_BN_BWD_REDUCE: 13, program.OP_BN_BWD_APPLY: 19, program.OP_BILINEAR_FWD: 3, program.OP_BILINEAR_BWD: 4,
         program.OP_WSUM: 9, program.OP_WSUM_BWD: 9, program.OP_WSUM_DOTS: 9, program.OP_AXPY: 9, program.OP_BN_UNIT_FWD: 20,
         program.OP_BN_UNIT_BWD: 20}
WIDTHS = [4. / 12, 6. / 12, 8. / 12, 10. / 12, 1.]


JOINS = []          # (op, join bit) of every command decoded last


def decode(words, n, sizes):
    """-> list of (op, [args]); asserts structural validity."""
    w = list(words[:n])
    pos, cmds = 0, []
    del JOINS[:]
    while pos < n:
        word, nargs = w[pos], w[pos + 1]
        op, join = word & 0xffff, (word >> 40) & 1          # bits 16-39: stream lane, bit 40: JOIN with the next command
        pos += 2
        assert op in NARGS and nargs == NARGS[op], (op, nargs)
        JOINS.append((op, join))
        args = []
        for _ in range(nargs):
            kind, s, v = w[pos:pos + 3]
            pos += 3
            if kind == 2:
                assert 0 <= s < program.N_SLOTS
                if s in sizes:
                    assert 0 <= v < sizes[s], (op, s, v, sizes[s])
                args.append(("ptr", s, v))
            elif kind == 4:
                arr = [(w[pos + 2 * j], w[pos + 2 * j + 1]) for j in range(s)]
                pos += 2 * s
                for sl, off in arr:
                    assert (sl, off) == (0, -1) or sl not in sizes or 0 <= off < sizes[sl]
                args.append(("ptrs", arr))
            elif kind == 5:
                args.append(("ints", w[pos:pos + s]))
                pos += s
            else:
                assert kind in (0, 1, 3)
                args.append((kind, v))
        cmds.append((op, args))
    assert pos == n
    return cmds


@pytest.mark.parametrize("stride", [1, 2])
@pytest.mark.parametrize("want_w", [False, True])
def test_mixed_op_lowering_structure(stride, want_w):
    torch.manual_seed(0)
    m = model_search.MixedOp(48, 48 * stride, stride=stride, width_mult_list=WIDTHS).train()
    sync = None
    if want_w:
        sync = FlatGradientSync(m.parameters())
        sync.prepare()
    else:
        for p in m.parameters():
            p.requires_grad_(False)
    m.set_prun_ratio((8. / 12, 10. / 12))
    cin = 32
    prog = program.lower_mixed_op(m, (2, cin, 16, 24), cin, torch.float32, torch.device("cpu"), need_x=True, need_coef=not want_w,
                                  want_w=want_w, sink=sync)
    cout = int(48 * stride * 10 / 12)
    assert prog.out_shape == (2, cout, 16 // stride, 24 // stride)
    fwd = decode(prog.f_words, prog.f_n, {program.SAVE: prog.save_bytes, program.TMPF: prog.tmpf_bytes, program.ZF: prog.zf_bytes})
    bwd = decode(prog.b_words, prog.b_n, {program.SAVE: prog.save_bytes, program.TMPB: prog.tmpb_bytes, program.ZB: prog.zb_bytes})
    f_ops, b_ops = [c[0] for c in fwd], [c[0] for c in bwd]
    # JOIN runs (round 5): the FactorizedReduce of a stride-2 MixedOp issues its two 1x1 convolutions, their two weight gradients and their
    # two data gradients as one grouped launch each; a joined command is always followed by the same op
    decode(prog.f_words, prog.f_n, {})
    f_join = list(JOINS)
    decode(prog.b_words, prog.b_n, {})
    b_join = list(JOINS)
    for seq in (f_join, b_join):
        for (op, j), (nxt, _) in zip(seq, seq[1:] + [(None, 0)]):
            assert not j or nxt == op, (op, nxt)
    assert [op for op, j in f_join if j] == ([program.OP_CONV_FWD] if stride == 2 else [])
    want_b = ([program.OP_WGRAD_STRIDED] if want_w else []) + [program.OP_CONV_FWD]
    assert [op for op, j in b_join if j] == (want_b if stride == 2 else [])
    # forward: one weighted sum last; 6 conv->BN units for stride 1 (skip 1x1, conv, downup, 2x conv_2x, 2x ...).  No fill command: the
    # zero-initialised accumulators (BN statistics / reductions, coefficient gradients) are slices of the step's zero arena (slots ZF / ZB)
    assert program.OP_MEMSET not in f_ops and program.OP_MEMSET not in b_ops and f_ops[-1] == program.OP_WSUM
    assert prog.zf_bytes > 0 and prog.zb_bytes > 0 and prog.zf_bytes % 256 == 0 and prog.zb_bytes % 256 == 0
    if not want_w:
        assert 0 <= prog.gcoef_off < prog.zb_bytes
    units = 7 if stride == 1 else 6            # stride 2: the skip is a FactorizedReduce (two plain convs + BN), no 1x1 unit
    assert f_ops.count(program.OP_UNIT_FWD) == units and b_ops.count(program.OP_UNIT_BWD) == units
    assert f_ops.count(program.OP_BILINEAR_FWD) == (4 if stride == 1 else 2)
    assert f_ops.count(program.OP_CONV_FWD) == (0 if stride == 1 else 2)
    assert f_ops.count(program.OP_BN_UNIT_FWD) == b_ops.count(program.OP_BN_UNIT_BWD) == (0 if stride == 1 else 1)
    assert b_ops.count(program.OP_WSUM_BWD) == 1 and b_ops[-1] == program.OP_WSUM
    assert (program.OP_WSUM_DOTS in b_ops) == (not want_w)
    if want_w:
        names = {id(p) for p in prog.touched}
        used = [p for n, p in m.named_parameters() if p.grad is not None]
        assert len(prog.touched) == len(names) and len(names) > 0
        assert all(sync.accepts(p) for p in prog.touched)
        if stride == 2:
            assert b_ops.count(program.OP_WGRAD_STRIDED) == 2
    else:
        assert prog.touched == []
    assert prog.valid()


@pytest.mark.parametrize("stride", [1, 2])
@pytest.mark.parametrize("want_w", [False, True])
def test_fused_mixed_op_lowering_structure(stride, want_w):
    """With the pairs' storage made adjacent (fusion.colocate / flat_order) a MixedOp lowers to 5 conv->BN units instead of 7 (stride
    1: skip, [conv | conv_2x.conv1], conv_2x.conv2, [conv_downup | conv_2x_downup.conv1], conv_2x_downup.conv2), ONE down-sample
    instead of two, and the two fused units carry the two-segment descriptor fields; without adjacency the same call lowers unfused."""
    import ctypes
    from fasterseg_amd import fusion
    from fasterseg_amd._lib import FS_CONV_RELU, FS_CONV_RELU_TAIL, ConvDesc
    torch.manual_seed(0)
    m = model_search.MixedOp(48, 48 * stride, stride=stride, width_mult_list=WIDTHS).train()
    m.set_prun_ratio((8. / 12, 10. / 12))
    for p in m.parameters():
        p.requires_grad_(False)
    plain = program.lower_mixed_op(m, (2, 32, 16, 24), 32, torch.float32, torch.device("cpu"), True, True, False, None)
    assert not plain.fused
    for p in m.parameters():
        p.requires_grad_(True)
    assert fusion.colocate(m) == 2 * len(WIDTHS)
    sd = m.state_dict()
    assert len(sd) == len({k: None for k in sd}) and all(sd[k].shape == v.shape for k, v in model_search.MixedOp(
        48, 48 * stride, stride=stride, width_mult_list=WIDTHS).state_dict().items()), "state_dict keys / shapes unchanged"
    sync = None
    if want_w:
        sync = FlatGradientSync(fusion.flat_order(m, m.parameters()))
        sync.prepare()
    else:
        for p in m.parameters():
            p.requires_grad_(False)
    m.set_prun_ratio((8. / 12, 10. / 12))
    prog = program.lower_mixed_op(m, (2, 32, 16, 24), 32, torch.float32, torch.device("cpu"), need_x=True, need_coef=not want_w,
                                  want_w=want_w, sink=sync)
    assert prog.fused
    fwd = decode(prog.f_words, prog.f_n, {program.SAVE: prog.save_bytes, program.TMPF: prog.tmpf_bytes})
    bwd = decode(prog.b_words, prog.b_n, {program.SAVE: prog.save_bytes, program.TMPB: prog.tmpb_bytes})
    f_ops, b_ops = [c[0] for c in fwd], [c[0] for c in bwd]
    units = 5 if stride == 1 else 4
    assert f_ops.count(program.OP_UNIT_FWD) == units and b_ops.count(program.OP_UNIT_BWD) == units
    assert f_ops.count(program.OP_BILINEAR_FWD) == (3 if stride == 1 else 1) and b_ops.count(program.OP_BILINEAR_BWD) == (3 if stride == 1 else 1)
    assert b_ops[-1] == program.OP_WSUM and [a for op, a in bwd if op == program.OP_WSUM][-1][2] == (0, 3), "dx = sum of three tensors"
    cout = int(48 * stride * 10 / 12)
    blob = bytes(prog.f_blob)
    seg = []
    for op, args in fwd:
        if op == program.OP_UNIT_FWD:
            d = ConvDesc.from_buffer_copy(blob[args[0][1]:args[0][1] + ctypes.sizeof(ConvDesc)])
            if d.n_seg:
                seg.append(d)
    assert len(seg) == 2 and all(d.Cout == 2 * cout and d.n_seg == cout and d.y_cs == 2 * cout for d in seg)
    assert seg[0].flags & FS_CONV_RELU and not seg[0].flags & FS_CONV_RELU_TAIL          # conv | conv_2x.conv1: ReLU on both
    # conv_downup's ReLU follows its up-sample (operations.py:271-276): at stride 1 only the conv_2x_downup half is rectified here
    assert bool(seg[1].flags & FS_CONV_RELU_TAIL) == (stride == 1)
    if want_w:
        assert all(d.g_jump == 48 * stride - cout for d in seg)
        assert len({id(p) for p in prog.touched}) == len(prog.touched) > 0 and all(sync.accepts(p) for p in prog.touched)
    assert prog.valid()


def test_lowering_reads_resident_packs_in_place():
    """With packed copies registered for every filter (what optim.FlatSGD does) the programs contain no fs_pack_weight
    commands and the conv descriptors carry the row / tap strides of the full-size packs."""
    import ctypes
    from fasterseg_amd import functional as FN
    from fasterseg_amd._lib import ConvDesc
    torch.manual_seed(0)
    m = model_search.MixedOp(48, 48, stride=1, width_mult_list=WIDTHS).train()
    for p in m.parameters():
        p.requires_grad_(False)
    packs = []
    for p in m.parameters():
        if p.dim() == 4:
            O, I, R, S = p.shape
            fwd, flip = torch.zeros(O, R, S, I), torch.zeros(I, R, S, O)
            packs.append((fwd, flip))
            FN.register_resident_pack(p, fwd, flip)
    m.set_prun_ratio((8. / 12, 10. / 12))
    prog = program.lower_mixed_op(m, (2, 32, 16, 24), 32, torch.float32, torch.device("cpu"), need_x=True, need_coef=True,
                                  want_w=False, sink=None)
    fwd = decode(prog.f_words, prog.f_n, {program.SAVE: prog.save_bytes, program.TMPF: prog.tmpf_bytes})
    bwd = decode(prog.b_words, prog.b_n, {program.SAVE: prog.save_bytes, program.TMPB: prog.tmpb_bytes})
    assert all(op != program.OP_PACK_WEIGHT for op, _ in fwd + bwd)
    units = [args for op, args in fwd if op == program.OP_UNIT_FWD]
    assert len(units) == 7
    blob = bytes(prog.f_blob)
    for args in units:
        d = ConvDesc.from_buffer_copy(blob[args[0][1]:args[0][1] + ctypes.sizeof(ConvDesc)])
        assert d.w_ts == 48 and d.w_os == d.R * d.S * 48 and d.Cin in (32, 40) and d.Cout == 40       # full-width pack, sliced conv
    # an in-place edit of a parameter invalidates its pack: the next lowering packs that filter again
    w = m._ops[1].conv1.weight
    with torch.no_grad():
        w.add_(0.0)
    prog2 = program.lower_mixed_op(m, (2, 32, 16, 24), 32, torch.float32, torch.device("cpu"), need_x=True, need_coef=True,
                                   want_w=False, sink=None)
    ops2 = [op for op, _ in decode(prog2.f_words, prog2.f_n, {program.SAVE: prog2.save_bytes, program.TMPF: prog2.tmpf_bytes})]
    assert ops2.count(program.OP_PACK_WEIGHT) == 1


def test_executor_rejects_malformed_programs():
    """fs_exec_program validates before it launches: unknown ops / wrong arity are status codes, not crashes."""
    import ctypes
    from fasterseg_amd import _lib
    lib = _lib.lib()
    slots = (ctypes.c_void_p * program.N_SLOTS)()
    blob = (ctypes.c_ubyte * 8)()
    for words in ([999, 0], [program.OP_AFFINE_ACT, 1, 0, 0, 5], [program.OP_WSUM, 9]):
        arr = (ctypes.c_longlong * len(words))(*words)
        assert lib.fs_exec_program(None, arr, len(words), blob, slots, program.N_SLOTS) == 1     # FS_ERR_INVALID
        assert b"fs_exec_program" in lib.fs_last_error()


def test_lowering_with_bn_groups_sizes_the_per_group_buffers():
    """groups=2 (one evaluation of the MixedOp on the batched from-down / from-keep pair): every fused unit's descriptor carries
    bn_groups=2, the statistics / saved-moments / backward-reduction buffers are sized per group, and the FactorizedReduce BN
    goes through the grouped BN unit ops."""
    import ctypes
    from fasterseg_amd._lib import ConvDesc
    torch.manual_seed(0)
    m = model_search.MixedOp(48, 96, stride=2, width_mult_list=WIDTHS).train()
    for p in m.parameters():
        p.requires_grad_(False)
    m.set_prun_ratio((8. / 12, 10. / 12))
    progs = {}
    for g in (1, 2):
        progs[g] = program.lower_mixed_op(m, (2 * g, 32, 16, 24), 32, torch.float32, torch.device("cpu"), need_x=True, need_coef=True,
                                          want_w=False, sink=None, groups=g)
    one, two = progs[1], progs[2]
    assert two.out_shape[0] == 4 and two.out_shape[1:] == one.out_shape[1:]
    assert two.n_launches == one.n_launches                      # same launches, twice the batch

    def descs(prog):
        fwd = decode(prog.f_words, prog.f_n, {program.SAVE: prog.save_bytes, program.TMPF: prog.tmpf_bytes})
        out = []
        for op, args in fwd:
            if op == program.OP_UNIT_FWD:
                off = args[0][1]
                out.append(ConvDesc.from_buffer_copy(bytes(prog.f_blob)[off:off + ctypes.sizeof(ConvDesc)]))
        return out, fwd
    d1, f1 = descs(one)
    d2, f2 = descs(two)
    assert [d.bn_groups for d in d1] == [1] * len(d1) and [d.bn_groups for d in d2] == [2] * len(d2)
    assert all(b.N == 2 * a.N and (b.Cin, b.Cout, b.H, b.W) == (a.Cin, a.Cout, a.H, a.W) for a, b in zip(d1, d2))
    g1 = [args[2][1] for op, args in f1 if op == program.OP_BN_UNIT_FWD]
    g2 = [args[2][1] for op, args in f2 if op == program.OP_BN_UNIT_FWD]
    assert g1 == [1] and g2 == [2]
    assert two.save_bytes > one.save_bytes and two.tmpb_bytes > one.tmpb_bytes


def test_prewarm_enumerates_every_width_combination_of_a_call_site():
    """MixedOp.prewarm_programs: after one call site has been seen (a lowering with some width pair), every (in, out) pair of the
    width list is lowered for the sampled ratios, only the seen value for a fixed one; the programs land in the cache under the
    keys later calls will look up, so those calls lower nothing."""
    torch.manual_seed(0)
    m = model_search.MixedOp(48, 48, stride=1, width_mult_list=WIDTHS).train()
    for p in m.parameters():
        p.requires_grad_(False)
    r0, r1 = 8. / 12, 10. / 12
    m.set_prun_ratio((r0, r1))
    x = torch.empty_strided((2, 32, 16, 24), (16 * 24 * 32, 1, 24 * 32, 32)).requires_grad_(True)
    coef = torch.ones(5, requires_grad=True)
    assert m.prewarm_programs() == 0                         # no call site, no sampling information yet
    model_search._SAMPLING_PASS = False
    assert m._program(x, coef, r0, r1) is not None and not m.__dict__.get("_sites"), "a fixed-width pass records no call site"
    m.__dict__["_programs"].clear()
    model_search._SAMPLING_PASS = True                       # as Network_Multi_Path.forward sets it for "random" / Gumbel passes
    assert m._program(x, coef, r0, r1) is not None and len(m.__dict__["_sites"]) == 1
    m.__dict__["_ratio_sampled"] = (True, False)             # in-width sampled, out-width fixed (e.g. the last layer's head width)
    assert m.prewarm_programs() == len(WIDTHS) - 1
    m.__dict__["_ratio_sampled"] = (True, True)
    assert m.prewarm_programs() == len(WIDTHS) * len(WIDTHS) - len(WIDTHS)
    assert len(m._programs) == len(WIDTHS) ** 2
    before = dict(m._programs)
    for w0 in WIDTHS:
        for w1 in WIDTHS:
            m.set_prun_ratio((w0, w1))
            cin = m._ops[1].conv1.active_channels()[1]
            xx = torch.empty_strided((2, cin, 16, 24), (16 * 24 * cin, 1, 24 * cin, cin)).requires_grad_(True)
            prog = m._program(xx, coef, w0, w1)
            assert prog is not None and prog.out_shape[0] == 2
    assert m._programs == before, "a later call lowered a program the prewarm should have built"
    assert m.prewarm_programs() == 0
    model_search._SAMPLING_PASS = False


def test_conflict_free_chunks_keep_same_key_items_apart_and_in_order():
    """Layer calls (model_search._run_tasks): no chunk holds two evaluations of one (MixedOp, output width); such evaluations keep their
    order across chunks; everything else packs into the earliest chunk with room."""
    from fasterseg_amd.model_search import conflict_free_chunks
    key = lambda item: item[1]
    a = [(i, "k%d" % i) for i in range(5)]                     # pass 1: five MixedOps
    b = [(10 + i, "k%d" % i) for i in range(5)]                # pass 2 drew the same widths everywhere
    chunks = conflict_free_chunks(a + b, 24, key)
    assert chunks == [a, b]
    b2 = [(10, "k0"), (11, "x1"), (12, "k2"), (13, "x3"), (14, "x4")]       # two collisions
    chunks = conflict_free_chunks(a + b2, 24, key)
    assert chunks == [a + [(11, "x1"), (13, "x3"), (14, "x4")], [(10, "k0"), (12, "k2")]]
    chunks = conflict_free_chunks(a + b2, 6, key)              # capacity: the first chunk takes one more item
    assert [len(c) for c in chunks] == [6, 4] and all(len({key(i) for i in c}) == len(c) for c in chunks)
    order = {it: (ci, pos) for ci, c in enumerate(chunks) for pos, it in enumerate(c)}
    assert order[(0, "k0")] < order[(10, "k0")] and order[(2, "k2")] < order[(12, "k2")]
    three = [(0, "k"), (1, "k"), (2, "k")]
    assert conflict_free_chunks(three, 24, key) == [[(0, "k")], [(1, "k")], [(2, "k")]]
    assert conflict_free_chunks([], 24, key) == []


def test_fast_division_magic_numbers_are_exact_below_2_31():
    """csrc/common.h DivInt / fast_div (element-wise kernels) and the magic pair of csrc/wgrad.hip, restated: n // d by one multiply-high and
    a shift must be exact for every 0 <= n < 2^31 - checked at the edges of every quotient step around random and extreme dividends."""
    import random
    rng = random.Random(7)

    def div_int(d):                                   # DivInt(int d)
        if d <= 1:
            return None
        l = 0
        while (1 << l) < d:
            l += 1
        magic = ((1 << (31 + l)) // d) + 1
        assert magic < (1 << 32)
        return magic, l - 1

    def fast_div(n, d, m):
        return n if m is None else ((n * m[0]) >> 32) >> m[1]         # __umulhi(n, magic) >> shift

    def wgrad_magic(d):                               # wgrad_prepare's lambda: 64-bit product, dividends < 2^32
        l = 0
        while (1 << l) < d:
            l += 1
        return ((1 << (32 + l)) // d) + 1, 32 + l

    top = (1 << 31) - 1
    divisors = list(range(1, 130)) + [192, 384, 768, 1000, 4095, 4096, 4097, 65535, 65536, 1 << 20, (1 << 30) + 1, top] + [rng.randrange(2, 1 << 30) for _ in range(200)]
    for d in divisors:
        m, (wm, ws) = div_int(d), wgrad_magic(d)
        probes = {0, 1, d - 1, d, d + 1, top, top - 1, top - d if top > d else 0}
        for _ in range(40):
            q = rng.randrange(0, top // d + 1)
            probes.update(x for x in (q * d - 1, q * d, q * d + d - 1) if 0 <= x <= top)
            probes.add(rng.randrange(0, top + 1))
        for n in probes:
            assert fast_div(n, d, m) == n // d, (n, d)
            assert (n * wm) >> ws == n // d, (n, d)
