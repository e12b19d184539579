"""The differentiable latency model of the search as tensor algebra (split out of model_search.py, round 6).

`Network_Multi_Path.forward_latency` (reference search/model_search.py:413-476) walks the supernet layer by layer with ~500 scalar ops per
call and one host read of the sampled widths; the architect calls it three times per step (architect.py:66-72).  Two of the three calls
(constant betas) are LINEAR in the per-MixedOp latencies and the third (live betas, constant alphas and widths) is a product of ~90 affine
maps: the plans below derive those forms once per (architecture, input size, lookup table) by running the reference's recurrence on
symbols, and a call is ~12-15 tensor ops.  The methods live in a mix-in of Network_Multi_Path; tests/test_supernet.py pins them to the
per-MixedOp evaluation and to the reference's fixtures.
"""
import torch
import torch.nn.functional as F

from .genotypes import PRIMITIVES


class _Lin:
    """A linear form sum_e coef[e] * x_e + const over the MixedOp latencies x_e (forward_latency with constant betas)."""
    __slots__ = ("c", "k")

    def __init__(self, c=None, k=0.0):
        self.c, self.k = (c or {}), k

    def __add__(self, o):
        if isinstance(o, _Lin):
            c = dict(self.c)
            for e, v in o.c.items():
                c[e] = c.get(e, 0.0) + v
            return _Lin(c, self.k + o.k)
        return _Lin(dict(self.c), self.k + float(o))
    __radd__ = __add__

    def __mul__(self, f):
        f = float(f)
        return _Lin({e: v * f for e, v in self.c.items()}, self.k * f)
    __rmul__ = __mul__


class LatencyModelMixin:
    # ---- forward_latency with constant betas as ONE dot product ---------------------------------------------------------------
    # With beta=False (two of the three calls of the architect's latency penalty, architect.py:66-72) the reference's layer
    # recurrence (model_search.py:430-470) is LINEAR in the per-MixedOp latencies x_e = <LUT row, alpha row> * score_in * score_out
    # with coefficients that depend only on the topology: total = const + <coef, x>.  The plan below derives coef once by running
    # the same recurrence on linear forms, and tabulates the LUT rows of every width pair of every MixedOp on the device, so a call
    # is ~15 tensor ops instead of ~500 scalar ones, and the Gumbel-width call needs NO host read-back of the sampled indices (the
    # per-MixedOp path reads them to build its LUT keys: a device sync in the middle of the architecture step).
    def _latency_plan(self, size):
        from . import operations
        lut = operations.latency_lookup_table
        key = (self.arch_idx, tuple(size), len(lut), float(sum(lut.values())))
        plans = self.__dict__.setdefault("_latency_plans", {})
        if key in plans:
            return plans[key]
        k = self.arch_idx
        W = self._width_mult_list
        counts = (self._layers - 1, self._layers - 1, self._layers - 2)
        offs = (0, counts[0], counts[0] + counts[1])
        slot_probe = [[("slot", offs[s] + n) for n in range(counts[s])] for s in range(3)]
        stem_latency, sz = 0.0, tuple(size)
        for m in self.stem[k]:
            latency, sz = m.forward_latency(sz)
            stem_latency += float(latency)
        evals, index = [], {}            # one variable per MixedOp that is evaluated

        def variable(mixed, hw, r_in, r_out, alpha_row):
            if id(mixed) not in index:
                index[id(mixed)] = len(evals)
                evals.append((mixed, hw, r_in, r_out, alpha_row))
            return _Lin({index[id(mixed)]: 1.0})
        alpha_off = (0, self._layers, 2 * self._layers - 1)
        hw_prev = [[(sz[1], sz[2]), None]]
        T = [[_Lin(k=stem_latency), _Lin()], [_Lin(), _Lin()], [_Lin(), _Lin()]]
        half = 0.5                                                   # the constant betas (`_arch_tensors(beta=False)`)
        for i, cells in enumerate(self.cells):
            hw_out, latency = [], []
            for j, cell in enumerate(cells):
                r = self._cell_ratio(i, j, slot_probe)
                row = alpha_off[j] + (i - j)
                if j == 0 or i == j:
                    hw = hw_prev[0][0] if j == 0 else hw_prev[j - 1][1]
                    o = variable(cell._op, hw, r[0], r[1], row)
                    d = variable(cell.downsample, hw, r[0], r[2], row) if cell._down else None
                    hw_out.append((hw, (hw[0] // 2, hw[1] // 2) if cell._down else None))
                    latency.append([o, d])
                else:           # from down (0) and from keep (1): the same MixedOp on inputs of one size, weights b0 + b1
                    hw = hw_prev[j][0]
                    assert hw_prev[j - 1][1] == hw
                    o = variable(cell._op, hw, r[0], r[1], row)
                    d = variable(cell.downsample, hw, r[0], r[2], row) if cell._down else None
                    hw_out.append((hw, (hw[0] // 2, hw[1] // 2) if cell._down else None))
                    latency.append([half * o + half * o, (half * d + half * d) if d is not None else _Lin()])
            hw_prev = hw_out
            for ii, lat in enumerate(latency):          # the reference's recurrence, including its use of the leftover `j`
                if ii == 0:
                    if lat[0] is not None: T[ii][0] = T[ii][0] + lat[0]
                    if lat[1] is not None: T[ii][1] = T[ii][0] + lat[1]
                elif i == ii:
                    if lat[0] is not None: T[ii][0] = T[ii - 1][1] + lat[0]
                    if lat[1] is not None: T[ii][1] = T[ii - 1][1] + lat[1]
                else:
                    if lat[0] is not None: T[ii][0] = half * T[ii][0] + half * T[ii - 1][1] + lat[0]
                    if lat[1] is not None: T[ii][1] = half * T[ii][0] + half * T[ii - 1][1] + lat[1]
        total = T[0][0] + T[1][0] + T[2][0]
        E, nW = len(evals), len(W)
        n_slots = sum(counts)
        table = torch.zeros(E, nW * nW, len(PRIMITIVES))
        slot_in, slot_out, n_out = [], [], []
        for e, (mixed, hw, r_in, r_out, _) in enumerate(evals):
            opts_in = list(W) if isinstance(r_in, tuple) else [r_in]
            opts_out = list(W) if isinstance(r_out, tuple) else [r_out]
            slot_in.append(r_in[1] if isinstance(r_in, tuple) else n_slots)       # n_slots: the fixed-width pseudo slot
            slot_out.append(r_out[1] if isinstance(r_out, tuple) else n_slots)
            n_out.append(len(opts_out))
            for a, w0 in enumerate(opts_in):
                for b, w1 in enumerate(opts_out):
                    mixed.set_prun_ratio((w0, w1))
                    for q, op in enumerate(mixed._ops):
                        table[e, a * len(opts_out) + b, q] = float(op.forward_latency((int(op.C_in * w0), hw[0], hw[1]))[0])
        dev = getattr(self, self._arch_names[k]["alphas"][0]).device
        coef = torch.tensor([total.c.get(e, 0.0) for e in range(E)], dtype=torch.float32)
        plan = dict(E=E, const=float(total.k), coef=coef.to(dev), table=table.to(dev), table_host=table, index=index,
                    stem_latency=stem_latency, slots_host=(slot_in, slot_out, n_out),
                    slot_in=torch.tensor(slot_in, device=dev), slot_out=torch.tensor(slot_out, device=dev),
                    n_out=torch.tensor(n_out, device=dev), alpha_rows=torch.tensor([ev[4] for ev in evals], device=dev),
                    rows=torch.arange(E, device=dev), n_slots=n_slots, fixed={})
        plans[key] = plan
        return plan

    def _forward_latency_linear(self, size, alpha, ratio):
        k = self.arch_idx
        plan = self._latency_plan(size)
        mode = "max"
        if ratio:
            mode = self.prun_mode if self.prun_mode is not None else self._prun_modes[k]
        dev = plan["coef"].device
        nW = len(self._width_mult_list)
        scores = None
        if mode == "arch_ratio":
            ratios = self.sample_prun_ratio(mode=mode, read_indices=False)          # same RNG draws as the per-MixedOp path
            if getattr(ratios, "stacked", None) is not None:
                R, idx = ratios.stacked, ratios.index                               # straight-through one-hots [slots, widths], arg-max per slot
            else:
                flat = [r for scale in ratios for r in scale]
                idx = torch.cat([r._fs_index_t for r in flat])                      # sampled width index per slot, on the device
                R = torch.stack(flat)
            score = R.gather(1, idx[:, None]).squeeze(1)                            # = ratio[k] of `_width_and_score`
            idx_ext = torch.cat([idx, idx.new_zeros(1)])
            scores = torch.cat([score, score.new_ones(1)])
            k_in, k_out = idx_ext[plan["slot_in"]], idx_ext[plan["slot_out"]]
            L = plan["table"][plan["rows"], k_in * plan["n_out"] + k_out]
        else:
            if mode not in ("max", "min"):
                return None                                                         # host-sampled widths: per-MixedOp path
            if mode not in plan["fixed"]:
                w = 0 if mode == "min" else nW - 1
                idx_ext = torch.full((plan["n_slots"] + 1,), w, device=dev)
                idx_ext[-1] = 0
                k_in, k_out = idx_ext[plan["slot_in"]], idx_ext[plan["slot_out"]]
                plan["fixed"][mode] = plan["table"][plan["rows"], k_in * plan["n_out"] + k_out]
            L = plan["fixed"][mode]
        if alpha:
            names = self._arch_names[k]["alphas"]
            A = torch.cat([F.softmax(getattr(self, n), dim=-1) for n in names])[plan["alpha_rows"]]
            x = (L * A).sum(1)
        else:
            x = L.sum(1) * (1. / len(PRIMITIVES))
        if scores is not None:
            x = x * scores[plan["slot_in"]] * scores[plan["slot_out"]]
        return plan["const"] + (plan["coef"] * x).sum()

    # ---- forward_latency with live betas and constant alpha / widths (the architect's third call) ----------------------------
    # Every assignment of the reference's recurrence is an affine map of the state (T00, T01, T10, T11, T20, T21, 1) whose
    # entries are linear in the softmaxed betas: M_k = C_k + sum_b beta_b G_k[.,.,b].  All ~90 maps are built by ONE einsum and
    # multiplied together by a log-depth tree of batched matmuls: ~12 tensor ops instead of ~350 scalar ones, same arithmetic up
    # to fp32 summation order.
    def _latency_beta_plan(self, size):
        base = self._latency_plan(size)
        if "beta" in base:
            return base["beta"]
        k = self.arch_idx
        nW = len(self._width_mult_list)
        slot_in, slot_out, n_out = base["slots_host"]
        n_slots = base["n_slots"]
        x = []                                         # per MixedOp: <LUT row at the maximum widths, uniform alpha>
        for e in range(base["E"]):
            k_in = nW - 1 if slot_in[e] < n_slots else 0
            k_out = nW - 1 if slot_out[e] < n_slots else 0
            x.append(float(base["table_host"][e, k_in * n_out[e] + k_out].sum()) * (1. / len(PRIMITIVES)))
        index = base["index"]
        rows = (0, self._layers - 2, self._layers - 3)                 # rows of betas[1], betas[2]
        boff = (0, 0, 2 * rows[1])

        def bsym(j, row, c):                                            # flat index of betas[j][row][c], python negative-row semantics
            return boff[j] + (row % rows[j]) * 2 + c
        nb = 2 * (rows[1] + rows[2])
        ONE = 6
        steps = []                                                      # (target, [(beta index or None, coefficient, source)])

        def var(ii, c):
            return 2 * ii + c
        for i, cells in enumerate(self.cells):
            lat = []
            for j, cell in enumerate(cells):
                xo = x[index[id(cell._op)]]
                xd = x[index[id(cell.downsample)]] if cell._down else None
                if j == 0 or i == j:
                    lat.append(([(None, xo, ONE)], [(None, xd, ONE)] if xd is not None else None))
                else:
                    b0, b1 = bsym(j, i - j - 1, 0), bsym(j, i - j - 1, 1)
                    lat.append(([(b0, xo, ONE), (b1, xo, ONE)], [(b0, xd, ONE), (b1, xd, ONE)] if xd is not None else []))
            jl = len(cells) - 1                                         # the reference's leftover loop variable
            for ii, (l0, l1) in enumerate(lat):
                if ii == 0:
                    steps.append((var(0, 0), [(None, 1.0, var(0, 0))] + l0))
                    if l1 is not None:
                        steps.append((var(0, 1), [(None, 1.0, var(0, 0))] + l1))
                elif i == ii:
                    steps.append((var(ii, 0), [(None, 1.0, var(ii - 1, 1))] + l0))
                    if l1 is not None:
                        steps.append((var(ii, 1), [(None, 1.0, var(ii - 1, 1))] + l1))
                else:
                    w0, w1 = bsym(jl, i - jl - 1, 0), bsym(jl, i - jl - 1, 1)
                    steps.append((var(ii, 0), [(w1, 1.0, var(ii, 0)), (w0, 1.0, var(ii - 1, 1))] + l0))
                    if l1 is not None:
                        steps.append((var(ii, 1), [(w1, 1.0, var(ii, 0)), (w0, 1.0, var(ii - 1, 1))] + l1))
        K_ = 1
        while K_ < len(steps):
            K_ *= 2
        C = torch.eye(7).repeat(K_, 1, 1)
        G = torch.zeros(K_, 7, 7, nb)
        for n, (target, terms) in enumerate(steps):
            C[n, target] = 0.0
            for b, coef, src in terms:
                if b is None:
                    C[n, target, src] += coef
                else:
                    G[n, target, src, b] += coef
        dev = base["coef"].device
        s0 = torch.zeros(7)
        s0[0], s0[ONE] = base["stem_latency"], 1.0
        sel = torch.zeros(7)
        sel[0] = sel[2] = sel[4] = 1.0
        base["beta"] = dict(C=C.to(dev), G=G.to(dev), s0=s0.to(dev), sel=sel.to(dev))
        return base["beta"]

    def _forward_latency_beta(self, size):
        plan = self._latency_beta_plan(size)
        names = self._arch_names[self.arch_idx]["betas"]
        bflat = torch.cat([F.softmax(getattr(self, n), dim=-1).reshape(-1) for n in names])
        M = plan["C"] + torch.matmul(plan["G"], bflat)                  # [K, 7, 7]; step n applies M[n]
        while M.shape[0] > 1:                                           # later steps multiply from the left
            M = torch.bmm(M[1::2], M[0::2])
        return torch.dot(plan["sel"], M[0] @ plan["s0"])
