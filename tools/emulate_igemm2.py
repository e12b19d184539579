"""CPU emulation of conv_igemm2.hip's addressing (block renumbering, parity classes, tap table, DMA lane map with the source-side
swizzle, K slices, fragment reads) against F.conv2d / its input gradient.  Catches index-logic errors without a GPU; it cannot
check hardware semantics (range-checked DMA zeros, M0 addressing), which tests/test_kernels_gpu.py covers on the device.

    python tools/emulate_igemm2.py
"""
import itertools

import numpy as np
import torch
import torch.nn.functional as F

OOB = 0x80000000
BKB = 128


def emulate(x_nhwc, w_flat, a, BM, BN, WAVES_K, ES=4):
    """x_nhwc: float32 array [N,H,W,x_cs]; w_flat: float32 1-D array; a: dict of ConvArgs fields.  Returns y [M, Cout] float64 (or the
    slab sum for split-K) and the set of (m) rows written."""
    xb = x_nhwc.reshape(-1).view(np.uint8)
    wb = w_flat.view(np.uint8)
    M, Cout, Cin = a["M"], a["Cout"], a["Cin"]
    y = np.zeros((M, Cout), np.float64)
    written = np.zeros(M, np.int32)
    nwg = a["tiles_m"] * a["tiles_n"] * a["slices"]
    seen = set()
    for bid in range(nwg):
        q, r, xcd = nwg >> 3, nwg & 7, bid & 7
        logical = (xcd * (q + 1) if xcd < r else r * (q + 1) + (xcd - r) * q) + (bid >> 3)
        assert logical not in seen and 0 <= logical < nwg
        seen.add(logical)
        ntiles = a["tiles_m"] * a["tiles_n"]
        sl = logical // ntiles
        t = logical - sl * ntiles
        if a["n_major"]:
            tile_n = t // a["tiles_m"]; tile_m = t - tile_n * a["tiles_m"]
        else:
            tile_m = t // a["tiles_n"]; tile_n = t - tile_m * a["tiles_n"]
        n0 = tile_n * BN
        classes = a["classes"]
        zero_insert = a["transposed"] and not classes
        ph = pw = 0
        tile_c, Hc, Wc, Mc = tile_m, a["Ho"], a["Wo"], M
        th, tw, nth, ntw = [0, 1, 2], [0, 1, 2], a["R"], a["S"]
        if classes:
            c = sum(1 for k in range(1, 4) if tile_m >= a["cls_start"][k])
            ph, pw = c >> 1, c & 1
            tile_c = tile_m - a["cls_start"][c]
            Hc = (a["Ho"] - ph + 1) >> 1
            Wc = (a["Wo"] - pw + 1) >> 1
            Mc = (M // a["HoWo"]) * Hc * Wc
            th = [k for k in range(a["R"]) if ((ph - a["pad"] + k) & 1) == 0]
            tw = [k for k in range(a["S"]) if ((pw - a["pad"] + k) & 1) == 0]
            nth, ntw = len(th), len(tw)
        ntaps = nth * ntw
        m0 = tile_c * BM
        CBU = Cin * ES // 16
        w_ts_bytes = (Cin + a["w_tgap"]) * ES
        KU_TOT = ntaps * CBU
        ku_lo = sl * a["slice_units"]
        ku_hi = ku_lo + a["slice_units"] if (a["slices"] > 1 and ku_lo + a["slice_units"] < KU_TOT) else KU_TOT
        nsteps = (ku_hi - ku_lo + 7) >> 3 if ku_hi > ku_lo else 0
        sTap = np.full((BM, 9), OOB, np.int64)
        rowM = np.full(BM, -1, np.int64)
        for row in range(BM):
            mi = m0 + row
            if mi >= Mc:
                continue
            HW = Hc * Wc
            n = mi // HW; rem = mi - n * HW; qh = rem // Wc; qw = rem - qh * Wc
            rowM[row] = ((n * a["Ho"] + 2 * qh + ph) * a["Wo"] + 2 * qw + pw) if classes else mi
            for j in range(ntaps):
                jh = j // ntw; jw = j - jh * ntw
                ok = True
                if classes:
                    ih = (2 * qh + ph - a["pad"] + th[jh]) >> 1
                    iw = (2 * qw + pw - a["pad"] + tw[jw]) >> 1
                else:
                    ih = qh * a["stride"] - a["pad"] + jh
                    iw = qw * a["stride"] - a["pad"] + jw
                    if zero_insert:
                        ok = ((ih | iw) & 1) == 0
                        ih >>= 1; iw >>= 1
                ok = ok and 0 <= ih < a["H"] and 0 <= iw < a["W"]
                if ok:
                    sTap[row, j] = ((n * a["H"] + ih) * a["W"] + iw) * a["x_cs"] * ES
        sW = np.zeros(12, np.int64)
        for j in range(ntaps):
            jh = j // ntw; jw = j - jh * ntw
            sW[j] = (th[jh] * a["S"] + tw[jw]) * w_ts_bytes
        kseg_eff = a["k_seg"] * ES if a["k_seg"] > 0 else 0x7fffffff
        kjump_bytes = a["k_jump"] * ES
        ROWS = BM + BN
        acc = np.zeros((BM, BN), np.float64)
        for step in range(nsteps):
            lds = np.zeros(ROWS * BKB, np.uint8)
            lds[:] = 0xAB                                   # garbage: every byte must be overwritten by a DMA (or zero-filled by OOB)
            for wave, lane in itertools.product(range(4), range(64)):
                lrow = wave * 8 + (lane >> 3)
                swz = (lrow >> 1) & 7
                chunk = (lane & 7) ^ swz
                ku = ku_lo + step * 8 + chunk
                valid = ku < ku_hi
                tap = (ku // CBU) if valid else 0
                cb = (ku - tap * CBU) * 16
                jump = kjump_bytes if cb >= kseg_eff else 0
                wt = sW[tap] + cb + jump
                for s in range(ROWS // 32):
                    dst = (s * 4 + wave) * 1024 + lane * 16
                    if s < BM // 32:
                        tt = sTap[s * 32 + lrow, tap]
                        off = (tt + cb) if valid else OOB
                        src = xb
                    else:
                        b = s - BM // 32
                        n = n0 + b * 32 + lrow
                        nrow = n + (a["n_jump"] if (a["n_seg"] > 0 and n >= a["n_seg"]) else 0)
                        boff = nrow * a["w_os"] * ES if n < Cout else OOB
                        off = ((boff + wt) & 0xffffffff) if valid else OOB
                        src = wb
                    if off >= OOB:
                        lds[dst:dst + 16] = 0
                    else:
                        assert off + 16 <= src.size, ("read beyond the operand", off, src.size)
                        lds[dst:dst + 16] = src[off:off + 16]
            ldsf = lds.view(np.float32)
            # fragment reads: logical chunk c of row r sits at slot c ^ ((r >> 1) & 7)
            tile = np.zeros((ROWS, BKB // ES), np.float32)
            for row in range(ROWS):
                fsw = ((row & 31) >> 1) & 7
                for c in range(8):
                    p = row * BKB + ((c ^ fsw) * 16)
                    tile[row, c * (16 // ES):(c + 1) * (16 // ES)] = ldsf[p // 4:p // 4 + 16 // ES]
            acc += tile[:BM].astype(np.float64) @ tile[BM:].astype(np.float64).T
        for row in range(BM):
            m = rowM[row]
            if m < 0:
                continue
            for col in range(BN):
                co = n0 + col
                if co < Cout:
                    y[m, co] += acc[row, col]
            if tile_n == 0 and sl == 0:
                written[m] += 1
    assert len(seen) == nwg
    assert (written == 1).all(), "every output pixel is owned by exactly one tile row"
    return y


def host_args(N, H, W, Cin, Cout, R, S, stride, pad, Ho, Wo, x_cs, transposed, BM, BN, slices=1, w_os=None, w_tgap=0, n_seg=0, n_jump=0,
              k_seg=0, k_jump=0, ES=4):
    a = dict(H=H, W=W, Cin=Cin, Cout=Cout, R=R, S=S, stride=stride, pad=pad, Ho=Ho, Wo=Wo, x_cs=x_cs, M=N * Ho * Wo, HoWo=Ho * Wo,
             transposed=transposed, classes=transposed, w_os=w_os or R * S * Cin, w_tgap=w_tgap, n_seg=n_seg, n_jump=n_jump, k_seg=k_seg,
             k_jump=k_jump)
    a["tiles_n"] = (Cout + BN - 1) // BN
    if transposed:
        acc, cs = 0, []
        for c in range(4):
            ph, pw = c >> 1, c & 1
            mc = N * ((Ho - ph + 1) // 2) * ((Wo - pw + 1) // 2)
            cs.append(acc)
            acc += (mc + BM - 1) // BM
        a["cls_start"] = cs + [acc]
        a["tiles_m"] = acc
    else:
        a["cls_start"] = [0] * 5
        a["tiles_m"] = (a["M"] + BM - 1) // BM
    cbu = Cin * ES // 16
    steps = (R * S * cbu + 7) // 8
    slices = min(slices, steps) if not transposed else 1
    a["slice_units"] = (steps + slices - 1) // slices * 8 if slices > 1 else 0
    if slices > 1:
        slices = (steps * 8 + a["slice_units"] - 1) // a["slice_units"]
    a["slices"] = slices
    a["n_major"] = 1 if 9 * Cout > a["M"] else 0
    return a


def run_case(N, Cin, H, W, Cout, k, stride, pad, BM, BN, slices=1, dgrad=False, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) * 0.2
    if pad < 0:
        ref_f = lambda xx: F.conv2d(xx[:, :, -pad:, -pad:], w, None, stride, 0)
    else:
        ref_f = lambda xx: F.conv2d(xx, w, None, stride, pad)
    if not dgrad:
        ref = ref_f(x)
        Ho, Wo = ref.shape[2:]
        xn = x.permute(0, 2, 3, 1).contiguous().numpy()
        wp = w.permute(0, 2, 3, 1).contiguous().numpy().reshape(-1)              # [Cout][R][S][Cin]
        a = host_args(N, H, W, Cin, Cout, k, k, stride, pad, Ho, Wo, Cin, False, BM, BN, slices)
        y = emulate(xn, wp, a, BM, BN, 1)
        want = ref.permute(0, 2, 3, 1).reshape(-1, Cout).double().numpy()
    else:
        xr = x.clone().requires_grad_(True)
        yv = ref_f(xr)
        dy = torch.randn(yv.shape, generator=g)
        yv.backward(dy)
        ho, wo = yv.shape[2:]
        dn = dy.permute(0, 2, 3, 1).contiguous().numpy()
        wf = torch.flip(w, (2, 3)).permute(1, 2, 3, 0).contiguous().numpy().reshape(-1)   # [Cin][R][S][Cout], rotated
        # the transposed conv's geometry: input = dz (ho, wo, Cout channels), output = dx (H, W, Cin channels)
        a = host_args(N, ho, wo, Cout, Cin, k, k, 1, k - 1 - pad, H, W, Cout, stride == 2, BM, BN, slices)
        y = emulate(dn, wf, a, BM, BN, 1)
        want = xr.grad.permute(0, 2, 3, 1).reshape(-1, Cin).double().numpy()
    err = np.abs(y - want).max()
    assert err < 1e-4, err
    return err


CASES = [
    dict(N=1, Cin=32, H=9, W=11, Cout=40, k=3, stride=1, pad=1, BM=64, BN=64),
    dict(N=2, Cin=16, H=7, W=9, Cout=24, k=3, stride=2, pad=1, BM=32, BN=32),
    dict(N=1, Cin=48, H=6, W=8, Cout=72, k=3, stride=1, pad=1, BM=64, BN=64, slices=3),
    dict(N=2, Cin=24, H=8, W=6, Cout=16, k=1, stride=2, pad=0, BM=32, BN=32),
    dict(N=2, Cin=24, H=8, W=6, Cout=16, k=1, stride=2, pad=-1, BM=32, BN=32),
    dict(N=1, Cin=8, H=5, W=7, Cout=8, k=3, stride=1, pad=1, BM=32, BN=32),
    dict(N=1, Cin=40, H=12, W=10, Cout=136, k=3, stride=1, pad=1, BM=128, BN=64, slices=2),
    # data gradients: stride 1 (plain), stride 2 by parity classes, even and odd maps, 1x1 with both FactorizedReduce offsets
    dict(N=1, Cin=16, H=8, W=10, Cout=24, k=3, stride=1, pad=1, BM=64, BN=64, dgrad=True),
    dict(N=2, Cin=16, H=8, W=12, Cout=24, k=3, stride=2, pad=1, BM=32, BN=32, dgrad=True),
    dict(N=1, Cin=16, H=9, W=13, Cout=32, k=3, stride=2, pad=1, BM=64, BN=64, dgrad=True),
    dict(N=2, Cin=24, H=8, W=6, Cout=16, k=1, stride=2, pad=0, BM=32, BN=32, dgrad=True),
    dict(N=2, Cin=24, H=8, W=6, Cout=16, k=1, stride=2, pad=-1, BM=32, BN=32, dgrad=True),
    dict(N=1, Cin=16, H=7, W=5, Cout=16, k=1, stride=2, pad=-1, BM=32, BN=32, dgrad=True),
]

if __name__ == "__main__":
    for c in CASES:
        print(c, "max err %.2e" % run_case(**c))
    print("ok")
