"""Fits the cost model behind conv_igemm2.hip's configuration heuristic to tools/conv_sweep.py measurements (offline, CPU).

    python tools/fit_igemm2.py gpurun_out/r04b_sweep_bf16.json gpurun_out/r04e_sweep_fp32_dgrad.json
Prints, per dtype, the fitted parameters and the regret of the model's choice against the best measured configuration and against the
round-3 kernel (code -2)."""
import itertools
import json
import math
import re
import sys

CFG = {100: (64, 64), 104: (32, 32), 105: (64, 32), 106: (32, 64), 103: (128, 128), 101: (128, 64), 102: (64, 128)}
CANDS = [100, 104, 105, 106]


def model(p, M, N, S, bm, bn, s, es):
    T0, a, bw, mf, ts, bws, ovl = p
    rows = bm + bn
    tiles = math.ceil(M / bm) * math.ceil(N / bn)
    nblk = tiles * s
    st = math.ceil(S / s)
    share = max(1.0, nblk / 256.0)
    fill = max(st * rows * a * 1e-3, tiles * S * rows * 128 / (bw * 1e6))          # us
    mfma = st * (bm * bn / 32.0) * (8.0 if es == 4 else 1.0) / 2400.0 * mf * share   # us
    loop = max(fill, mfma) + ovl * min(fill, mfma)
    t = T0 + loop
    if s > 1:
        t += ts + (s + 1) * M * N * 4 / (bws * 1e6)
    return t


def load(path, kind):
    d = json.load(open(path))
    rows = []
    for k, r in d.items():
        m = re.search(r"(\w+) %s N(\d+)\s+(\d+)->\s*(\d+) @\s*(\d+)x\s*(\d+)" % kind, k)
        if not m:
            continue
        dt, n, cin, cout, h, w = m.group(1), *map(int, m.groups()[1:])
        es = 4 if dt == "fp32" else 2
        M = n * h * w
        # dgrad rows: the kernel contracts over 9 * cout input channels and writes cin output channels
        Nn, K = cin, 9 * cout
        S = K * es / 128.0
        meas = {}
        for c in CANDS:
            for s in (1, 2, 4, 8):
                v = r.get(str(1000 * s + c))
                if v is not None:
                    meas[(c, s)] = v
        if meas:
            rows.append(dict(key=k, M=M, N=Nn, S=S, es=es, meas=meas, old=r["-2"]))
    return rows


def regret(p, rows):
    tot, worst = 0.0, 0.0
    for r in rows:
        pick = min(r["meas"], key=lambda cs: model(p, r["M"], r["N"], r["S"], *CFG[cs[0]], cs[1], r["es"]))
        best = min(r["meas"].values())
        g = r["meas"][pick] / best
        tot += math.log(g)
        worst = max(worst, g)
    return tot / len(rows), worst


def fit(rows):
    import random
    random.seed(1)
    best_p, best_r = None, (1e9, 1e9)
    base = [4.0, 2.6, 21.5, 1.0, 3.5, 2.5, 0.3]
    for it in range(6000):
        p = [b * math.exp(random.uniform(-0.7, 0.7)) for b in base] if it else base
        r = regret(p, rows)
        if r[0] < best_r[0]:
            best_p, best_r = p, r
            if it > 300:
                base = p
    return best_p, best_r


if __name__ == "__main__":
    for path in sys.argv[1:]:
        rows = load(path, "dgrad")
        if not rows:
            continue
        p, r = fit(rows)
        print(path, len(rows), "shapes; params T0 %.2f a %.2f bw %.1f mf %.2f ts %.2f bws %.2f ovl %.2f" % tuple(p), "mean log regret %.3f worst %.2f" % r)
        tot_m, tot_b, tot_o = 0, 0, 0
        for rr in rows:
            pick = min(rr["meas"], key=lambda cs: model(p, rr["M"], rr["N"], rr["S"], *CFG[cs[0]], cs[1], rr["es"]))
            best = min(rr["meas"], key=rr["meas"].get)
            tot_m += rr["meas"][pick]; tot_b += rr["meas"][best]; tot_o += rr["old"]
            print("  M=%5d N=%3d S=%5.1f pick %s %.1f  best %s %.1f  old %.1f" % (rr["M"], rr["N"], rr["S"], pick, rr["meas"][pick], best, rr["meas"][best], rr["old"]))
        print("  sum: model %.1f best %.1f old %.1f" % (tot_m, tot_b, tot_o))
