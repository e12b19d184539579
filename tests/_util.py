"""Shared helpers for the parity tests (golden fixtures + oracle drivers)."""
import json
import os

import numpy as np
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_json(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)


_npz_cache = {}


def load_npz(name):
    if name not in _npz_cache:
        _npz_cache[name] = dict(np.load(os.path.join(GOLD, name)))
    return _npz_cache[name]


def golden_get(store, key):
    """Returns (array, step) — arrays above the fixture size limit are stored as flat[::step]."""
    if key in store:
        return store[key], 1
    for k in store:
        if k.startswith(key + "@"):
            return store[k], int(k.rsplit("@", 1)[1])
    raise KeyError(key)


def assert_close_golden(actual, store, key, atol, rtol, what=""):
    want, step = golden_get(store, key)
    got = actual.detach().cpu().float().numpy() if torch.is_tensor(actual) else np.asarray(actual)
    if step > 1:
        got = got.reshape(-1)[::step]
    assert got.shape == want.shape, "%s %s: shape %s vs %s" % (what, key, got.shape, want.shape)
    err = np.abs(got - want)
    tol = atol + rtol * np.abs(want)
    assert (err <= tol).all(), "%s %s: max err %.3e (tol %.1e+%.1e*|ref|), max|ref| %.3e" % (
        what, key, float(err.max()), atol, rtol, float(np.abs(want).max()))


def shapes_template(shapes):
    return {k: torch.empty(v) for k, v in shapes.items()}


def arch_tensors(idx):
    raw = load_npz("arch_%d.npz" % idx)
    t = lambda k: torch.tensor(raw[k])
    alphas = [t("alpha_%d_%d" % (idx, s)) for s in range(3)]
    betas = [None, t("beta_%d_1" % idx), t("beta_%d_2" % idx)]
    ratios = [t("ratio_%d_%d" % (idx, s)) for s in range(3)]
    return alphas, betas, ratios, raw
