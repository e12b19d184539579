"""TEST INFRASTRUCTURE — ctypes loader for oracle/prim.c (the plain-C restatement of conv2d / bilinear / batchnorm)."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "prim.c")
LIB = os.path.join(HERE, "_build", "libprim.so")


def build(force=False):
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        subprocess.run(["gcc", "-O2", "-shared", "-fPIC", SRC, "-o", LIB, "-lm"], check=True)
    return LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
    return _lib


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def conv2d(x, w, bias=None, stride=1, pad=0):
    x, w = _f(x), _f(w)
    b = _f(bias) if bias is not None else None
    N, Cin, H, W = x.shape
    Cout, _, R, S = w.shape
    Ho, Wo = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - S) // stride + 1
    y = np.empty((N, Cout, Ho, Wo), np.float32)
    lib().prim_conv2d(_p(x), _p(w), _p(b), _p(y), N, Cin, H, W, Cout, R, S, stride, pad)
    return y


def bilinear(x, size):
    x = _f(x)
    N, C, Hi, Wi = x.shape
    y = np.empty((N, C, size[0], size[1]), np.float32)
    lib().prim_bilinear(_p(x), _p(y), N, C, Hi, Wi, int(size[0]), int(size[1]))
    return y


def batchnorm(x, gamma, beta, running_mean, running_var, training, momentum=0.1, eps=1e-5):
    """Returns y; running_mean / running_var (float32 numpy arrays) are updated in place when training."""
    x = _f(x)
    N, C, H, W = x.shape
    y = np.empty_like(x)
    assert running_mean.dtype == np.float32 and running_var.dtype == np.float32
    lib().prim_batchnorm(_p(x), _p(y), N, C, H, W, _p(_f(gamma)), _p(_f(beta)), _p(running_mean), _p(running_var),
                         int(training), ctypes.c_float(momentum), ctypes.c_float(eps))
    return y
