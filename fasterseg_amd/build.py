"""Builds libfasterseg_hip.so (gfx950) in-tree with hipcc.  `python -m fasterseg_amd.build [--force]`.

Cross-compiles without a GPU.  Objects are cached under csrc/_build/ by source mtime; the shared library lands
next to this file so it travels to the GPU box with the repo snapshot (it is git-ignored, not gpurun-ignored).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_build")
LIB = os.path.join(HERE, "libfasterseg_hip.so")
SOURCES = ["api.cpp", "census.hip", "conv_igemm.hip", "conv_igemm2.hip", "conv3x3_halo.hip", "zoom_cell.hip", "elementwise.hip", "bn_col.hip", "resize.hip", "stem.hip", "wgrad.hip", "units.hip", "optim.hip", "program.hip", "loss.hip", "loss_up.hip", "eval.hip"]
HEADERS = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "bn_bodies.h"), os.path.join(CSRC, "conv_igemm.h"), os.path.join(CSRC, "group.h"), os.path.join(os.path.dirname(HERE), "include", "fasterseg_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
if os.environ.get("FS_BUILD_PROBES", "0") not in ("", "0"):
    # measurement-only kernel instantiations (conv_igemm2 ablations / 6- and 8-stage rings: tools/conv_sweep.py codes 107, 108, 120-133);
    # the product library is built without them.  Use with --force: objects are cached by mtime, not by flags.
    FLAGS.append("-DFS_BUILD_PROBES=1")


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src, force):
    obj = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
    path = os.path.join(CSRC, src)
    if force or _stale(obj, [path] + HEADERS):
        cmd = [_hipcc()] + FLAGS + (["-x", "hip"] if src.endswith(".hip") else []) + ["-c", path, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
        if r.stderr.strip():
            sys.stderr.write(r.stderr)
        return obj, True
    return obj, False


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    with ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        results = list(ex.map(lambda s: _compile(s, force), SOURCES))
    objs = [o for o, _ in results]
    if force or any(ch for _, ch in results) or _stale(LIB, objs):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
        if verbose:
            print("built", LIB)
    elif verbose:
        print("up to date:", LIB)
    # undefined symbols (a kernel whose host stub hipcc dropped, a missing object) must fail HERE, not at first use on the GPU box.
    # In a CHILD process: the library links the system libamdhip64, torch ships its own copy - whichever is loaded first serves both, and
    # loading this library before `import torch` leaves the process with two HIP runtimes (the library's launches then fail with "no
    # ROCm-capable device is detected": `python __graft_entry__.py smoke` did, round 5).  Every other loader imports torch first (_lib.py).
    check = subprocess.run([sys.executable, "-c", "import ctypes, os, sys; ctypes.CDLL(sys.argv[1], mode=os.RTLD_NOW)", LIB],
                           capture_output=True, text=True)
    if check.returncode != 0:
        raise RuntimeError("libfasterseg_hip.so does not load:\n%s" % check.stderr)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
