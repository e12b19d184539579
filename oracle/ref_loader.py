"""TEST INFRASTRUCTURE — imports the UNMODIFIED reference on CPU (build container only).

/root/reference does not exist on the GPU box, so nothing under tests -m gpu, smoke() or
bench.py may call this module; it is used by oracle/make_golden.py (fixture generation) and
by the optional `-m "not gpu"` cross-checks that skip when the reference is absent.

Recipe: the reference computes its root from `realpath('.')` and needs the
substring 'FasterSeg' in it (search/operations.py:14-17), imports `thop` and `easydict`
(operations.py:8,11) which are not installed, and train/operations.py:36 calls np.load
without allow_pickle.  We copy the tree to a scratch dir named FasterSeg (reads only), stub
the two modules and chdir there.  No reference file is edited and none is copied into the repo.
"""
import contextlib
import functools
import importlib
import os
import shutil
import sys
import tempfile
import types

REFERENCE_ROOT = os.environ.get("FASTERSEG_REFERENCE", "/root/reference")

_REF_MODULES = ("operations", "slimmable_ops", "seg_oprs", "genotypes", "model_seg",
                "model_search", "architect", "utils", "utils.darts_utils", "utils.init_func")


def available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "train"))


def _install_stubs():
    if "thop" not in sys.modules:
        thop = types.ModuleType("thop")
        thop.profile = lambda *a, **k: (0, 0)
        sys.modules["thop"] = thop
    if "easydict" not in sys.modules:
        ed = types.ModuleType("easydict")

        class EasyDict(dict):
            __getattr__ = dict.__getitem__
            __setattr__ = dict.__setitem__
        ed.EasyDict = EasyDict
        sys.modules["easydict"] = ed


_scratch = None


def scratch_copy():
    """Copy of the reference under <tmp>/FasterSeg (created once per process)."""
    global _scratch
    if _scratch is None:
        base = tempfile.mkdtemp(prefix="fsref_")
        _scratch = os.path.join(base, "FasterSeg")
        shutil.copytree(REFERENCE_ROOT, _scratch)
    return _scratch


@contextlib.contextmanager
def reference(subdir="train"):
    """Context manager: cwd + sys.path set so `import operations, model_seg` gives the
    reference's modules from `subdir` ('train' | 'search' | 'latency').  Modules are purged
    on exit so a different subdir can be imported next."""
    if not available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    import numpy as np
    _install_stubs()
    root = scratch_copy()
    wd = os.path.join(root, subdir)
    old_cwd = os.getcwd()
    old_path = list(sys.path)
    old_np_load = np.load
    for m in _REF_MODULES:
        sys.modules.pop(m, None)
    os.chdir(wd)
    sys.path.insert(0, wd)
    np.load = functools.partial(old_np_load, allow_pickle=True)
    try:
        yield wd
    finally:
        np.load = old_np_load
        os.chdir(old_cwd)
        sys.path[:] = old_path
        for m in _REF_MODULES:
            sys.modules.pop(m, None)


def import_ref(name):
    return importlib.import_module(name)
