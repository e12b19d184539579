"""CPU-side checks of the boundary: the shared library loads and exports every symbol include/fasterseg_hip.h declares
(no compute is launched), the ctypes descriptors mirror the C structs, and the product never touches the oracle."""
import ctypes
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "fasterseg_hip.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fs_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from fasterseg_amd import _lib, build
    build.build(verbose=False)
    handle = _lib.lib()
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(handle, n), "libfasterseg_hip.so does not export " + n
    assert sorted(_lib.ALL_SYMBOLS) == names, "ctypes binding table and header disagree"
    assert handle.fs_version() >= 100
    assert handle.fs_packed_weight_elems(19, 1, 1, 128) == 19 * 128


def test_descriptor_layouts_match_header():
    from fasterseg_amd._lib import ConvDesc, ResizeDesc
    text = open(HEADER).read()
    conv = re.search(r"typedef struct fs_conv_desc \{(.*?)\} fs_conv_desc;", text, re.S).group(1)
    conv = re.sub(r"/\*.*?\*/", "", conv, flags=re.S)
    fields = [f.strip() for decl in re.findall(r"int ([^;]+);", conv) for f in decl.split(",")]
    assert fields == [n for n, _ in ConvDesc._fields_]
    rs = re.search(r"typedef struct fs_resize_desc \{(.*?)\} fs_resize_desc;", text, re.S).group(1)
    rs = re.sub(r"/\*.*?\*/", "", rs, flags=re.S)
    fields = [f.strip() for decl in re.findall(r"int ([^;]+);", rs) for f in decl.split(",")]
    assert fields == [n for n, _ in ResizeDesc._fields_]
    assert ctypes.sizeof(ConvDesc) == 4 * len(ConvDesc._fields_)


def test_invalid_arguments_return_status_not_abort():
    """Error convention of the boundary: status code + message, never a crash (validated before any launch)."""
    from fasterseg_amd import _lib
    handle = _lib.lib()
    d = _lib.ConvDesc(1, 8, 8, 8, 8, 5, 5, 1, 2, 8, 8, 8, 8, 0, 0)         # 5x5 filter
    buf = (ctypes.c_char * 64)()
    status = handle.fs_conv2d_fwd(None, ctypes.byref(d), buf, buf, None, None, buf, None)
    assert status != 0 and b"5x5" in handle.fs_last_error()
    assert handle.fs_conv2d_fwd(None, None, None, None, None, None, None, None) == 1


def test_product_does_not_import_the_oracle():
    """The oracle is test infrastructure: nothing under fasterseg_amd/ may import, call or read oracle/ or tests/."""
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "fasterseg_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                if re.search(r"^\s*(from|import)\s+(oracle|tests)\b", src, re.M) or "oracle/" in src.replace("the oracle", ""):
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad


def test_no_compat_layers_in_kernels():
    for f in os.listdir(os.path.join(ROOT, "fasterseg_amd", "csrc")):
        if f.endswith((".hip", ".h", ".cpp")):
            src = open(os.path.join(ROOT, "fasterseg_amd", "csrc", f)).read()
            assert "__HIP_PLATFORM" not in src and "cuda_runtime" not in src and "hipify" not in src.lower(), f


def test_missing_library_fails_loudly(tmp_path):
    code = ("import fasterseg_amd._lib as L, sys; L.LIB_PATH='/nonexistent/libfasterseg_hip.so'; L._lib=None\n"
            "try:\n    L.lib()\nexcept ImportError as e:\n    print('LOUD', e)\n")
    out = subprocess.run(["python", "-c", code], capture_output=True, text=True, cwd=ROOT)
    assert "LOUD" in out.stdout and "no CPU/eager fallback" in out.stdout


def test_integration_doc_binds_the_same_conv_descriptor():
    """INTEGRATION.md section 2 shows a maintainer how to bind fs_conv_desc by hand: its field list must be the header's
    (a short struct makes the kernel read stack garbage as filter strides)."""
    import re
    from fasterseg_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    block = re.search(r"class fs_conv_desc\(ctypes.Structure\):.*?_fields_ = \[\(n, ctypes.c_int\) for n in \((.*?)\)\]", doc, re.S)
    assert block, "INTEGRATION.md no longer shows the fs_conv_desc binding"
    fields = re.findall(r'"(\w+)"', block.group(1))
    assert fields == [n for n, _ in _lib.ConvDesc._fields_]
    header = open(os.path.join(root, "include", "fasterseg_hip.h")).read()
    struct = re.search(r"typedef struct fs_conv_desc \{(.*?)\} fs_conv_desc;", header, re.S).group(1)
    struct = re.sub(r"/\*.*?\*/", "", struct, flags=re.S)
    names = [n.strip() for decl in re.findall(r"int ([^;]+);", struct) for n in decl.split(",")]
    assert names == fields
    sections = [int(m) for m in re.findall(r"^## (\d+)\.", doc, re.M)]
    assert sections == list(range(1, len(sections) + 1)), "section numbering of INTEGRATION.md"
    # the documented binding refuses a library of another ABI revision: the number it asserts must be the header's and the bindings'
    # (the doc said 206 while header and _lib.py said 208)
    doc_abi = re.search(r"assert lib\.fs_version\(\) == (\d+)", doc)
    header_abi = re.search(r"#define FS_ABI_VERSION (\d+)", header)
    assert doc_abi and header_abi and int(doc_abi.group(1)) == int(header_abi.group(1)) == _lib.EXPECTED_ABI


def test_product_does_not_reach_into_tests_or_fixtures():
    """Nothing under fasterseg_amd/ may open files of tests/ (golden fixtures are reference-held data for the checker)."""
    import re
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fasterseg_amd")
    bad = []
    for dirpath, _, files in os.walk(root):
        for fn in files:
            if fn.endswith(".py"):
                src = open(os.path.join(dirpath, fn)).read()
                for m in re.finditer(r"""["'](tests|golden)["']|tests[/\\]golden|latency_lut_1080ti""", src):
                    bad.append((fn, m.group(0)))
    assert not bad, bad


def test_library_reports_the_abi_the_bindings_expect():
    from fasterseg_amd import _lib
    h = _lib.lib()
    assert h.fs_version() == _lib.EXPECTED_ABI
    for which, struct in enumerate((_lib.ConvDesc, _lib.ResizeDesc, _lib.ZoomDesc, _lib.SgdTensor)):
        assert h.fs_struct_size(which) == __import__("ctypes").sizeof(struct)
    assert h.fs_struct_size(99) == -1


def test_measurement_scripts_parse_and_probes_are_not_in_the_product_build():
    """Round 5 hygiene: one GPU job script and one profiling script (shell syntax checked here); the product library
    carries no measurement-only conv_igemm2 instantiation (ablation builds ABL 1-4, 6- and 8-stage rings); no built probe is tracked."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for script in ("tools/gpu_job.sh", "tools/prof_round.sh", "tools/prof_step.sh"):
        r = subprocess.run(["bash", "-n", os.path.join(root, script)], capture_output=True, text=True)
        assert r.returncode == 0, (script, r.stderr)
    assert not [f for f in os.listdir(os.path.join(root, "tools")) if f.startswith("gpu_job_r0") or f.startswith("prof_r0")]
    src = open(os.path.join(root, "fasterseg_amd", "csrc", "conv_igemm2.hip")).read()
    guarded = src[src.index("#ifdef FS_BUILD_PROBES"):]
    assert "conv_igemm2_kernel<T, 2, 2, 1, 1, 1, 4, 1>" in guarded[:guarded.index("#endif")]           # the ablations live inside the guard
    assert "conv_igemm2_kernel<T, 2, 2, 1, 1, 1, 4, 1>" not in src[:src.index("#ifdef FS_BUILD_PROBES")]
    lib = os.path.join(root, "fasterseg_amd", "libfasterseg_hip.so")
    if os.path.exists(lib):
        syms = subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True).stdout
        # mangled names end in the last two template arguments <..., NSTAGE, ABL>: the product build has 3- / 4-stage rings with ABL = 0, or
        # 5 = ABL_X3, the fp32 form on the bf16 matrix cores (conv_igemm.h), which shares the template slot - never the ablations 1-4
        tails = [re.search(r"Li(\d)ELi(\d)EEEvNS_8ConvArgsE$", l) for l in syms.splitlines() if "conv_igemm2_kernelI" in l]
        assert tails and all(t is not None for t in tails)
        assert {(t.group(1), t.group(2)) for t in tails} <= {("3", "0"), ("4", "0"), ("3", "5"), ("4", "5")}, sorted({(t.group(1), t.group(2)) for t in tails})
        # ... and the grouped launches carry no ablation instantiation either (FS_IGEMM2_GROUP_ABL lives behind FS_BUILD_PROBES)
        gtails = [re.search(r"Li(\d)ELi(\d)EEEvNS_13ConvGroupArgsE$", l) for l in syms.splitlines() if "conv_igemm2_group_kernelI" in l]
        assert gtails and all(t is not None for t in gtails)
        assert {(t.group(1), t.group(2)) for t in gtails} <= {("4", "0"), ("4", "5")}, sorted({(t.group(1), t.group(2)) for t in gtails})
    tracked = subprocess.run(["git", "ls-files", "tools/probes"], capture_output=True, text=True, cwd=root).stdout.split()
    assert all(f.endswith(".hip") for f in tracked), tracked
