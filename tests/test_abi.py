"""The C-ABI library builds, loads without a GPU and exports exactly what include/hhsr.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "include", "hhsr.h")


def header_functions():
    src = open(HDR).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(hhsr_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported_and_bound():
    from handheld_super_resolution import _lib

    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__

        __graft_entry__.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = header_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in hhsr.h but not exported"
    # the Python binding table covers the header one-to-one
    assert sorted(_lib.exported_symbols()) == names
    assert b"gfx950" in ctypes.cast(lib.hhsr_version, ctypes.CFUNCTYPE(ctypes.c_char_p))()


def test_argument_errors_do_not_touch_the_gpu():
    """Argument validation happens on the host before any HIP call: usable without a device."""
    from handheld_super_resolution import _lib

    lib = _lib.load()
    assert lib.hhsr_divide(None, None, 4, None) == -1
    assert b"invalid argument" in lib.hhsr_last_error()
    with pytest.raises(RuntimeError, match="hhsr_add failed"):
        _lib.call("hhsr_add", None, None, 4, None)
    # ICA with an unsupported tile size reports the reference's NotImplementedError text
    rc = lib.hhsr_cov_from_raw(None, 4, 4, 4, None, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1, None)
    assert rc == -1


def test_no_cpu_fallback():
    import numpy as np
    import torch

    import handheld_super_resolution as hsr

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    cfg = hsr.default_config()
    cfg.exif = {"cfa_pattern": [[0, 1], [1, 2]], "white_balance": [1, 1, 1]}
    cfg.noise_model.update({"std_curve": [1.0] * 1001, "diff_curve": [1.0] * 1001})
    with pytest.raises(RuntimeError):
        hsr.main(np.zeros((64, 64), np.float32), np.zeros((1, 64, 64), np.float32), cfg)


def test_product_does_not_import_oracle():
    """The product path must not route through the oracle (or any CPU fallback)."""
    pkg = os.path.join(ROOT, "handheld-multi-frame-super-resolution_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", txt, flags=re.M), f


def _build_c_demo(tmp_path):
    """examples/hhsr_c_demo.c with plain gcc -std=c11: include/hhsr.h is valid C and every symbol links."""
    import shutil
    import subprocess

    if shutil.which("gcc") is None or not os.path.isdir("/opt/rocm/include"):
        pytest.skip("gcc / ROCm headers not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "handheld-multi-frame-super-resolution_amd", "handheld_super_resolution")
    exe = str(tmp_path / "hhsr_c_demo")
    cmd = ["gcc", "-std=c11", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
           "-I" + os.path.join(root, "include"), os.path.join(root, "examples", "hhsr_c_demo.c"), "-L" + libdir,
           "-lhhsr_hip", "-L/opt/rocm/lib", "-lamdhip64", "-lm", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib",
           "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_c_client_compiles_and_links(tmp_path):
    _build_c_demo(tmp_path)


@pytest.mark.gpu
def test_c_client_runs(tmp_path):
    """The plain-C client (no Python, no torch) drives the library on the GPU and checks its results itself."""
    import subprocess

    r = subprocess.run([_build_c_demo(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "bit-exact" in r.stdout and "rejected factor 3" in r.stdout


# ---- AddressSanitizer build of the host shim ---------------------------------------------------------------
CLANG = "/opt/rocm/lib/llvm/bin/clang"


def _asan_env():
    import glob

    rt = glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so")
    if not rt or not os.path.exists(CLANG):
        pytest.skip("ROCm clang / ASan runtime not available")
    # protect_shadow_gap=0: the HIP runtime maps memory inside ASan's shadow gap; leaks: the runtime's own singletons
    return dict(os.environ, ASAN_OPTIONS="protect_shadow_gap=0:detect_leaks=0:abort_on_error=0",
                LD_LIBRARY_PATH=os.path.dirname(rt[0]) + ":" + os.environ.get("LD_LIBRARY_PATH", ""))


ASAN_DIR = os.path.join(ROOT, "handheld-multi-frame-super-resolution_amd", "build", "asan")


@pytest.fixture(scope="module")
def asan_lib(tmp_path_factory):
    """The ASan build of the library: the one __graft_entry__.build() left in-tree (it travels to the GPU box with the
    snapshot: rebuilding 14 translation units there took 140 s of the suite) when it is newer than every source, else
    built now.  The drivers are compiled into a scratch directory either way."""
    import importlib.util

    csrc = os.path.join(ROOT, "handheld-multi-frame-super-resolution_amd", "csrc")
    newest = max([os.path.getmtime(os.path.join(csrc, f)) for f in os.listdir(csrc)] + [os.path.getmtime(HDR)])
    pre = os.path.join(ASAN_DIR, "libhhsr_hip_asan.so")
    if os.path.exists(pre) and os.path.getmtime(pre) >= newest:
        return ASAN_DIR, pre
    out = str(tmp_path_factory.mktemp("asan"))
    spec = importlib.util.spec_from_file_location("hhsr_build_asan", os.path.join(
        ROOT, "handheld-multi-frame-super-resolution_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return out, mod.build_asan(out)


def _asan_exe(out, src, extra=(), libs=()):
    import subprocess

    import tempfile

    exe = os.path.join(tempfile.mkdtemp(prefix="hhsr_asan_"), os.path.splitext(os.path.basename(src))[0] + "_asan")
    cmd = [CLANG, "-std=c11", "-Wall", "-g", "-fsanitize=address", "-shared-libsan", "-I" + os.path.join(ROOT, "include"),
           *extra, src, "-L" + out, "-lhhsr_hip_asan", *libs, "-Wl,-rpath," + out, "-Wl,-rpath,/opt/rocm/lib", "-lm", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


@pytest.mark.timeout(900)
def test_asan_host_shim_argument_paths(asan_lib):
    """Every translation unit rebuilt with -fsanitize=address (host side; device code untouched) and driven by
    tests/asan_host_driver.c: the HOST arrays of the C ABI (pointer tables of the batched entry points, CFA bytes, tap /
    white-balance vectors) are exact-length heap allocations, the calls pass every table check and fail a later
    validation — no GPU needed.  ASan reports abort the driver; a deliberately short table is caught (checked once by
    hand: heap-buffer-overflow READ in hhsr_gauss_decimate_batch)."""
    import subprocess

    env = _asan_env()
    out, _ = asan_lib
    exe = _asan_exe(out, os.path.join(ROOT, "tests", "asan_host_driver.c"))
    r = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and "all argument paths returned their error codes" in r.stdout, r.stdout + r.stderr[-3000:]
    assert "AddressSanitizer" not in r.stderr


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_asan_c_client_runs_on_the_gpu(asan_lib):
    """The plain-C client of examples/ against the ASan build of the library, on the GPU: launches included."""
    import subprocess

    env = _asan_env()
    out, _ = asan_lib
    exe = _asan_exe(out, os.path.join(ROOT, "examples", "hhsr_c_demo.c"),
                    extra=("-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include"), libs=("-L/opt/rocm/lib", "-lamdhip64"))
    r = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "bit-exact" in r.stdout, r.stdout + r.stderr[-3000:]
    assert "AddressSanitizer" not in r.stderr


def test_library_is_newer_than_its_sources():
    """The in-tree .so travels to the GPU box as it is: a source edited after the last build would be tested stale."""
    from handheld_super_resolution import _lib

    csrc = os.path.join(ROOT, "handheld-multi-frame-super-resolution_amd", "csrc")
    newest = max(os.path.getmtime(os.path.join(csrc, f)) for f in os.listdir(csrc))
    newest = max(newest, os.path.getmtime(HDR))
    if os.path.getmtime(_lib.LIB_PATH) < newest:
        import __graft_entry__

        __graft_entry__.build()
    assert os.path.getmtime(_lib.LIB_PATH) >= newest
