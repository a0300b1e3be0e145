"""Build libhhsr_hip.so (gfx950) in-tree with hipcc.  Usage: python build.py [--force]"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "handheld_super_resolution", "libhhsr_hip.so")
OBJ = os.path.join(HERE, "build")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"

# IEEE maths everywhere (no -ffast-math).  FMA contraction is disabled for the stages that take
# discrete decisions on float32 sums (argmin, rounding) so they associate exactly like the oracle;
# the merge keeps hipcc's default contraction (its weights are float64 / tolerance-checked).
COMMON = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
SOURCES = {
    "hhsr_api.hip": ["-ffp-contract=off"],
    "hhsr_pyramid.hip": ["-ffp-contract=off"],
    "hhsr_align.hip": ["-ffp-contract=off", "-fno-slp-vectorize"],  # (SLP off: level 0 172.5 -> 167.2 us, coarse levels 55.7 -> 54.5)
    "hhsr_kernels.hip": ["-ffp-contract=off", "-fno-slp-vectorize"],
    "hhsr_robustness.hip": ["-ffp-contract=off", "-fno-slp-vectorize"],
    "hhsr_merge.hip": [],       # C ABI + kernel choice, per-pixel kernels, border bands
    "hhsr_merge_tile.hip": [],  # first-generation LDS tile kernels
    "hhsr_merge_x2.hip": [],    # k_merge_x2
    "hhsr_merge_xs.hip": [],    # k_merge_xs<3>
    "hhsr_grey.hip": ["-ffp-contract=off"],
    # (the SLP vectoriser packs the complex butterflies into v_pk_* instructions: measured 121 -> 98 us per launch of the
    # row kernels, 94 -> 89 us of the column kernel at 12 MP x 3-4 frames — tools/ab.sh --kernels "k_rows|k_cols" slp@fft_slp default)
    "hhsr_fft.hip": ["-fno-slp-vectorize"],
    "hhsr_io.hip": ["-ffp-contract=off"],
    "hhsr_post.hip": ["-ffp-contract=off"],
}


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, "hhsr_common.h"), os.path.join(CSRC, "hhsr_fft.h"), os.path.join(CSRC, "hhsr_fft_bfly.h"), os.path.join(CSRC, "hhsr_merge.h"), os.path.join(HERE, "..", "include", "hhsr.h"), __file__]
    jobs = []
    objs = []
    for src, extra in SOURCES.items():
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + headers):
            jobs.append([HIPCC, *COMMON, *extra, "-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    with ThreadPoolExecutor(max_workers=max(1, min(len(jobs), os.cpu_count() or 1))) as ex:
        list(ex.map(run, jobs))
    if jobs or force or _stale(OUT, objs):
        run([HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", OUT, *objs, "-L/opt/rocm/lib", "-lhipfft",
             "-Wl,-rpath,/opt/rocm/lib"])
    return OUT


def build_asan(out_dir, verbose=False):
    """AddressSanitizer build of the HOST side of the library: every translation unit with
    -fsanitize=address -fno-gpu-sanitize (the device code is untouched), linked against the shared ASan runtime, into
    `out_dir`/libhhsr_hip_asan.so.  Used by tests/test_abi.py with a C driver built by the same compiler; not shipped."""
    os.makedirs(out_dir, exist_ok=True)
    flags = ["--offload-arch=" + ARCH, "-O1", "-g", "-std=c++17", "-fPIC", "-fsanitize=address", "-fno-gpu-sanitize",
             "-shared-libsan", "-fno-omit-frame-pointer"]
    objs, jobs = [], []
    for src, extra in SOURCES.items():
        o = os.path.join(out_dir, src.replace(".hip", ".o"))
        objs.append(o)
        jobs.append([HIPCC, *flags, *extra, "-c", os.path.join(CSRC, src), "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    with ThreadPoolExecutor(max_workers=max(1, min(len(jobs), os.cpu_count() or 1))) as ex:
        list(ex.map(run, jobs))
    out = os.path.join(out_dir, "libhhsr_hip_asan.so")
    run([HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-fsanitize=address", "-shared-libsan", "-o", out, *objs,
         "-L/opt/rocm/lib", "-lhipfft", "-Wl,-rpath,/opt/rocm/lib"])
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
