"""Builds the in-tree HIP shared libraries for gfx950 (hipcc cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libipkgpu.so")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared",
         "-ffp-contract=off", "-fno-fast-math", "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-pthread"]


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def build(force=False, verbose=False):
    srcs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC))]
    hdrs = [os.path.join(HERE, "..", "include", "ipkgpu.h")]
    if force or _stale(LIB, srcs + hdrs):
        units = [s for s in srcs if s.endswith((".hip", ".cpp"))]
        cmd = [HIPCC] + FLAGS + ["-o", LIB] + units
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    return LIB


VARIANTS = {"execassert": ["-DIPK_EXEC_ASSERT=1"]}          # diagnostic builds of the same library (tests only)


def build_variant(name, force=False, verbose=False):
    """ipk_amd/_variants/v_<name>.so: the library with extra defines; loaded through IPKGPU_LIB by the test that needs it."""
    vdir = os.path.join(HERE, "_variants")
    os.makedirs(vdir, exist_ok=True)
    target = os.path.join(vdir, f"v_{name}.so")
    srcs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC))]
    if force or _stale(target, srcs + [os.path.join(HERE, "..", "include", "ipkgpu.h"), os.path.abspath(__file__)]):
        units = [s for s in srcs if s.endswith((".hip", ".cpp"))]
        cmd = [HIPCC] + FLAGS + VARIANTS[name] + ["-o", target] + units
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    return target


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    for v in VARIANTS:
        print(build_variant(v, force="--force" in sys.argv, verbose=True))
