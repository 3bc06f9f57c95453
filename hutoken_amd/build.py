"""In-tree build of the native code (no JIT cache: the .so files travel with the
repo snapshot to the GPU box).

  hutoken_amd/lib/libhutoken_amd.so   C-ABI + HIP kernels for gfx950 (hipcc)
  hutoken_amd/lib/libhutk_synth.so    synthetic corpus generator (gcc)
  hutoken_amd/lib/_hutoken_amd*.so    CPython shim mirroring the reference's
                                      src/lib.c method table (gcc)
"""
import os
import shutil
import subprocess
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
INCLUDE = os.path.join(ROOT, "include")

HIP_SOURCES = ["hutk_loader.cpp", "hutk_api.cpp", "hutk_kernels.hip", "hutk_ptiles.hip", "hutk_decode.hip"]
HIP_HEADERS = ["hutk_internal.h", "hutk_kdev.h", "hutk_device.h", "hutk_classify.h", "hutk_lab.h", os.path.join(INCLUDE, "hutoken_amd.h")]

LIB_HIP = os.path.join(LIBDIR, "libhutoken_amd.so")
LIB_SYNTH = os.path.join(LIBDIR, "libhutk_synth.so")
LIB_PYSHIM = os.path.join(LIBDIR, "_hutoken_amd" + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.exists(s) and os.path.getmtime(s) > t for s in sources)


def _hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the HIP extension cannot be built")


def build_synth(force=False):
    src = os.path.join(CSRC, "hutk_synth.c")
    if force or _stale(LIB_SYNTH, [src]):
        os.makedirs(LIBDIR, exist_ok=True)
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-std=gnu11", "-Wall",
                               "-o", LIB_SYNTH, src, "-lpthread"])
    return LIB_SYNTH


def build_hip(force=False, extra_flags=()):
    srcs = [os.path.join(CSRC, s) for s in HIP_SOURCES]
    deps = srcs + [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HIP_HEADERS]
    if force or _stale(LIB_HIP, deps):
        os.makedirs(LIBDIR, exist_ok=True)
        cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
               "-Wall", "-Wno-unused-result", "-I" + INCLUDE, "-I" + CSRC,
               *extra_flags, "-o", LIB_HIP, *srcs, "-lpthread"]
        subprocess.check_call(cmd)
    return LIB_HIP


def build_pyshim(force=False):
    src = os.path.join(CSRC, "pyshim", "_hutoken_amd.c")
    if not os.path.exists(src):
        return None
    if force or _stale(LIB_PYSHIM, [src, os.path.join(INCLUDE, "hutoken_amd.h")]):
        os.makedirs(LIBDIR, exist_ok=True)
        inc = sysconfig.get_paths()["include"]
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-std=gnu11", "-Wall",
                               "-I" + inc, "-I" + INCLUDE, "-o", LIB_PYSHIM, src,
                               "-L" + LIBDIR, "-lhutoken_amd", "-Wl,-rpath,$ORIGIN"])
    return LIB_PYSHIM


def build_all(force=False):
    build_synth(force)
    build_hip(force)
    build_pyshim(force)


if __name__ == "__main__":
    import sys
    build_all(force="--force" in sys.argv)
    print("built:", LIB_HIP, LIB_SYNTH)
