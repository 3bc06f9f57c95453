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
HIP_HEADERS = ["hutk_internal.h", "hutk_seam2.h", "hutk_kdev.h", "hutk_device.h", "hutk_classify.h", "hutk_lab.h", os.path.join(INCLUDE, "hutoken_amd.h")]

LIB_HIP = os.path.join(LIBDIR, "libhutoken_amd.so")
LIB_SYNTH = os.path.join(LIBDIR, "libhutk_synth.so")
LIB_PYSHIM = os.path.join(LIBDIR, "_hutoken_amd" + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.exists(s) and os.path.getmtime(s) > t for s in sources)


class _BuildLock:
    """One builder at a time per target (the ranks of a torchrun job import the package together), and a target that is
    never seen half written: every compiler writes a temporary file that is renamed over the target."""

    def __init__(self, target):
        self.path = target + ".lock"
        self.f = None

    def __enter__(self):
        import fcntl
        os.makedirs(os.path.dirname(self.path), exist_ok=True)
        self.f = open(self.path, "w")
        fcntl.flock(self.f, fcntl.LOCK_EX)
        return self

    def __exit__(self, *exc):
        import fcntl
        fcntl.flock(self.f, fcntl.LOCK_UN)
        self.f.close()


def _compile(cmd, target):
    tmp = "%s.tmp%d" % (target, os.getpid())
    try:
        subprocess.check_call([tmp if a == target else a for a in cmd])
        os.replace(tmp, target)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)


def _hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the HIP extension cannot be built")


def build_synth(force=False):
    src = os.path.join(CSRC, "hutk_synth.c")
    if force or _stale(LIB_SYNTH, [src]):
        with _BuildLock(LIB_SYNTH):
            if force or _stale(LIB_SYNTH, [src]):  # (another process may have built it while this one waited)
                _compile(["gcc", "-O2", "-fPIC", "-shared", "-std=gnu11", "-Wall", "-o", LIB_SYNTH, src, "-lpthread"], LIB_SYNTH)
    return LIB_SYNTH


def build_hip(force=False, extra_flags=()):
    srcs = [os.path.join(CSRC, s) for s in HIP_SOURCES]
    deps = srcs + [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HIP_HEADERS]
    if force or _stale(LIB_HIP, deps):
        with _BuildLock(LIB_HIP):
            if force or _stale(LIB_HIP, deps):
                cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                       "-Wall", "-Wno-unused-result", "-I" + INCLUDE, "-I" + CSRC,
                       *extra_flags, "-o", LIB_HIP, *srcs, "-lpthread"]
                _compile(cmd, LIB_HIP)
    return LIB_HIP


def build_pyshim(force=False):
    src = os.path.join(CSRC, "pyshim", "_hutoken_amd.c")
    if not os.path.exists(src):
        return None
    deps = [src, os.path.join(INCLUDE, "hutoken_amd.h")]
    if force or _stale(LIB_PYSHIM, deps):
        with _BuildLock(LIB_PYSHIM):
            if force or _stale(LIB_PYSHIM, deps):
                inc = sysconfig.get_paths()["include"]
                _compile(["gcc", "-O2", "-fPIC", "-shared", "-std=gnu11", "-Wall",
                          "-I" + inc, "-I" + INCLUDE, "-o", LIB_PYSHIM, src,
                          "-L" + LIBDIR, "-lhutoken_amd", "-Wl,-rpath,$ORIGIN"], LIB_PYSHIM)
    return LIB_PYSHIM


def build_all(force=False):
    build_synth(force)
    build_hip(force)
    build_pyshim(force)


if __name__ == "__main__":
    import sys
    build_all(force="--force" in sys.argv)
    print("built:", LIB_HIP, LIB_SYNTH)
