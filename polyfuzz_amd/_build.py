"""Build libpolyfuzz_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import glob
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(_HERE)
CSRC = os.path.join(_HERE, "csrc")
INCLUDE = os.path.join(REPO, "include")
LIB_PATH = os.path.join(_HERE, "libpolyfuzz_hip.so")

HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
               "-Wall", "-Wno-unused-function"]


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libpolyfuzz_hip.so can only be built with the ROCm toolchain")
    return exe


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def is_stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(INCLUDE, "*.h"))
    return any(os.path.getmtime(d) > t for d in deps)


PACK_SRC = os.path.join(_HERE, "csrc_host", "_pack.c")
PACK_PATH = os.path.join(_HERE, "_pack.so")


def build_host_helpers(force=False, verbose=False):
    """Compile the CPython helper polyfuzz_amd/_pack.so (string packing for the upload, object-column gathers
    of the result frame).  It is glue, not compute: when it is missing the pure-Python twins run."""
    if not force and os.path.exists(PACK_PATH) and os.path.getmtime(PACK_PATH) >= os.path.getmtime(PACK_SRC):
        return PACK_PATH
    import sysconfig
    cmd = ["gcc", "-O3", "-msse4.1", "-shared", "-fPIC", "-I", sysconfig.get_paths()["include"], PACK_SRC, "-lm", "-lpthread", "-o", PACK_PATH + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=REPO)
    os.replace(PACK_PATH + ".tmp", PACK_PATH)
    return PACK_PATH


OBJ_ROOT = os.path.join(CSRC, "_obj")


def _compile_flags():
    return [f for f in HIPCC_FLAGS if f != "-shared"] + os.environ.get("PFZ_EXTRA_HIPCC_FLAGS", "").split()


def _obj_dir():
    """csrc/_obj/<hash of the compile flags>: objects built under other flags (a -D variant through
    PFZ_EXTRA_HIPCC_FLAGS) are never linked into this build, nor this build's objects into theirs."""
    import hashlib
    return os.path.join(OBJ_ROOT, hashlib.sha256(" ".join(_compile_flags()).encode()).hexdigest()[:12])


def _compile_one(src, force, verbose):
    """one translation unit -> csrc/_obj/<flags>/<name>.o (skipped when newer than the source and every header)"""
    obj = os.path.join(_obj_dir(), os.path.basename(src)[:-4] + ".o")
    deps = [src] + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(INCLUDE, "*.h"))
    if not force and os.path.exists(obj) and all(os.path.getmtime(d) <= os.path.getmtime(obj) for d in deps):
        return obj, False
    flags = _compile_flags()
    cmd = [_hipcc()] + flags + ["-I", INCLUDE, "-I", CSRC, "-c", src, "-o", obj + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=REPO)
    os.replace(obj + ".tmp", obj)
    return obj, True


def build(force=False, verbose=False, out=None):
    """Compile every HIP source of the package (one object per source, in parallel, re-used while up to date) and link
    them into polyfuzz_amd/libpolyfuzz_hip.so; also builds the host helper.  `out`: link to another path (variants)."""
    build_host_helpers(force, verbose)
    lib = out or LIB_PATH
    if not force and out is None and not is_stale():
        return lib
    import concurrent.futures as cf
    os.makedirs(_obj_dir(), exist_ok=True)
    with cf.ThreadPoolExecutor(max(1, min(8, os.cpu_count() or 1))) as ex:
        objs = [o for o, _ in ex.map(lambda s: _compile_one(s, force, verbose), sources())]
    rocm_lib = "/opt/rocm/lib"
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + \
          ["-L", rocm_lib, "-lrccl", "-Wl,-rpath," + rocm_lib, "-o", lib + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=REPO)
    os.replace(lib + ".tmp", lib)
    return lib


if __name__ == "__main__":
    print(build(force=True, verbose=True))
