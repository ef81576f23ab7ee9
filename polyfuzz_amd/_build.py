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


def build(force=False, verbose=False):
    """Compile every HIP source of the package into polyfuzz_amd/libpolyfuzz_hip.so (and the host helper)."""
    build_host_helpers(force, verbose)
    if not force and not is_stale():
        return LIB_PATH
    rocm_lib = "/opt/rocm/lib"
    cmd = [_hipcc()] + HIPCC_FLAGS + ["-I", INCLUDE, "-I", CSRC] + sources() + \
          ["-L", rocm_lib, "-lrccl", "-Wl,-rpath," + rocm_lib, "-o", LIB_PATH + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=REPO)
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
