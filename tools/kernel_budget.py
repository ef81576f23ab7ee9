"""Registers, LDS and scratch of every gfx950 kernel in the built library (the code objects' own metadata).
usage: python tools/kernel_budget.py [libpolyfuzz_hip.so] [name filter]

Why: round 3 found K7 and K3 sitting ON occupancy steps -- 256 B of LDS more cost K7 a workgroup per CU and 5 %, K3 4.4 %
(DESIGN.md, K7 / "What comes next") -- and nothing but this metadata says when an edit crosses one.
tests/test_kernel_budget_cpu.py holds the budgets of the kernels where it matters."""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"


def kernel_metadata(lib):
    """{mangled kernel name: {vgpr, sgpr, lds (static bytes), scratch (bytes per lane)}}"""
    out = {}
    with tempfile.TemporaryDirectory() as d:
        fat = os.path.join(d, "fat.bin")
        subprocess.check_call([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", lib, os.path.join(d, "copy.so")])
        blob = open(fat, "rb").read()
        # one offload bundle per translation unit, back to back
        starts = [m.start() for m in re.finditer(re.escape(b"__CLANG_OFFLOAD_BUNDLE__"), blob)]
        for k, a in enumerate(starts):
            piece, co = os.path.join(d, f"b{k}.bin"), os.path.join(d, f"b{k}.co")
            with open(piece, "wb") as f:
                f.write(blob[a:starts[k + 1] if k + 1 < len(starts) else len(blob)])
            r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={piece}", f"--targets={TARGET}",
                                f"--output={co}"], capture_output=True)
            if r.returncode or not os.path.exists(co) or os.path.getsize(co) == 0:
                continue
            notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
            cur = {}
            for line in notes.splitlines():
                m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)$", line)
                if not m:
                    continue
                key, val = m.group(1), m.group(2).strip().strip("'")
                if key in ("group_segment_fixed_size", "private_segment_fixed_size", "sgpr_count", "vgpr_count"):
                    cur[{"group_segment_fixed_size": "lds", "private_segment_fixed_size": "scratch", "sgpr_count": "sgpr",
                         "vgpr_count": "vgpr"}[key]] = int(val)
                elif key == "name":
                    cur["name"] = val
                elif key == "wavefront_size":           # (the last key of a kernel's record)
                    if "name" in cur:
                        out[cur.pop("name")] = cur
                    cur = {}
    return out


def demangled(names):
    import shutil
    exe = shutil.which("c++filt")
    if not exe:
        return list(names)
    r = subprocess.run([exe], input="\n".join(names), capture_output=True, text=True)
    return r.stdout.splitlines() if r.returncode == 0 else list(names)


if __name__ == "__main__":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from polyfuzz_amd import _build
    lib = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].endswith(".so") else _build.LIB_PATH
    flt = [a for a in sys.argv[1:] if not a.endswith(".so")]
    md = kernel_metadata(lib)
    names = sorted(md)
    print(f"{'vgpr':>5} {'sgpr':>5} {'lds B':>7} {'scratch':>7}  kernel")
    for n, pretty in zip(names, demangled(names)):
        if flt and not any(f in pretty for f in flt):
            continue
        v = md[n]
        print(f"{v.get('vgpr', -1):5d} {v.get('sgpr', -1):5d} {v.get('lds', -1):7d} {v.get('scratch', -1):7d}  {pretty[:110]}")
