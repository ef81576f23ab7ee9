"""Registers and LDS of the kernels that sit on occupancy steps, read from the built library's code objects
(tools/kernel_budget.py).  Round 3 measured what a step costs: 256 B of LDS more took a workgroup per CU from K7 (5 %)
and from K3 (4.4 %); an edit that crosses one fails here, on the CPU build box, before anybody times anything."""
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))


@pytest.fixture(scope="module")
def kernels():
    import kernel_budget
    from polyfuzz_amd import _build
    for exe in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf"):
        if not os.path.exists(os.path.join(kernel_budget.LLVM, exe)):
            pytest.skip(f"{exe} not in {kernel_budget.LLVM}")
    md = kernel_budget.kernel_metadata(_build.build())
    assert len(md) > 50, "the library's device code could not be read"
    pretty = dict(zip(md, kernel_budget.demangled(list(md))))
    return {pretty[k]: v for k, v in md.items()}


def _one(kernels, prefix):
    hits = [v for k, v in kernels.items() if k.startswith(prefix)]
    assert len(hits) == 1, (prefix, [k for k in kernels if prefix[:20] in k])
    return hits[0]


def test_k7_one_word_class_keeps_16_workgroups_per_cu(kernels):
    """64-lane workgroups: 128 VGPRs = 4 waves per SIMD; LDS per workgroup = static + the match table (143 symbols of config
    3's titles: 3432 B) + the byte scratch columns (4096 B) must stay within 9872 B (the most that was measured to hold 16)."""
    for scorer in ("-1", "0"):                            # (the scorer at run time; WRatio's own instance)
        k = _one(kernels, f"void pfz::k7_fuzz_kernel<1, {scorer}>(")
        assert k["vgpr"] <= 128, k
        assert k["lds"] + 3432 + 4096 <= 9872, k
        assert k["scratch"] <= 128, k                  # (a few spilled dwords, none in a hot loop: DESIGN.md)


def test_k7_longer_classes(kernels):
    for scorer in ("-1", "0"):
        assert _one(kernels, f"void pfz::k7_fuzz_kernel<2, {scorer}>(")["vgpr"] <= 170          # 3 waves per SIMD
        assert _one(kernels, f"void pfz::k7_fuzz_kernel<4, {scorer}>(")["vgpr"] <= 256          # 2


def test_k3_headline_kernel_keeps_18_workgroups_per_cu(kernels):
    """2048 accumulator columns + the 96-key candidate buffer: 8960 B; one more 256-B step is a workgroup per CU less."""
    k = _one(kernels, "void pfz::k3_cossim_topn_kernel<2048, 96, false>(")
    assert k["lds"] <= 8960 and k["vgpr"] <= 72 and k["scratch"] == 0, k


def test_k3_symmetric_passes_keep_their_workgroups_per_cu(kernels):
    """k3_symmetric.hip: pass 1 carries the 128-entry staging buffer of the candidates it hands over (9984 B: 16 workgroups per
    CU, and its registers must allow 4 waves per SIMD); passes 0 and 2 are the row-major kernel's size (18 per CU)."""
    p1 = _one(kernels, "void pfz::k3_sym_kernel<2048, 1>(")
    assert p1["lds"] <= 10240 and p1["vgpr"] <= 96 and p1["scratch"] == 0, p1
    for mode in (0, 2):
        k = _one(kernels, f"void pfz::k3_sym_kernel<2048, {mode}>(")
        assert k["lds"] <= 9102 and k["vgpr"] <= 96 and k["scratch"] == 0, k


def test_no_scratch_outside_k7(kernels):
    """register spills (or arrays the compiler could not keep in registers) anywhere else would be news"""
    bad = {k: v["scratch"] for k, v in kernels.items() if v.get("scratch", 0) and "k7_" not in k}
    assert not bad, bad
