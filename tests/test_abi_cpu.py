"""The C-ABI shared library: loads without a GPU, exports every symbol include/polyfuzz_hip.h
declares, and fails loudly (no CPU fallback) when asked to compute without a device."""
import ctypes
import os
import re

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(REPO, "include", "polyfuzz_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pfz_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    from polyfuzz_amd import _build, _lib
    if _build.is_stale():
        _build.build()
    return _lib.load()


def test_library_exports_every_declared_symbol(lib):
    from polyfuzz_amd import _lib
    declared = _declared_symbols()
    assert len(declared) >= 40
    so = ctypes.CDLL(_lib.lib_path())
    missing = [s for s in declared if not hasattr(so, s)]
    assert not missing, missing
    assert sorted(_lib.SIGNATURES) == declared     # the ctypes table binds exactly the header
    assert lib.pfz_version() == 100


def test_no_silent_cpu_fallback(lib):
    import polyfuzz_amd
    from polyfuzz_amd import _lib
    if polyfuzz_amd.device_count() > 0:
        pytest.skip("a GPU is visible: the no-device failure path cannot be exercised")
    with pytest.raises(_lib.PfzNoDevice, match="no CPU fallback"):
        polyfuzz_amd.Context(0)
    from polyfuzz_amd.models import TFIDF, EditDistance
    with pytest.raises(_lib.PfzNoDevice):
        TFIDF().match(["apple"], ["apples"])
    with pytest.raises(_lib.PfzNoDevice):
        EditDistance().match(["apple"], ["apples"])


def test_product_code_never_imports_the_oracle():
    pkg = os.path.join(REPO, "polyfuzz_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(root, f), encoding="utf-8").read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", src, flags=re.M), f
                # the product never calls scikit-learn either (the name appears only in comments/docstrings
                # and as the reference's back-end name "sklearn")
                assert not re.search(r"^\s*(import|from)\s+sklearn\b", src, flags=re.M), f


def test_shipped_library_has_no_wrong_on_purpose_knobs():
    """the timing experiments that make results wrong on purpose (PFZ_K3_ABLATE, PFZ_K3_SYM_EXP,
    PFZ_K3_SYM_SOLO, PFZ_K7_EXP, the PFZ_K3_EXP what-ifs) exist in variant builds only (tools/build_variant.sh -DPFZ_EXPERIMENTS /
    -DPFZ_K3_EXP=n): the shipped library does not even contain their names, so no environment can switch them on."""
    from polyfuzz_amd import _build
    blob = open(_build.LIB_PATH, "rb").read()
    for knob in (b"PFZ_K3_ABLATE", b"PFZ_K3_SYM_EXP", b"PFZ_K3_SYM_SOLO", b"PFZ_K7_EXP", b"PFZ_K3_EXP"):
        assert knob not in blob, knob
    assert b"PFZ_K3_SYM_MIN" in blob            # (a tuning knob that does not change results is still there: the check reads the right file)
