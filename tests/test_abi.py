"""The C-ABI library loads and exports every symbol include/evhip.h declares (no compute calls: no GPU here)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "evhip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(evh_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from evenvizion_amd import _lib
    _lib.build()
    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), "libevhip.so does not export %s" % n
    assert sorted(_lib.SIGNATURES) == names, "ctypes signature table and include/evhip.h disagree"
    assert lib.evh_version() == 100
    assert lib.evh_profile_stage_name(2) == b"fast"


def test_no_cpu_fallback_without_gpu():
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from evenvizion_amd import _lib, runtime
    with pytest.raises(_lib.EvhError):
        runtime.get_context(400, 224)


def test_product_never_imports_oracle():
    bad = []
    for d, _, files in os.walk(os.path.join(ROOT, "evenvizion_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                t = open(os.path.join(d, f), errors="ignore").read()
                if re.search(r"^\s*(from|import)\s+oracle\b|#include\s+\"[^\"]*oracle/", t, flags=re.M):
                    bad.append(f)
    assert not bad, bad
