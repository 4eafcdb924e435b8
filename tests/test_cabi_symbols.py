"""CPU-side checks of the drop-in boundary: the library builds/loads and exports every symbol
include/humid_hip.h declares; without a GPU it refuses to run (no CPU fallback)."""
import os
import re

import pytest

import humid_amd
from humid_amd import _lib, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "humid_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(humid_[a-z_0-9]+)\s*\(", text)))


def test_library_builds_and_exports_every_declared_symbol():
    build.build_hip()
    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 12
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(_lib.SYMBOLS) == names
    assert lib.humid_abi_version() == 1


def test_no_cpu_fallback_without_gpu():
    lib = _lib.load()
    if lib.humid_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(humid_amd.HumidError) as ei:
        humid_amd.Dedup()
    assert "no CPU fallback" in str(ei.value)


def test_product_package_never_imports_the_oracle():
    for d, _, fs in os.walk(os.path.join(ROOT, "humid_amd")):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cc", ".cpp", ".hpp")):
                src = open(os.path.join(d, f)).read()
                assert "pyoracle" not in src and "humid_oracle" not in src and "liboracle" not in src, f
