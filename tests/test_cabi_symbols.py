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


def test_header_is_plain_c_and_links(tmp_path):
    """include/humid_hip.h compiles as C99 (no C++ in the boundary) and a C program that takes the
    address of every declared entry point links against libhumid_hip.so"""
    import subprocess
    build.build_hip()
    names = declared_symbols()
    src = tmp_path / "abi.c"
    body = "\n".join("  p[%d] = (fn)%s;" % (i, n) for i, n in enumerate(names))
    src.write_text('#include "humid_hip.h"\n#include <stdio.h>\ntypedef void (*fn)(void);\n'
                   'int main(void) {\n  fn p[%d];\n%s\n'
                   '  humid_summary s; s.total = 0; (void)s;\n'
                   '  printf("%%u %%d\\n", humid_abi_version(), p[0] != 0);\n  return 0;\n}\n' % (len(names), body))
    exe = tmp_path / "abi"
    libdir = os.path.join(ROOT, "humid_amd")
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
           str(src), "-o", str(exe), "-L", libdir, "-lhumid_hip", "-Wl,-rpath," + libdir,
           "-Wl,-rpath-link,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    out = subprocess.check_output([str(exe)]).decode()
    assert out.split()[0] == "1"
