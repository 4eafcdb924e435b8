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
    assert lib.humid_abi_version() == 5


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
    assert out.split()[0] == "5"


def test_pigeonhole_plan_never_exceeds_its_tables():
    """make_plan through humid_stage_plan_info with a NULL context (host arithmetic only).  The plan
    tables hold 20 combinations; d >= 20 has no plan with so few (C(d+1, 1) = d+1) and must fall back
    to the single all-pairs combination instead of writing past the tables (ADVICE round 1:
    make_plan(24, 20) returned 21 combinations)."""
    import ctypes as C
    lib = _lib.load()
    nc, pb = C.c_uint32(), C.c_uint32()
    for n in (24,):
        for m in (19, 20, 23, 24, 30):
            assert lib.humid_stage_plan_info(None, n, m, 3_000_000, C.byref(nc), C.byref(pb)) == 0
            assert (nc.value, pb.value) == ((20, 4) if m == 19 else (1, 0)), (n, m, nc.value, pb.value)
    for n in range(1, 65):
        for d in range(0, n + 3):
            for u in (10, 3_000_000, 1 << 40):
                assert lib.humid_stage_plan_info(None, n, d, u, C.byref(nc), C.byref(pb)) == 0
                assert 1 <= nc.value <= 20 and pb.value <= min(64, 2 * n), (n, d, u, nc.value, pb.value)
    # the metric configuration: two combinations of 12 nt
    assert lib.humid_stage_plan_info(None, 24, 1, 2_700_000, C.byref(nc), C.byref(pb)) == 0
    assert (nc.value, pb.value) == (2, 24)
    assert lib.humid_stage_plan_info(None, 65, 1, 10, C.byref(nc), C.byref(pb)) == -2     # unsupported


def test_every_kernel_guards_its_last_vector_register():
    """common.hip.h: HUMID_GUARD_LAST_VGPR() is the first statement of every __global__ function of
    the library (the platform defect of DESIGN.md section 3a overwrites a wave's last register)"""
    import glob
    n = 0
    for p in glob.glob(os.path.join(ROOT, "humid_amd", "csrc", "*.hip*")):
        src = open(p).read()
        for m in re.finditer(r"__global__", src):
            line_start = src.rfind("\n", 0, m.start()) + 1
            if src[line_start:m.start()].lstrip().startswith("//"):
                continue                                   # the word in a comment
            body = src.index("{", m.end())
            assert src[body + 1:body + 60].split(";")[0].strip() == "HUMID_GUARD_LAST_VGPR()", \
                (os.path.basename(p), src[m.start():m.start() + 80])
            n += 1
    assert n >= 60


LLVM_BIN = "/opt/rocm/lib/llvm/bin"


def gfx950_kernel_descriptors(so_path, tmp_path):
    """([(kernel symbol, vgpr_count, agpr_count)], number of code objects) of the gfx950 code objects embedded in the
    library (no GPU needed: llvm-objcopy dumps .hip_fatbin, clang-offload-bundler unbundles, llvm-readelf prints the
    AMDGPU metadata note).  The section holds ONE BUNDLE PER HIP TRANSLATION UNIT, one behind the other; the bundler
    reads only the bundle a file starts with, so the section is cut at every bundle magic first."""
    import subprocess
    fat = str(tmp_path / "fat.bin")
    subprocess.check_call([os.path.join(LLVM_BIN, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, so_path])
    data = open(fat, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = []
    i = data.find(magic)
    while i >= 0:
        starts.append(i)
        i = data.find(magic, i + 1)
    out = []
    for k, beg in enumerate(starts):
        end = starts[k + 1] if k + 1 < len(starts) else len(data)
        part, co = str(tmp_path / ("bundle%d.bin" % k)), str(tmp_path / ("gfx950_%d.co" % k))
        open(part, "wb").write(data[beg:end])
        subprocess.check_call([os.path.join(LLVM_BIN, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + part,
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
        notes = subprocess.check_output([os.path.join(LLVM_BIN, "llvm-readelf"), "--notes", co]).decode()
        # one "  - .agpr_count: N" ... ".symbol: name.kd" ... ".vgpr_count: N" block per kernel
        for blk in re.split(r"\n  - (?=\.agpr_count:)", notes)[1:]:
            ag = int(re.search(r"\.agpr_count:\s+(\d+)", blk).group(1))
            vg = int(re.search(r"\.vgpr_count:\s+(\d+)", blk).group(1))
            sym = re.search(r"\.symbol:\s+'?([^\s']+)", blk).group(1)
            out.append((sym, vg, ag))
    return out, len(starts)


def test_code_object_every_kernel_allocates_an_accumulation_register(tmp_path):
    """The BUILT gfx950 code object, not the source text: every kernel that uses vector registers also
    allocates accumulation registers behind them (agpr_count > 0), so the last VGPR of its allocation
    -- the one DESIGN.md section 3a shows being overwritten -- holds nothing.  Since round 3 the
    library links no third-party device code (rocPRIM's scans and sorts are gone: prims.hip.h), so
    the rule holds for EVERY kernel symbol, with no allow-list."""
    if not os.path.exists(os.path.join(LLVM_BIN, "clang-offload-bundler")):
        pytest.skip("no LLVM binutils in this image")
    so = build.build_hip()
    ks, n_objects = gfx950_kernel_descriptors(so, tmp_path)
    # one code object per HIP translation unit (humid_hip.hip, humid_exchange.hip): every one of them is read
    n_units = 1 + sum(1 for u in build.HOST_UNITS if u.endswith(".hip"))
    assert n_objects == n_units, (n_objects, n_units)
    assert len(ks) >= 100 * n_objects, len(ks)
    bad = [(s, v, a) for s, v, a in ks if v > 0 and a == 0]
    assert not bad, bad[:10]
    foreign = [s for s, _, _ in ks if "rocprim" in s or "hipcub" in s or "thrust" in s]
    assert not foreign, foreign[:5]
