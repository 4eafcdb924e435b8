#!/usr/bin/env python3
"""Kernel descriptors (register allocation) of named kernels, read from BUILT gfx950 code objects.

  python tools/kernel_descriptors.py k_combo_keys > profiles/r02_edge_loss/k_combo_keys_descriptors.txt

Prints, for every kernel of humid_amd/libhumid_hip.so whose symbol contains one of the given names, the
note fields the last-VGPR guard is about: .vgpr_count, .agpr_count,
.sgpr_count, and from llvm-readelf's descriptor dump next_free_vgpr / accum_offset.  Then it compiles
the same kernels once more WITHOUT the guard (a scratch translation unit that redefines
HUMID_GUARD_LAST_VGPR() as nothing; outputs under /tmp only) and prints the same fields: the "before".
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
sys.path.insert(0, ROOT)


def code_objects(so, tmp):
    """the gfx950 code objects of a library: .hip_fatbin holds one bundle per HIP translation unit, one behind the
    other, and clang-offload-bundler reads only the bundle a file starts with"""
    fat = os.path.join(tmp, "fat.bin")
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, so])
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
        part, co = os.path.join(tmp, "bundle%d.bin" % k), os.path.join(tmp, "gfx950_%d.co" % k)
        open(part, "wb").write(data[beg:end])
        subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + part,
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
        out.append(co)
    return out


def code_object(so, tmp):
    """(kept for one-unit libraries: the first code object)"""
    return code_objects(so, tmp)[0]


def descriptors(co, names):
    notes = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", co], text=True)
    out = []
    for blk in re.split(r"\n  - (?=\.agpr_count:)", notes)[1:]:
        sym = re.search(r"\.symbol:\s+'?([^\s']+)", blk).group(1)
        if not any(n in sym for n in names):
            continue
        f = {k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1)) for k in ("vgpr_count", "agpr_count", "sgpr_count")}
        out.append((sym, f))
    return out


def show(title, so, names, tmp):
    print("## " + title)
    demangle = "c++filt"
    found = {}
    for co in code_objects(so, tmp):                     # (a kernel of pipeline.hip.h is in both units' objects: listed once)
        for sym, f in descriptors(co, names):
            found[sym] = f
    for sym, f in sorted(found.items()):
        nice = subprocess.check_output([demangle, sym.replace(".kd", "")], text=True).strip()
        # on gfx90a+ (unified register file) .vgpr_count is the TOTAL: architectural VGPRs, rounded up to the
        # accumulation offset (a multiple of 4), plus the accumulation registers behind them
        print("%s\n    .vgpr_count %d (total)  .agpr_count %d  .sgpr_count %d   -> %s" % (
            nice[:110], f["vgpr_count"], f["agpr_count"], f["sgpr_count"],
            "accumulation registers start at %d: the last architectural VGPR (v%d) is not the last register of the allocation"
            % (f["vgpr_count"] - f["agpr_count"], f["vgpr_count"] - f["agpr_count"] - 1)
            if f["agpr_count"] else "v%d is the LAST register of the allocation" % (f["vgpr_count"] - 1)))
    print()


def main():
    names = sys.argv[1:] or ["k_combo_keys"]
    from humid_amd import build
    so = build.build_hip()
    with tempfile.TemporaryDirectory() as tmp:
        show("product library (%s), every kernel starts with HUMID_GUARD_LAST_VGPR()" % os.path.relpath(so, ROOT), so, names, tmp)
        # every HIP unit of the library once more with the guard defined away (scratch units that include the real ones)
        units = [build.SRC] + [u for u in build.HOST_UNITS if u.endswith(".hip")]
        srcs = []
        for k, u in enumerate(units):
            src = os.path.join(tmp, "noguard%d.hip" % k)
            with open(src, "w") as fh:
                fh.write('#include "common.hip.h"\n#undef HUMID_GUARD_LAST_VGPR\n#define HUMID_GUARD_LAST_VGPR()\n'
                         '#include "%s"\n' % os.path.basename(u))
            srcs.append(src)
        so2 = os.path.join(tmp, "noguard.so")
        subprocess.check_call(["hipcc"] + build.HIPCC_FLAGS + ["-I", os.path.join(ROOT, "humid_amd", "csrc"), "-o", so2] + srcs +
                              [u for u in build.HOST_UNITS if not u.endswith(".hip")], cwd=ROOT)
        show("the same sources with the guard defined away (scratch build, not shipped)", so2, names, tmp)


if __name__ == "__main__":
    main()
