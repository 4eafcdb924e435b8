"""Builds and loads tests/csrc/prims_harness.hip (test infrastructure: the scan and radix sort of
humid_amd/csrc/prims.hip.h behind a C interface).  hipcc cross-compiles for gfx950 without a GPU; the built
library lies under tests/_build/ (git-ignored) and travels to the GPU box with the tree."""
import ctypes
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "csrc", "prims_harness.hip")
SO = os.path.join(HERE, "_build", "libprims_harness.so")
DEPS = [SRC] + [os.path.join(ROOT, "humid_amd", "csrc", f) for f in ("prims.hip.h", "common.hip.h")]


def build(force: bool = False) -> str:
    if not force and os.path.exists(SO) and all(os.path.getmtime(p) <= os.path.getmtime(SO) for p in DEPS):
        return SO
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                           "-I", os.path.join(ROOT, "humid_amd", "csrc"), "-o", SO, SRC], cwd=ROOT)
    return SO


def load():
    import torch  # noqa: F401  (the HIP runtime torch loads comes first, as in humid_amd/_lib.py)
    lib = ctypes.CDLL(build())
    vp, u64, u32, i = ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int
    lib.ph_exscan_u32.argtypes = [vp, vp, u64, i]
    lib.ph_exscan_u64.argtypes = [vp, vp, u64, i]
    lib.ph_sort_u32.argtypes = [vp, vp, vp, vp, u64, u32, u32, i, i]
    lib.ph_sort_u64.argtypes = [vp, vp, vp, vp, u64, u32, u32, i, i]
    lib.ph_set_epoch.argtypes = [u32]
    lib.ph_epoch.restype = u32
    return lib


if __name__ == "__main__":
    print(build(force=True))
