"""Builds libhumid_hip.so (hipcc, gfx950) in-tree.  No torch involvement: the library is a
plain C-ABI shared object (include/humid_hip.h)."""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "csrc", "humid_hip.hip")
HDR = os.path.join(ROOT, "include", "humid_hip.h")
SO = os.path.join(HERE, "libhumid_hip.so")

HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared"]


def _sources():
    out = [HDR]
    for d, _, fs in os.walk(os.path.join(HERE, "csrc")):
        if os.sep + "host" in d:
            continue
        out += [os.path.join(d, f) for f in fs if f.endswith((".hip", ".h", ".hpp", ".cpp"))]
    return out


def is_stale() -> bool:
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    return any(os.path.getmtime(p) > t for p in _sources())


# translation units of the library: the HIP ones -- humid_hip.hip (context, single-GPU entry points, accessors) and
# humid_exchange.hip (exchange pass, multi-GPU stages), both over pipeline.hip.h -- and the host-only ones (plain
# C++ that hipcc compiles as such)
HOST_UNITS = [os.path.join(HERE, "csrc", "humid_exchange.hip"), os.path.join(HERE, "csrc", "shm.cpp")]


def build_hip(force: bool = False, verbose: bool = False) -> str:
    if not force and not is_stale():
        return SO
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc] + HIPCC_FLAGS + ["-o", SO, SRC] + HOST_UNITS
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=ROOT)
    return SO


HOST_DIR = os.path.join(HERE, "csrc", "host")
HOST_EXE = os.path.join(HERE, "humid")


def build_host(force: bool = False, verbose: bool = False) -> str:
    """The `humid` command-line host (C++17, g++): FastQ streaming + the C ABI."""
    srcs = [os.path.join(HOST_DIR, f) for f in ("main.cpp", "sharded.cpp", "fastq_io.cpp", "fastq_mmap.cpp", "fast_inflate.cpp", "words.cpp")]
    deps = srcs + [os.path.join(HOST_DIR, f) for f in ("fastq_io.hpp", "fastq_mmap.hpp", "fast_inflate.hpp", "words.hpp", "sharded.hpp")] + [HDR]
    build_hip(force=False, verbose=verbose)
    if not force and os.path.exists(HOST_EXE) and \
            all(os.path.getmtime(p) <= os.path.getmtime(HOST_EXE) for p in deps + [SO]):
        return HOST_EXE
    # sharded.cpp calls the HIP runtime (memory, streams, peer copies) and uses RCCL's types; librccl
    # itself is loaded with dlopen when a -g N run wants it
    cmd = ["g++", "-O3", "-std=c++17", "-Wall", "-Wextra", "-D__HIP_PLATFORM_AMD__", "-isystem", "/opt/rocm/include",
           "-o", HOST_EXE] + srcs + \
          ["-L" + HERE, "-lhumid_hip", "-L/opt/rocm/lib", "-lamdhip64", "-ldl", "-lz", "-lpthread",
           "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,/opt/rocm/lib", "-Wl,-rpath-link,/opt/rocm/lib"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=ROOT)
    return HOST_EXE


if __name__ == "__main__":
    print(build_hip(force=True, verbose=True))
    print(build_host(force=True, verbose=True))
