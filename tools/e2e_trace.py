#!/usr/bin/env python3
"""The `humid` command line on bench.py's end-to-end shape (10 M read pairs PE150, files in /dev/shm), HUMID_TIMING=1:
prints the CLI's own phase lines of each of three runs.  usage: e2e_trace.py [n_reads] [extra env K=V ...]"""
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from humid_amd.synth import fast_fastq  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
extra = dict(kv.split("=", 1) for kv in sys.argv[2:])
d = tempfile.mkdtemp(prefix="humid_e2e_", dir="/dev/shm")
try:
    r1, r2 = os.path.join(d, "R1.fastq"), os.path.join(d, "R2.fastq")
    fast_fastq(r1, n, 1002, mate=0)
    fast_fastq(r2, n, 1002, mate=1)
    exe = os.path.join(ROOT, "humid_amd", "humid")
    for rep in range(3):
        out = os.path.join(d, "out%d" % rep)
        t0 = time.perf_counter()
        p = subprocess.run([exe, "-n", "24", "-m", "1", "-d", out, "-l", os.path.join(d, "log.txt"), r1, r2],
                           env=dict(os.environ, HUMID_TIMING="1", **extra), stderr=subprocess.PIPE)
        dt = time.perf_counter() - t0
        print("== run %d: %.4f s wall, exit %d" % (rep, dt, p.returncode))
        print(p.stderr.decode())
        shutil.rmtree(out, ignore_errors=True)
finally:
    shutil.rmtree(d, ignore_errors=True)
