#!/usr/bin/env python3
"""End-to-end wall time of the `humid` CLI (T_e2e) on a synthetic SE FastQ with the
UMI in the header: parse + H2D + device path + D2H + write.  Fast FastQ writer (numpy), not the
test generator."""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def write_fastq(path, n_reads, seed, read_len=150, umi_len=8):
    rng = np.random.Generator(np.random.PCG64(seed))
    n_mol = n_reads // 4 + 1
    mol = rng.integers(0, n_mol, size=n_reads)
    alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
    umi = alpha[rng.integers(0, 4, size=(n_mol, umi_len))][mol]
    seq = alpha[rng.integers(0, 4, size=(n_mol, read_len))][mol]
    err = rng.random(seq.shape) < 1e-3
    seq[err] = alpha[rng.integers(0, 4, size=int(err.sum()))]
    idx = np.char.zfill(np.arange(n_reads).astype("U9"), 9).astype("S9").view(np.uint8).reshape(n_reads, 9)
    rec_len = 2 + 9 + 1 + umi_len + 1 + read_len + 3 + read_len + 1
    buf = np.empty((n_reads, rec_len), dtype=np.uint8)
    p = 0
    buf[:, p:p + 2] = np.frombuffer(b"@r", dtype=np.uint8); p += 2
    buf[:, p:p + 9] = idx; p += 9
    buf[:, p] = ord("_"); p += 1
    buf[:, p:p + umi_len] = umi; p += umi_len
    buf[:, p] = ord("\n"); p += 1
    buf[:, p:p + read_len] = seq; p += read_len
    buf[:, p:p + 3] = np.frombuffer(b"\n+\n", dtype=np.uint8); p += 3
    buf[:, p:p + read_len] = ord("I"); p += read_len
    buf[:, p] = ord("\n")
    buf.tofile(path)
    return os.path.getsize(path)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
    tmp = sys.argv[2] if len(sys.argv) > 2 else "/tmp/humid_cli_bench"
    os.makedirs(tmp, exist_ok=True)
    fq = os.path.join(tmp, "reads.fastq")
    size = write_fastq(fq, n, 7)
    exe = os.path.join(ROOT, "humid_amd", "humid")
    inputs = [("plain", fq)]
    if "--gz" in sys.argv:
        gz = fq + ".gz"
        t0 = time.perf_counter()
        subprocess.check_call("gzip -4 -c %s > %s" % (fq, gz), shell=True)
        print("gzip -4 of the input took %.1f s (%.0f MB)" % (time.perf_counter() - t0, os.path.getsize(gz) / 1e6), flush=True)
        inputs.append(("gzip in, gzip out", gz))
    for kind, path in inputs:
        for label, extra in (("dedup only", []), ("dedup + annotate + stats", ["-a", "-s"])):
            for env_label, env in (("mapped/inflate-once host path", {}), ("streaming host path", {"HUMID_HOST_SLOW": "1"})):
                if kind == "plain" and env:
                    continue
                e = dict(os.environ)
                e.update(env)
                e["HUMID_TIMING"] = "1"
                t0 = time.perf_counter()
                subprocess.check_call([exe, "-d", os.path.join(tmp, "out"), "-l", os.path.join(tmp, "log.txt")] + extra + [path], env=e)
                dt = time.perf_counter() - t0
                print("%s, %s, %s: %d reads (%.0f MB FastQ): %.2f s wall = %.2f M reads/s, %.0f MB/s of FastQ"
                      % (kind, env_label, label, n, size / 1e6, dt, n / dt / 1e6, size / dt / 1e6), flush=True)
    print(open(os.path.join(tmp, "log.txt")).read())


if __name__ == "__main__":
    main()
