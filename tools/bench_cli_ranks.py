"""`humid -g N` timing on what the box has (tools; not a test): PE150 FastQ in /dev/shm, -g 1 against
-g 2 / -g 4 (ranks share the GPUs that exist).  python tools/bench_cli_ranks.py [--reads 10000000]"""
import argparse
import os
import shutil
import subprocess
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from humid_amd.synth import fast_fastq   # noqa: E402

HUMID = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "humid_amd", "humid")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=10_000_000)
    ap.add_argument("--ranks", default="1,2,4")
    a = ap.parse_args()
    d = "/dev/shm/humid_ranks"
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(d)
    files = [os.path.join(d, "r%d.fastq" % m) for m in (1, 2)]
    for m, f in enumerate(files):
        fast_fastq(f, a.reads, 11, read_len=150, umi_len=8, mate=m + 1)
    ref = None
    try:
        for g in [int(x) for x in a.ranks.split(",")]:
            out = os.path.join(d, "out%d" % g)
            best = None
            for _ in range(2):
                shutil.rmtree(out, ignore_errors=True)
                t = time.perf_counter()
                r = subprocess.run([HUMID, "-g", str(g), "-d", out, "-l", "/dev/null"] + files, capture_output=True,
                                   text=True, env=dict(os.environ, HUMID_TIMING="1"))
                dt = time.perf_counter() - t
                assert r.returncode == 0, r.stderr
                best = dt if best is None else min(best, dt)
            sizes = {f: os.path.getsize(os.path.join(out, f)) for f in sorted(os.listdir(out))}
            if ref is None:
                ref = sizes
            lines = [l for l in r.stderr.split("\n") if "ranks," in l or "device path done" in l or "pass 1 done" in l]
            print("-g %d: %.3f s wall (%.1f M pairs/s)  same output sizes: %s" % (g, best, a.reads / best / 1e6, sizes == ref))
            for l in lines:
                print("    " + l.strip())
            shutil.rmtree(out, ignore_errors=True)
    finally:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
