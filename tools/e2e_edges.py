"""where the wall time of one `humid` run goes OUTSIDE main(): spawn -> main() and main() end -> reaped.
python tools/e2e_edges.py [--reads 10000000] [--se]"""
import argparse, os, re, shutil, subprocess, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from humid_amd.synth import fast_fastq   # noqa: E402
HUMID = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "humid_amd", "humid")
ap = argparse.ArgumentParser()
ap.add_argument("--reads", type=int, default=10_000_000)
ap.add_argument("--se", action="store_true")
a = ap.parse_args()
d = "/dev/shm/humid_edges"
shutil.rmtree(d, ignore_errors=True)
os.makedirs(d)
files = [os.path.join(d, "r%d.fastq" % m) for m in ((1,) if a.se else (1, 2))]
for m, f in enumerate(files):
    fast_fastq(f, a.reads, 11, read_len=150, umi_len=8, mate=m + 1)
try:
    for env_extra in ({}, {"HUMID_SLOW_EXIT": "1"}, {}):
        out = os.path.join(d, "out")
        shutil.rmtree(out, ignore_errors=True)
        t0 = time.time()
        r = subprocess.run([HUMID, "-d", out, "-l", "/dev/null"] + files, capture_output=True, text=True,
                           env=dict(os.environ, HUMID_TIMING="1", **env_extra))
        t1 = time.time()
        m0 = float(re.search(r"main\(\) entered at ([0-9.]+)", r.stderr).group(1))
        m1 = float(re.search(r"leaving main\(\) at ([0-9.]+)", r.stderr).group(1))
        print("%s wall %.3f s: spawn->main %.3f, main %.3f, main end->reaped %.3f" % (env_extra, t1 - t0, m0 - t0, m1 - m0, t1 - m1))
        for l in r.stderr.split("\n"):
            if l.startswith("[humid]") and "epoch" not in l:
                print("    " + l)
finally:
    shutil.rmtree(d, ignore_errors=True)
