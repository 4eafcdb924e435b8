"""isolate the tear-down cost: (a) big inputs, no GPU (--dump-words); (b) GPU, tiny inputs"""
import os, re, shutil, subprocess, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from humid_amd.synth import fast_fastq   # noqa: E402
HUMID = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "humid_amd", "humid")
d = "/dev/shm/humid_edges"
shutil.rmtree(d, ignore_errors=True)
os.makedirs(d)
def run(args, files):
    t0 = time.time()
    r = subprocess.run([HUMID] + args + files, capture_output=True, text=True, env=dict(os.environ, HUMID_TIMING="1"))
    t1 = time.time()
    m0 = float(re.search(r"main\(\) entered at ([0-9.]+)", r.stderr).group(1))
    m = re.search(r"leaving main\(\) at ([0-9.]+)", r.stderr)
    m1 = float(m.group(1)) if m else None
    return t1 - t0, m0 - t0, (m1 - m0) if m1 else None, (t1 - m1) if m1 else None, r.returncode
try:
    big = [os.path.join(d, "r%d.fastq" % m) for m in (1, 2)]
    for m, f in enumerate(big):
        fast_fastq(f, 10_000_000, 11, read_len=150, umi_len=8, mate=m + 1)
    small = [os.path.join(d, "s%d.fastq" % m) for m in (1, 2)]
    for m, f in enumerate(small):
        fast_fastq(f, 10_000, 11, read_len=150, umi_len=8, mate=m + 1)
    for _ in range(2):
        print("big inputs, --dump-words (no GPU; returns through ordinary exit):", run(["--dump-words", os.path.join(d, "w.bin")], big))
        print("tiny inputs, GPU:", run(["-d", os.path.join(d, "o"), "-l", "/dev/null"], small))
        print("big inputs, GPU:", run(["-d", os.path.join(d, "o2"), "-l", "/dev/null"], big))
finally:
    shutil.rmtree(d, ignore_errors=True)
