import os, sys, subprocess, time, shutil, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from humid_amd.synth import fast_fastq
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
d = tempfile.mkdtemp(prefix="humid_e2e_", dir="/dev/shm")
r1, r2 = os.path.join(d, "R1.fastq"), os.path.join(d, "R2.fastq")
fast_fastq(r1, n, 1002, mate=0); fast_fastq(r2, n, 1002, mate=1)
exe = os.path.join(ROOT, "humid_amd", "humid")
for label, env in (("default", {}), ("slow exit", {"HUMID_SLOW_EXIT": "1"}), ("default again", {})):
    best = None
    for rep in range(3):
        out = os.path.join(d, "out")
        e = dict(os.environ, HUMID_TIMING="1", **env)
        t0 = time.perf_counter(); w0 = time.time()
        p = subprocess.run([exe, "-d", out, "-l", os.path.join(d, "log.txt"), r1, r2], env=e, stderr=subprocess.PIPE)
        dt = time.perf_counter() - t0; w1 = time.time()
        if best is None or dt < best[0]:
            best = (dt, p.stderr.decode() + "[humid] caller: started the process at %.6f, saw it gone at %.6f\n" % (w0, w1))
        shutil.rmtree(out, ignore_errors=True)
    print("=== %s: %.3f s" % (label, best[0]))
    for l in best[1].splitlines():
        if l.startswith("[humid]"):
            print("   ", l[8:])
shutil.rmtree(d, ignore_errors=True)
