import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, humid_amd
from humid_amd.synth import synth_words
os.environ["HUMID_TRACE_COUNT"] = "1"
w, f = synth_words(10_000_000, 1002, 24)
dd = humid_amd.Dedup()
for i in range(3):
    cid, keep, s = dd.run(w, f, word_nt=24, distance=1)
    print({k: s[k] for k in ("unique", "clusters", "edges", "count_mode_used", "ms_total", "ms_count")})
