"""In-situ rerun of the plan-passing forms that lost edges in round 1.

tools/exp_plan/make_exp.py builds experiment variants of the library (libexp_<name>.so) in which
k_combo_keys / k_pairs read the pigeonhole plan from a struct passed by value and indexed dynamically
(form 1) or from a freshly uploaded device copy through wave-uniform loads (form 2).  This script
runs the whole pipeline `reps` times on the 10 M-read metric workload with one variant and compares
unique / edges / clusters and every leaf's degree with the known-good values (oracle: 218 883 edges).

    python tools/exp_plan/run.py <variant | product> [reps]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
name = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
from humid_amd import _lib  # noqa: E402

if name != "product":
    _lib.SO_PATH = os.path.join(ROOT, "tools", "exp_plan", "libexp_%s.so" % name)
import humid_amd  # noqa: E402
from humid_amd.synth import synth_words  # noqa: E402

cache = "/tmp/exp_plan_words.npz"
if os.path.exists(cache):
    z = np.load(cache)
    words, filt = z["w"], z["f"]
else:
    words, filt = synth_words(10_000_000, 1002, 24)
    np.savez(cache, w=words, f=filt)
dd = humid_amd.Dedup()
edges = []
for rep in range(reps):
    cid, keep, s = dd.run(words, filt)
    edges.append(int(s["edges"]))
tag = " ".join("%s=%s" % (k, os.environ[k]) for k in ("HIP_FORCE_DEV_KERNARG", "EXP_SYNC_UPLOAD") if k in os.environ)
lost = [218883 - e for e in edges]
print("%-12s %-28s unique %d  edges lost per run: %s  -> %s" %
      (name, tag, s["unique"], lost, "EXACT" if not any(lost) else "LOSES EDGES"), flush=True)
