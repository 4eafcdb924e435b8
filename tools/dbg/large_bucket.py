import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import humid_amd
from oracle import pyoracle as orc
rng = np.random.default_rng(11)
n_reads = 60000
umi = rng.integers(0, 4 ** 6, size=n_reads, dtype=np.uint64)
words = (umi << np.uint64(36)) | np.uint64(0x123456789)
filt = np.zeros(n_reads, np.uint8)
p = orc.Pipeline(24); p.read_data(words, filt); p.find_hamming_neighbours(1); p.find_clusters(False)
print("oracle", p.summary())
for compact in (1, 0):
    for walk in (1024, 0):
        dd = humid_amd.Dedup()
        dd.set_option("compact_graph", compact)
        dd.set_option("bucket_walk", walk)
        cid, keep, s = dd.run(words, filt, word_nt=24, distance=1, method=0)
        print("compact", compact, "walk", walk, {k: s[k] for k in ("unique", "clusters", "edges", "nonsingle")})
        lv = dd.leaves(); olv = p.leaves()
        print("  degree equal", np.array_equal(lv["degree"], olv["degree"]), "cid equal", np.array_equal(lv["cluster_id"], olv["cluster_id"]))
        dd.close()
dd = humid_amd.Dedup()
cid, keep, s = dd.run(words, filt, word_nt=24, distance=1, method=0)
off, idx = dd.adjacency(); ooff, oidx = p.adjacency()
bad = 0
for u in range(4096):
    a = list(idx[off[u]:off[u+1]]); b = list(oidx[ooff[u]:ooff[u+1]])
    if a != b:
        extra = sorted(set(a) - set(b)); dup = [x for x in set(a) if a.count(x) > 1]; miss = sorted(set(b) - set(a))
        if bad < 12: print(u, "len", len(a), len(b), "extra", extra[:6], "dup", dup[:6], "missing", miss[:6])
        bad += 1
print("bad rows", bad)
