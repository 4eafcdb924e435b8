"""Host-side mirror of the reference surface for the hot path, over the C ABI.

Names follow the reference (/root/reference/src/humid.cc, src/cluster.h):
  Dedup.run(words, filtered)            readData + findHammingNeighbours + findClusters +
                                        the writeFiltered/writeAnnotated lookups
  Dedup.leaves()/adjacency()/clusters() what Trie::walk() / NLeaf / Cluster expose
  ClusterGraph                          NLeaf graphs built with link() (tests/test_cluster.cc)
  at_least_double                       src/cluster.cc:31-33
Everything executes in libhumid_hip.so on the GPU; a missing library or GPU raises.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib

DIRECTIONAL = 0
MAXIMUM = 1


class HumidError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("humid_hip error %d: %s" % (code, msg))
        self.code = code


def _vp(a):
    if a is None:
        return None
    return C.c_void_p(a.ctypes.data)


class Context:
    """Owns a humid_ctx (device workspace + stream)."""

    def __init__(self, device: int = -1, stream: int | None = None):
        self._lib = _lib.load()
        h = C.c_void_p()
        rc = self._lib.humid_ctx_create(C.byref(h), device, C.c_void_p(stream) if stream else None)
        if rc != 0:
            raise HumidError(rc, self._lib.humid_last_error(None).decode())
        self._h = h

    def set_option(self, key: str, value: int):
        """tuning knobs of include/humid_hip.h (humid_ctx_set_option); never change results"""
        self._check(self._lib.humid_ctx_set_option(self._h, key.encode(), int(value)))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.humid_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise HumidError(rc, self._lib.humid_last_error(self._h).decode())


class Dedup(Context):
    """The whole hot path on one GPU."""

    def run(self, words, filtered, word_nt=24, distance=1, method=DIRECTIONAL, edit=False):
        """Host numpy buffers in, (cluster_id u32[N], keep u8[N], summary dict) out.
        edit: neighbours under Levenshtein instead of Hamming distance (the reference's -e).

        word_nt <= 32: words is u64[N].  33 <= word_nt <= 64: words is u64[N, 2], [:, 0] = the
        first word_nt-32 nucleotides, [:, 1] = the last 32 (include/humid_hip.h)."""
        w = np.ascontiguousarray(words, dtype=np.uint64)
        f = np.ascontiguousarray(filtered, dtype=np.uint8)
        want = (len(f), 2) if word_nt > 32 else (len(f),)
        if f.ndim != 1 or w.shape != want:
            raise ValueError("words must have shape %r for word_nt=%d (filtered: %r)" % (want, word_nt, f.shape))
        n = len(f)
        self._wide = word_nt > 32
        self.set_option("edit_distance", int(bool(edit)))
        cid = np.zeros(n, dtype=np.uint32)
        keep = np.zeros(n, dtype=np.uint8)
        s = _lib.HumidSummary()
        self._check(self._lib.humid_dedup_run(self._h, _vp(w), _vp(f), n, word_nt, distance, method,
                                              _vp(cid), _vp(keep), C.byref(s)))
        self.summary = s.asdict()
        return cid, keep, self.summary

    def run_bases(self, bases, word_nt=24, distance=1, method=DIRECTIONAL, edit=False):
        """bases: uint8[N, word_nt] -- the symbols of every record as the ASCII of the FastQ (what
        getNucleotides, src/fastq.cc:116-144, assembles); the words are packed on the device.
        Returns (cluster_id, keep, summary) like run()."""
        b = np.ascontiguousarray(bases, dtype=np.uint8)
        if b.ndim != 2 or b.shape[1] != word_nt:
            raise ValueError("bases must have shape (N, %d)" % word_nt)
        n = b.shape[0]
        self._wide = word_nt > 32
        self.set_option("edit_distance", int(bool(edit)))
        cid = np.zeros(n, dtype=np.uint32)
        keep = np.zeros(n, dtype=np.uint8)
        s = _lib.HumidSummary()
        self._check(self._lib.humid_dedup_run_bases(self._h, _vp(b), n, word_nt, distance, method, _vp(cid),
                                                    _vp(keep), C.byref(s)))
        self.summary = s.asdict()
        return cid, keep, self.summary

    def packed_words(self):
        """words and filtered flags the device packed in the last run_bases()"""
        n = int(self.summary["total"])
        w = np.zeros((n, 2) if getattr(self, "_wide", False) else n, np.uint64)
        f = np.zeros(n, np.uint8)
        self._check(self._lib.humid_get_packed_words(self._h, _vp(w), _vp(f)))
        return w, f

    def run_device(self, d_words, d_filtered, d_cluster_id, d_keep, n_reads, word_nt=24,
                   distance=1, method=DIRECTIONAL):
        """Device pointers (ints, e.g. tensor.data_ptr()); results stay in HBM."""
        s = _lib.HumidSummary()
        self._check(self._lib.humid_dedup_run_device(
            self._h, C.c_void_p(d_words), C.c_void_p(d_filtered), n_reads, word_nt, distance,
            method, C.c_void_p(d_cluster_id), C.c_void_p(d_keep), C.byref(s)))
        self.summary = s.asdict()
        self._wide = word_nt > 32
        return self.summary

    def leaves(self):
        u = int(self.summary["unique"])
        wshape = (u, 2) if getattr(self, "_wide", False) else u
        out = dict(word=np.zeros(wshape, np.uint64), count=np.zeros(u, np.uint32),
                   first_read=np.zeros(u, np.uint32), degree=np.zeros(u, np.uint32),
                   cluster_id=np.zeros(u, np.uint32), is_max_leaf=np.zeros(u, np.uint8))
        self._check(self._lib.humid_get_leaves(self._h, _vp(out["word"]), _vp(out["count"]),
                                               _vp(out["first_read"]), _vp(out["degree"]),
                                               _vp(out["cluster_id"]), _vp(out["is_max_leaf"])))
        return out

    def adjacency(self):
        u = int(self.summary["unique"])
        e2 = 2 * int(self.summary["edges"])
        off = np.zeros(u + 1, np.uint32)
        idx = np.zeros(max(e2, 1), np.uint32)
        self._check(self._lib.humid_get_adjacency(self._h, _vp(off), _vp(idx)))
        return off, idx[:e2]

    def clusters(self):
        c = int(self.summary["clusters"])
        size = np.zeros(max(c, 1), np.uint64)
        mc = np.zeros(max(c, 1), np.uint32)
        ml = np.zeros(max(c, 1), np.uint32)
        self._check(self._lib.humid_get_clusters(self._h, _vp(size), _vp(mc), _vp(ml)))
        return dict(size=size[:c], max_count=mc[:c], max_leaf=ml[:c])

    def histogram(self, which: int):
        """which: 0 counts.dat, 1 neigh.dat, 2 clusters.dat -> sorted [(key, value)]"""
        n = C.c_uint64()
        self._check(self._lib.humid_get_histogram(self._h, which, None, None, 0, C.byref(n)))
        k = np.zeros(max(n.value, 1), np.uint64)
        v = np.zeros(max(n.value, 1), np.uint64)
        self._check(self._lib.humid_get_histogram(self._h, which, _vp(k), _vp(v), n.value, C.byref(n)))
        return [(int(a), int(b)) for a, b in zip(k[:n.value], v[:n.value])]

    def histograms(self):
        s = self.summary
        return dict(counts=self.histogram(0), neigh=self.histogram(1), clusters=self.histogram(2),
                    stats=dict(total=int(s["total"]), usable=int(s["usable"]),
                               unique=int(s["unique"]), clusters=int(s["clusters"])))


class ClusterGraph(Context):
    """NLeaf graphs built by hand, as tests/test_cluster.cc:11-14 does with link()."""

    def __init__(self, counts, device: int = -1):
        super().__init__(device)
        self.counts = [int(c) for c in counts]
        self.nbrs = [[] for _ in self.counts]

    def link(self, a, b):
        self.nbrs[a].append(b)
        self.nbrs[b].append(a)

    def find_clusters(self, maximum=False):
        """findClusters over the leaves in index order.  Returns dict(leaf_cluster, size,
        max_count, max_leaf, n_clusters)."""
        u = len(self.counts)
        cnt = np.asarray(self.counts, dtype=np.uint32)
        off = np.zeros(u + 1, dtype=np.uint32)
        for i, l in enumerate(self.nbrs):
            off[i + 1] = off[i] + len(l)
        idx = np.asarray([x for l in self.nbrs for x in l] or [0], dtype=np.uint32)
        lc = np.zeros(max(u, 1), np.uint32)
        size = np.zeros(max(u, 1), np.uint64)
        mc = np.zeros(max(u, 1), np.uint32)
        ml = np.zeros(max(u, 1), np.uint32)
        nc = C.c_uint32()
        self._check(self._lib.humid_cluster_graph(self._h, _vp(cnt), _vp(off), _vp(idx), u,
                                                  MAXIMUM if maximum else DIRECTIONAL, _vp(lc),
                                                  _vp(size), _vp(mc), _vp(ml), C.byref(nc)))
        c = nc.value
        return dict(leaf_cluster=lc[:u], size=size[:c], max_count=mc[:c], max_leaf=ml[:c],
                    n_clusters=c)


def at_least_double(a: int, b: int, ctx: Context | None = None) -> bool:
    own = ctx is None
    ctx = ctx or Context()
    r = C.c_int()
    ctx._check(ctx._lib.humid_at_least_double(ctx._h, a, b, C.byref(r)))
    if own:
        ctx.close()
    return bool(r.value)
