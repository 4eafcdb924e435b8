"""ctypes front end of the CPU oracle (oracle/humid_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  humid_amd/ never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

u64p = C.POINTER(C.c_uint64)
u32p = C.POINTER(C.c_uint32)
u8p = C.POINTER(C.c_uint8)
i64p = C.POINTER(C.c_int64)
f64p = C.POINTER(C.c_double)


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "humid_oracle.c")
    hdr = os.path.join(_HERE, "humid_oracle.h")
    stale = (not os.path.exists(so)) or any(
        os.path.getmtime(p) > os.path.getmtime(so) for p in (src, hdr))
    if force or stale:
        subprocess.check_call(
            ["gcc", "-O2", "-std=c11", "-fPIC", "-shared", "-o", so, src], cwd=_HERE)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.orc_at_least_double.argtypes = [C.c_size_t, C.c_size_t]
        L.orc_at_least_double.restype = C.c_int
        L.orc_graph_create.argtypes = [C.c_size_t]
        L.orc_graph_create.restype = C.c_void_p
        L.orc_graph_destroy.argtypes = [C.c_void_p]
        L.orc_graph_set_count.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t]
        L.orc_graph_link.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t]
        L.orc_graph_preassign.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t]
        L.orc_graph_max_neighbour.argtypes = [C.c_void_p, C.c_size_t]
        L.orc_graph_max_neighbour.restype = C.c_size_t
        L.orc_graph_assign.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_int]
        L.orc_graph_find_clusters.argtypes = [C.c_void_p, C.c_int]
        L.orc_graph_find_clusters.restype = C.c_size_t
        L.orc_graph_export.argtypes = [C.c_void_p, u32p, u64p, u64p, i64p, C.c_size_t]
        L.orc_make_string_size.argtypes = [C.c_char_p, C.c_size_t, C.c_char, C.c_char_p]
        L.orc_make_string_size.restype = C.c_size_t
        L.orc_extract_last_field.argtypes = [C.c_char_p, C.c_char, C.c_char_p]
        L.orc_extract_last_field.restype = C.c_size_t
        L.orc_valid_umi.argtypes = [C.c_char_p]
        L.orc_valid_umi.restype = C.c_int
        L.orc_extract_umi.argtypes = [C.c_char_p, C.c_char_p]
        L.orc_extract_umi.restype = C.c_size_t
        L.orc_nt_from_file.argtypes = [C.c_size_t, C.c_size_t, C.POINTER(C.c_size_t)]
        L.orc_get_nucleotides.argtypes = [C.c_char_p, C.POINTER(C.c_char_p), C.c_size_t,
                                          C.POINTER(C.c_size_t), C.c_size_t, C.c_char_p]
        L.orc_get_nucleotides.restype = C.c_size_t
        L.orc_make_word.argtypes = [C.c_char_p, C.c_size_t, u8p]
        L.orc_make_word.restype = C.c_int
        L.orc_pack_word.argtypes = [u8p, C.c_size_t]
        L.orc_pack_word.restype = C.c_uint64
        L.orc_pre_compute.argtypes = [C.c_size_t, C.c_size_t, C.c_size_t,
                                      C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        L.orc_create.argtypes = [C.c_uint32]
        L.orc_create.restype = C.c_void_p
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_read_data.argtypes = [C.c_void_p, u64p, u8p, C.c_uint64]
        L.orc_find_hamming_neighbours.argtypes = [C.c_void_p, C.c_uint32]
        L.orc_find_edit_neighbours.argtypes = [C.c_void_p, C.c_uint32]
        L.orc_find_edit_neighbours.restype = C.c_uint64
        L.orc_find_hamming_neighbours.restype = C.c_uint64
        L.orc_find_clusters.argtypes = [C.c_void_p, C.c_int]
        L.orc_find_clusters.restype = C.c_uint64
        L.orc_map_reads.argtypes = [C.c_void_p, u64p, u8p, C.c_uint64, u32p, u8p]
        for f in ("orc_total", "orc_usable", "orc_unique", "orc_n_clusters", "orc_n_edges"):
            getattr(L, f).argtypes = [C.c_void_p]
            getattr(L, f).restype = C.c_uint64
        L.orc_export_leaves.argtypes = [C.c_void_p, u64p, u64p, u32p, u32p, u8p]
        L.orc_export_adjacency.argtypes = [C.c_void_p, u64p, u32p]
        L.orc_export_clusters.argtypes = [C.c_void_p, u64p, u64p, u32p]
        L.orc_dedup_run.argtypes = [u64p, u8p, C.c_uint64, C.c_uint32, C.c_uint32,
                                    C.c_uint32, u32p, u8p, u64p, f64p]
        L.orc_dedup_run.restype = C.c_int
        _LIB = L
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(t)


# --------------------------------------------------------------------------- #
# word extraction (src/fastq.cc)
# --------------------------------------------------------------------------- #
def extract_umi(header: str) -> str:
    buf = C.create_string_buffer(len(header) + 2)
    lib().orc_extract_umi(header.encode(), buf)
    return buf.value.decode()


def extract_last_field(s: str, sep: str) -> str:
    buf = C.create_string_buffer(len(s) + 2)
    lib().orc_extract_last_field(s.encode(), sep.encode(), buf)
    return buf.value.decode()


def valid_umi(s: str) -> bool:
    return bool(lib().orc_valid_umi(s.encode()))


def make_string_size(s: str, size: int, pad: str) -> str:
    buf = C.create_string_buffer(size + 2)
    lib().orc_make_string_size(s.encode(), size, pad.encode(), buf)
    return buf.value.decode()


def nt_from_file(files: int, length: int):
    out = (C.c_size_t * files)()
    lib().orc_nt_from_file(files, length, out)
    return list(out)


def get_nucleotides(first_header: str, seqs, nt_to_take, header_umi_size: int) -> str:
    n = len(seqs)
    arr = (C.c_char_p * n)(*[s.encode() for s in seqs])
    take = (C.c_size_t * n)(*nt_to_take)
    buf = C.create_string_buffer(header_umi_size + sum(nt_to_take) + 2)
    lib().orc_get_nucleotides(first_header.encode(), arr, n, take, header_umi_size, buf)
    return buf.value.decode()


def make_word(nucleotides: str):
    n = len(nucleotides)
    data = np.zeros(max(n, 1), dtype=np.uint8)
    filt = lib().orc_make_word(nucleotides.encode(), n, _p(data, u8p))
    return data[:n].tolist(), bool(filt)


def pack_word(data) -> int:
    a = np.asarray(data, dtype=np.uint8)
    return int(lib().orc_pack_word(_p(a, u8p), len(a)))


def pre_compute(first_header_umi: int, n_files: int, word_length: int):
    h = C.c_size_t()
    take = (C.c_size_t * n_files)()
    lib().orc_pre_compute(first_header_umi, n_files, word_length, C.byref(h), take)
    return h.value, list(take)


# --------------------------------------------------------------------------- #
# hand-built graphs (tests/test_cluster.cc style)
# --------------------------------------------------------------------------- #
class Graph:
    def __init__(self, counts):
        self.n = len(counts)
        self.h = lib().orc_graph_create(self.n)
        for i, c in enumerate(counts):
            lib().orc_graph_set_count(self.h, i, int(c))

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_graph_destroy(self.h)
            self.h = None

    def link(self, a, b):
        lib().orc_graph_link(self.h, a, b)

    def preassign(self, leaf, cluster_id):
        lib().orc_graph_preassign(self.h, leaf, cluster_id)

    def max_neighbour(self, leaf):
        return int(lib().orc_graph_max_neighbour(self.h, leaf))

    def assign(self, leaf, cluster_id, maximum=False):
        lib().orc_graph_assign(self.h, leaf, cluster_id, int(maximum))

    def find_edit_neighbours(self, distance):
        return int(lib().orc_find_edit_neighbours(self.h, distance))

    def find_clusters(self, maximum=False):
        return int(lib().orc_graph_find_clusters(self.h, int(maximum)))

    def export(self, n_clusters):
        lc = np.zeros(self.n, dtype=np.uint32)
        size = np.zeros(max(n_clusters, 1), dtype=np.uint64)
        mc = np.zeros(max(n_clusters, 1), dtype=np.uint64)
        ml = np.zeros(max(n_clusters, 1), dtype=np.int64)
        lib().orc_graph_export(self.h, _p(lc, u32p), _p(size, u64p), _p(mc, u64p),
                               _p(ml, i64p), n_clusters)
        return lc, size[:n_clusters], mc[:n_clusters], ml[:n_clusters]


# --------------------------------------------------------------------------- #
# the pipeline (trie + src/humid.cc loops)
# --------------------------------------------------------------------------- #
class Pipeline:
    """read_data -> find_hamming_neighbours -> find_clusters -> map_reads."""

    def __init__(self, word_nt):
        self.h = lib().orc_create(word_nt)
        if not self.h:
            raise ValueError("word_nt must be 1..64")
        self.wpr = 2 if word_nt > 32 else 1   # wide words: (N, 2) uint64 arrays, [hi, lo]

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_destroy(self.h)
            self.h = None

    def read_data(self, words, filtered):
        self._w = np.ascontiguousarray(words, dtype=np.uint64).reshape(-1, self.wpr)
        self._f = np.ascontiguousarray(filtered, dtype=np.uint8)
        assert len(self._f) == len(self._w)
        lib().orc_read_data(self.h, _p(self._w, u64p), _p(self._f, u8p), len(self._w))

    def find_hamming_neighbours(self, distance):
        return int(lib().orc_find_hamming_neighbours(self.h, distance))

    def find_edit_neighbours(self, distance):
        return int(lib().orc_find_edit_neighbours(self.h, distance))

    def find_clusters(self, maximum=False):
        return int(lib().orc_find_clusters(self.h, int(maximum)))

    def map_reads(self):
        n = len(self._w)
        cid = np.zeros(n, dtype=np.uint32)
        keep = np.zeros(n, dtype=np.uint8)
        lib().orc_map_reads(self.h, _p(self._w, u64p), _p(self._f, u8p), n,
                            _p(cid, u32p), _p(keep, u8p))
        return cid, keep

    @property
    def unique(self):
        return int(lib().orc_unique(self.h))

    @property
    def n_clusters(self):
        return int(lib().orc_n_clusters(self.h))

    @property
    def n_edges(self):
        return int(lib().orc_n_edges(self.h))

    def summary(self):
        L = lib()
        return dict(total=int(L.orc_total(self.h)), usable=int(L.orc_usable(self.h)),
                    unique=int(L.orc_unique(self.h)), clusters=int(L.orc_n_clusters(self.h)),
                    edges=int(L.orc_n_edges(self.h)))

    def leaves(self):
        u = self.unique
        word = np.zeros(u if self.wpr == 1 else (u, 2), dtype=np.uint64)
        count = np.zeros(u, dtype=np.uint64)
        deg = np.zeros(u, dtype=np.uint32)
        cid = np.zeros(u, dtype=np.uint32)
        ismax = np.zeros(u, dtype=np.uint8)
        lib().orc_export_leaves(self.h, _p(word, u64p), _p(count, u64p), _p(deg, u32p),
                                _p(cid, u32p), _p(ismax, u8p))
        return dict(word=word, count=count, degree=deg, cluster_id=cid, is_max_leaf=ismax)

    def adjacency(self):
        u = self.unique
        off = np.zeros(u + 1, dtype=np.uint64)
        idx = np.zeros(max(2 * self.n_edges, 1), dtype=np.uint32)
        lib().orc_export_adjacency(self.h, _p(off, u64p), _p(idx, u32p))
        return off, idx[:2 * self.n_edges]

    def clusters(self):
        c = self.n_clusters
        size = np.zeros(max(c, 1), dtype=np.uint64)
        mc = np.zeros(max(c, 1), dtype=np.uint64)
        ml = np.zeros(max(c, 1), dtype=np.uint32)
        lib().orc_export_clusters(self.h, _p(size, u64p), _p(mc, u64p), _p(ml, u32p))
        return dict(size=size[:c], max_count=mc[:c], max_leaf=ml[:c])


def dedup_run(words, filtered, word_nt, distance=1, method=0, edit=False):
    """One-call oracle run.  Returns (cluster_id, keep, summary dict, phase seconds).
    method 0 directional / 1 maximum; edit: Levenshtein instead of Hamming neighbours (-e)."""
    method = (int(method) & 1) | (2 if edit else 0)
    w = np.ascontiguousarray(words, dtype=np.uint64).reshape(-1, 2 if word_nt > 32 else 1)
    f = np.ascontiguousarray(filtered, dtype=np.uint8)
    n = len(w)
    assert len(f) == n
    cid = np.zeros(n, dtype=np.uint32)
    keep = np.zeros(n, dtype=np.uint8)
    s = np.zeros(5, dtype=np.uint64)
    ph = np.zeros(4, dtype=np.float64)
    rc = lib().orc_dedup_run(_p(w, u64p), _p(f, u8p), n, word_nt, distance, method,
                             _p(cid, u32p), _p(keep, u8p), _p(s, u64p), _p(ph, f64p))
    if rc != 0:
        raise RuntimeError("orc_dedup_run failed: %d" % rc)
    summary = dict(total=int(s[0]), usable=int(s[1]), unique=int(s[2]), clusters=int(s[3]), edges=int(s[4]))
    return cid, keep, summary, ph.tolist()


def histograms(p: "Pipeline"):
    """counts.dat / neigh.dat / clusters.dat / stats.dat contents
    (src/humid.cc:301-357, src/cluster.cc:89-95) as sorted (key, value) lists."""
    lv = p.leaves()
    cl = p.clusters()

    def hist(a):
        k, v = np.unique(np.asarray(a, dtype=np.uint64), return_counts=True)
        return [(int(x), int(y)) for x, y in zip(k, v)]

    s = p.summary()
    return dict(counts=hist(lv["count"]), neigh=hist(lv["degree"]), clusters=hist(cl["size"]),
                stats=dict(total=s["total"], usable=s["usable"], unique=s["unique"],
                           clusters=s["clusters"]))
