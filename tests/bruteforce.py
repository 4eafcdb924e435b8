"""Second, independent CPU checker for SMALL cases (pure numpy / Python loops).

It shares no code with oracle/humid_oracle.c: unique words come from np.unique,
neighbours from an all-pairs nucleotide Hamming matrix, clustering from literal
Python recursion over src/cluster.cc:10-87.  Used to cross-check the C oracle's
trie search and explicit-stack traversals.
"""
import sys

import numpy as np

M55 = np.uint64(0x5555555555555555)


def nt_hamming(a, b):
    """nucleotide mismatches between packed words (a, b uint64 arrays, broadcast)."""
    x = np.bitwise_xor(a, b)
    y = (x | (x >> np.uint64(1))) & M55
    # popcount
    y = y.astype(np.uint64)
    cnt = np.zeros(np.broadcast(a, b).shape, dtype=np.int64)
    for s in range(0, 64, 2):
        cnt += ((y >> np.uint64(s)) & np.uint64(1)).astype(np.int64)
    return cnt


def unique_counts(words, filtered):
    """1-D words (n <= 32) or (N, 2) [hi, lo] wide words: rows sort lexicographically."""
    w = np.asarray(words, dtype=np.uint64)[np.asarray(filtered) == 0]
    if w.ndim == 2:
        if len(w) == 0:
            return w.reshape(0, 2), np.zeros(0, dtype=np.int64)
        return np.unique(w, axis=0, return_counts=True)
    return np.unique(w, return_counts=True)


def adjacency(uw, distance):
    """ascending neighbour lists (H1+H2): list of lists of ranks."""
    u = len(uw)
    out = []
    if u == 0:
        return out
    if uw.ndim == 2:
        d = nt_hamming(uw[:, None, 0], uw[None, :, 0]) + nt_hamming(uw[:, None, 1], uw[None, :, 1])
    else:
        d = nt_hamming(uw[:, None], uw[None, :])
    for i in range(u):
        nb = np.nonzero((d[i] <= distance) & (np.arange(u) != i))[0]
        out.append(nb.tolist())
    return out


def levenshtein(a, b):
    """plain dynamic programme over two symbol lists"""
    prev = list(range(len(b) + 1))
    for i, x in enumerate(a, 1):
        row = [i]
        for j, y in enumerate(b, 1):
            row.append(min(prev[j - 1] + (x != y), prev[j] + 1, row[j - 1] + 1))
        prev = row
    return prev[-1]


def unpack(w, n):
    return [(int(w) >> (2 * (n - 1 - i))) & 3 for i in range(n)]


def edit_adjacency(uw, n, distance):
    """ascending neighbour lists under Levenshtein distance (small U only; two-word words: rows [hi, lo], the first
    n - 32 nucleotides in hi)"""
    if uw.ndim == 2:
        syms = [unpack(hi, n - 32) + unpack(lo, 32) for hi, lo in uw.tolist()]
    else:
        syms = [unpack(w, n) for w in uw.tolist()]
    u = len(syms)
    out = [[] for _ in range(u)]
    for i in range(u):
        for j in range(i + 1, u):
            if levenshtein(syms[i], syms[j]) <= distance:
                out[i].append(j)
                out[j].append(i)
    return out


def cluster(counts, nbrs, maximum=False, order=None):
    """findClusters (src/humid.cc:176-189) + src/cluster.cc, literal recursion."""
    sys.setrecursionlimit(max(10000, 4 * len(counts) + 100))
    u = len(counts)
    cl = [0] * u
    info = {}  # id -> dict(size, maxCount, maxLeaf)

    def assign_leaf(l, c):
        cl[l] = c
        info[c]["size"] += int(counts[l])

    def update_max(l, c):
        if int(counts[l]) > info[c]["maxCount"]:
            info[c]["maxLeaf"] = l
            info[c]["maxCount"] = int(counts[l])

    def max_neighbour(l):
        i = 0
        while i < len(nbrs[l]):
            nb = nbrs[l][i]
            i += 1
            if cl[nb] == 0 and int(counts[nb]) >= 2 * int(counts[l]):
                l = nb
                i = 0
        return l

    def assign_dir(l, c):
        assign_leaf(l, c)
        for nb in nbrs[l]:
            if cl[nb] == 0 and int(counts[l]) >= 2 * int(counts[nb]):
                assign_dir(nb, c)

    def assign_max(l, c):
        assign_leaf(l, c)
        update_max(l, c)
        for nb in nbrs[l]:
            if cl[nb] == 0:
                assign_max(nb, c)

    cid = 1
    for l in (order if order is not None else range(u)):
        if cl[l] == 0:
            info[cid] = dict(size=0, maxCount=0, maxLeaf=-1)
            if maximum:
                assign_max(l, cid)
            else:
                node = max_neighbour(l)
                update_max(node, cid)
                assign_dir(node, cid)
            cid += 1
    return cl, info


def dedup(words, filtered, distance=1, maximum=False, edit_nt=0):
    """(cluster_id[N], keep[N], detail) for a small read set.  edit_nt = word length: Levenshtein
    instead of Hamming neighbours (-e)."""
    words = np.asarray(words, dtype=np.uint64)
    filtered = np.asarray(filtered, dtype=np.uint8)
    uw, cnt = unique_counts(words, filtered)
    nbrs = edit_adjacency(uw, edit_nt, distance) if edit_nt else adjacency(uw, distance)
    cl, info = cluster(cnt, nbrs, maximum)
    key = (lambda w: tuple(int(x) for x in w)) if words.ndim == 2 else int
    rank = {key(w): i for i, w in enumerate(uw.tolist())}
    n = len(words)
    cid = np.zeros(n, dtype=np.uint32)
    keep = np.zeros(n, dtype=np.uint8)
    visited = set()
    for r in range(n):
        if filtered[r]:
            continue
        l = rank[key(words[r])]
        c = cl[l]
        cid[r] = c
        if c not in visited and info[c]["maxLeaf"] == l:
            keep[r] = 1
            visited.add(c)
    detail = dict(unique=uw, count=cnt, nbrs=nbrs, leaf_cluster=cl, info=info)
    return cid, keep, detail
