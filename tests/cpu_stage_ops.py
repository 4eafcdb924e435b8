"""CPU stand-in for humid_amd.sharded.HipStageOps, built on numpy and the oracle.

TEST INFRASTRUCTURE: lets the multi-rank orchestration (humid_amd/sharded.py) run under gloo
on a CPU-only box.  Same method signatures as HipStageOps; tensors are CPU torch tensors."""
import numpy as np
import torch

from oracle import pyoracle as orc


class CpuStageOps:
    def histogram(self, g_w, g_f, word_nt, bits):
        w = g_w.numpy().view(np.uint64)
        f = g_f.numpy()
        shift = np.uint64(2 * word_nt - bits)
        b = (w[f == 0] >> shift).astype(np.int64)
        return torch.from_numpy(np.bincount(b, minlength=1 << bits).astype(np.int32))

    def count(self, g_w, g_f, word_nt, lo, hi, expected):
        w = g_w.numpy().view(np.uint64)
        f = g_f.numpy()
        if lo > hi:
            own = np.zeros(len(w), dtype=bool)
        else:
            own = (f == 0) & (w >= np.uint64(lo)) & (w <= np.uint64(hi))
        assert int(own.sum()) <= expected
        self.read_idx = np.nonzero(own)[0]
        uw, first, inv, cnt = np.unique(w[own], return_index=True, return_inverse=True, return_counts=True)
        self.uw, self.cnt = uw, cnt.astype(np.int32)
        self.first = self.read_idx[first] if len(first) else np.zeros(0, np.int64)
        self.inv = inv
        return len(uw), int(own.sum())

    def unique(self):
        return (torch.from_numpy(self.uw.view(np.int64).copy()), torch.from_numpy(self.cnt.copy()))

    def graph(self, g_word, g_cnt, word_nt, distance, method):
        uw = g_word.numpy().view(np.uint64)
        cnt = g_cnt.numpy().astype(np.int64)
        assert np.all(uw[1:] > uw[:-1]), "global unique array must be strictly ascending"
        reads = np.repeat(uw, cnt)
        p = orc.Pipeline(word_nt)
        p.read_data(reads, np.zeros(len(reads), np.uint8))
        p.find_hamming_neighbours(distance)
        p.find_clusters(bool(method))
        lv = p.leaves()
        s = dict(clusters=p.n_clusters, edges=p.n_edges, nonsingle=int((lv["degree"] > 0).sum()))
        return (torch.from_numpy(lv["cluster_id"].astype(np.int32)),
                torch.from_numpy(lv["is_max_leaf"].copy()), s)

    def map(self, l_cid, l_ismax, out_cid, out_keep):
        out_cid.zero_()
        out_keep.zero_()
        if len(self.read_idx) == 0:
            return
        cid = l_cid.numpy()[self.inv]
        keep = (l_ismax.numpy()[self.inv] != 0) & (self.first[self.inv] == self.read_idx)
        out_cid.numpy()[self.read_idx] = cid
        out_keep.numpy()[self.read_idx] = keep.astype(np.uint8)

    # ---- dense count / map ----
    def count_dense(self, g_w, g_f, word_nt, lo, hi, shard_begin):
        if g_f is None:                       # exchange mode: every read is owned
            g_f = torch.zeros(len(g_w), dtype=torch.uint8)
        u, usable = self.count(g_w, g_f, word_nt, lo, hi, len(g_w))
        idx = self.read_idx
        counts = [int(((idx >= shard_begin[q]) & (idx < shard_begin[q + 1])).sum())
                  for q in range(len(shard_begin) - 1)]
        return u, usable, counts

    def map_dense(self, l_cid, l_ismax):
        pk, _ = self.owned_results(l_cid, l_ismax, [0, 1 << 62])
        return pk

    # ---- partitioned pair search ----
    def pairs(self, g_word, word_nt, distance, part_rank, part_world):
        uw = g_word.numpy().view(np.uint64)
        p = orc.Pipeline(word_nt)
        p.read_data(uw, np.zeros(len(uw), np.uint8))
        p.find_hamming_neighbours(distance)
        off, idx = p.adjacency()
        src = np.repeat(np.arange(len(uw), dtype=np.int64), np.diff(off.astype(np.int64)))
        dst = idx.astype(np.int64)
        m = src < dst
        a, b = src[m], dst[m]
        mine = (a * 7 + b) % part_world == part_rank        # any disjoint cover of the pairs will do
        return torch.from_numpy(((a[mine] << 32) | b[mine]).astype(np.int64))

    def pairs_edit(self, g_word, word_nt, distance, part_rank, part_world):
        uw = g_word.numpy().view(np.uint64)
        p = orc.Pipeline(word_nt)
        p.read_data(uw, np.zeros(len(uw), np.uint8))
        p.find_edit_neighbours(distance)
        off, idx = p.adjacency()
        src = np.repeat(np.arange(len(uw), dtype=np.int64), np.diff(off.astype(np.int64)))
        dst = idx.astype(np.int64)
        m = src < dst
        a, b = src[m], dst[m]
        mine = ((a * 7 + b) % part_world == part_rank) | ((a + b) % 5 == 0)   # shares may overlap
        return torch.from_numpy(((a[mine] << 32) | b[mine]).astype(np.int64))

    def unique_edges(self, edges, n_unique):
        return torch.from_numpy(np.unique(edges.numpy().astype(np.int64)))

    def graph_edges(self, g_word, g_cnt, edges, word_nt, distance, method):
        cnt = g_cnt.numpy().astype(np.int64)
        e = np.sort(edges.numpy().astype(np.int64))             # (a, b) ascending: lists come out ascending
        g = orc.Graph(cnt)
        for x in e.tolist():
            g.link(x >> 32, x & 0xffffffff)
        nc = g.find_clusters(bool(method))
        lc, size, mc, ml = g.export(nc)
        ismax = np.zeros(len(cnt), dtype=np.uint8)
        ismax[ml[ml >= 0]] = 1
        nonsingle = len(np.unique(np.concatenate([e >> 32, e & 0xffffffff]))) if len(e) else 0
        return (torch.from_numpy(lc.astype(np.int32)), torch.from_numpy(ismax),
                dict(clusters=nc, edges=len(e), nonsingle=nonsingle))

    # ---- exchange mode: a plain pigeonhole plan of d+1 single-segment combinations ----
    @staticmethod
    def _segments(word_nt, distance):
        k = distance + 1
        base, rem = divmod(word_nt, k)
        segs, pos = [], 0
        for t in range(k):
            ln = base + (1 if t < rem else 0)
            segs.append((2 * (word_nt - pos - ln), 2 * ln))       # (shift, width in bits)
            pos += ln
        return segs

    def plan_info(self, word_nt, distance, plan_unique):
        if distance >= word_nt:
            return 1, 0
        segs = self._segments(word_nt, distance)
        return len(segs), segs[0][1]

    @staticmethod
    def _seg_key(w, seg):
        shift, width = seg
        return (w >> np.uint64(shift)) & np.uint64((1 << width) - 1)

    def combo_route(self, l_word, l_cnt, id_base, word_nt, distance, plan_unique, combo, n_ranks):
        w = l_word.numpy().view(np.uint64)
        cnt = l_cnt.numpy().astype(np.int64)
        key = self._seg_key(w, self._segments(word_nt, distance)[combo])
        owner = ((key * np.uint64(0x9e3779b97f4a7c15)) >> np.uint64(40)) % np.uint64(n_ranks)
        order = np.argsort(owner, kind="stable")
        second = (id_base + order).astype(np.int64) | (cnt[order] << 32)
        items = np.stack([w[order].view(np.int64), second], axis=1)
        counts = [int((owner == q).sum()) for q in range(n_ranks)]
        return torch.from_numpy(np.ascontiguousarray(items)), counts

    def pairs_keyed(self, items, interleaved, id_base, l_cnt, word_nt, distance, plan_unique, combo):
        if interleaved:
            a = items.numpy()
            w = a[:, 0].view(np.uint64)
            ids, cnt = a[:, 1] & 0xffffffff, (a[:, 1] >> 32) & 0xffffffff
        else:
            assert combo == 0
            w = items.numpy().view(np.uint64)
            ids = id_base + np.arange(len(w), dtype=np.int64)
            cnt = l_cnt.numpy().astype(np.int64)
        segs = self._segments(word_nt, distance)
        key = self._seg_key(w, segs[combo])
        order = np.argsort(key, kind="stable")
        w, ids, cnt, key = w[order], ids[order], cnt[order], key[order]
        out = []
        start = 0
        m55 = np.uint64(0x5555555555555555)
        for end in list(np.nonzero(key[1:] != key[:-1])[0] + 1) + [len(w)]:
            for i in range(start, end):
                x = w[i] ^ w[i + 1:end]
                y = (x | (x >> np.uint64(1))) & m55
                ham = np.array([bin(int(v)).count("1") for v in y], dtype=np.int64)
                ok = ham <= distance
                for q in range(combo):                                   # found by an earlier combination
                    ok &= self._seg_key(x, segs[q]) != 0
                for j in np.nonzero(ok)[0]:
                    a_, b_ = (int(ids[i]), int(cnt[i])), (int(ids[i + 1 + j]), int(cnt[i + 1 + j]))
                    lo_, hi_ = min(a_, b_), max(a_, b_)
                    out.append(((lo_[0] << 32) | hi_[0], lo_[1] | (hi_[1] << 32)))
            start = end
        return torch.from_numpy(np.array(out, dtype=np.int64).reshape(-1, 2))

    def compact_nodes(self, records):
        rec = records.numpy().astype(np.int64)
        e, cc = rec[:, 0], rec[:, 1]
        a, b = e >> 32, e & 0xffffffff
        nodes = np.unique(np.concatenate([a, b]))
        pa, pb = np.searchsorted(nodes, a), np.searchsorted(nodes, b)
        cnt = np.zeros(len(nodes), dtype=np.int64)
        cnt[pa] = cc & 0xffffffff
        cnt[pb] = (cc >> 32) & 0xffffffff
        return (torch.from_numpy(nodes.astype(np.uint32).view(np.int32)),
                torch.from_numpy(((pa << 32) | pb).astype(np.int64)), torch.from_numpy(cnt.astype(np.int32)))

    # ---- dense result return ----
    max_ranks_dense = 16

    def owned_results(self, l_cid, l_ismax, shard_begin):
        idx = self.read_idx
        if len(idx) == 0:
            return torch.zeros(0, dtype=torch.int32), [0] * (len(shard_begin) - 1)
        cid = l_cid.numpy()[self.inv].astype(np.uint32)
        keep = (l_ismax.numpy()[self.inv] != 0) & (self.first[self.inv] == idx)
        packed = (cid | (keep.astype(np.uint32) << np.uint32(31))).view(np.int32)
        counts = [int(((idx >= shard_begin[q]) & (idx < shard_begin[q + 1])).sum())
                  for q in range(len(shard_begin) - 1)]
        assert sum(counts) == len(idx)
        return torch.from_numpy(packed.copy()), counts

    def owner_perm(self, d_w, d_f, ranges):
        w = d_w.numpy().view(np.uint64)
        f = d_f.numpy()
        P = len(ranges)
        owner = np.full(len(w), P, dtype=np.int64)
        for q, (lo, hi, _) in enumerate(ranges):
            if lo <= hi:
                owner[(f == 0) & (w >= np.uint64(lo)) & (w <= np.uint64(hi))] = q
        perm = np.argsort(owner, kind="stable").astype(np.int32)
        counts = [int((owner == q).sum()) for q in range(P)]
        return torch.from_numpy(perm), counts

    def route(self, d_w, d_f, ranges, send_counts):
        perm, counts = self.owner_perm(d_w, d_f, ranges)
        assert counts == list(send_counts), (counts, send_counts)
        n = sum(counts)
        return d_w[perm[:n].long()], perm

    def scatter(self, perm, recv, out_cid, out_keep):
        out_cid.zero_()
        out_keep.zero_()
        n = recv.numel()
        if n:
            p = perm.numpy()[:n]
            t = recv.numpy().view(np.uint32)
            out_cid.numpy()[p] = (t & np.uint32(0x7fffffff)).astype(np.int32)
            out_keep.numpy()[p] = (t >> np.uint32(31)).astype(np.uint8)
