"""Multi-GPU driver of the hot path: one process per GPU, torch.distributed over RCCL/xGMI.

The read set is sharded over the ranks in input order.  Two orchestrations share the stage entry
points of include/humid_hip.h:

mode "exchange" (default, 2..16 ranks) -- every word travels to the rank that owns its VALUE range;
per-rank work and traffic stay constant as ranks are added (weak scaling):

  1. local histogram of the top word bits, all-gathered (P x 16 KB) -> P ordered, balanced value
     ranges, cut at boundaries of the prefix combination of the pigeonhole plan, and every split size
     of the word exchange;
  2. all-to-all of the usable words to their range owners (8 B per read leaves the rank once);
  3. the owner counts its words (LDS tables, as on one GPU); its ascending unique array is a slice
     of Trie::walk() order, global index = sum of the lower ranks' unique counts + local index;
  4. neighbour pairs: the prefix combination is local to a range; for each other combination the
     unique words go to the rank that owns their combination key (all-to-all of (word, index),
     12..16 B per unique word) and are compared there; pairs come out in global indices;
  5. all-gather of the pairs (~2 % of the reads; each carries the counts of its two endpoints) ->
     compact graph over the pairs' endpoints only, clustered replicated (singletons never enter: a
     singleton is its own cluster and its own maxLeaf); cluster ids from closed-form prefix counts;
  6. per-read results at the owner (humid_stage_map_dense), all-to-all back (4 B per read), scatter.

mode "allgather" (BASELINE.json's literal wording; also the fallback for > 16 ranks and for plans
without a usable prefix) -- every rank sees every word.  Per pass:

  1. RCCL all-gather of the packed words (+ filtered flags): every rank sees every candidate
     neighbour (BASELINE.json north_star).  This is the path's one real exchange step.
  2. every rank histograms the gathered words (same input -> same result on all ranks, no
     collective) and derives P disjoint, ordered value ranges with balanced read counts;
  3. humid_stage_count: rank r counts only the words of range r (the random-access work, the
     bulk of the single-GPU time, is divided by P); ranges are ordered, so the concatenation of
     the per-rank sorted unique arrays IS Trie::walk() order;
  4. all-gather of the unique (word, count) arrays (U is ~N/4, far smaller than step 1);
  5. neighbours + clusters over the global unique array: every rank searches its SHARE of the
     pairs (humid_stage_pairs), the shares are all-gathered (few: ~2 % of N) and every rank
     builds components and clusters from the same complete list (humid_stage_graph_edges);
  6. result return: the owner of a word emits the packed results of its reads as dense
     per-shard streams (humid_stage_owned_results), one all-to-all moves 4 B per read, the home
     rank scatters them (humid_stage_owner_perm / humid_stage_scatter).  Both sides derive the
     split sizes from the value ranges, so the streams carry no indices.  (Fallback for > 16
     ranks: humid_stage_map + reduce-scatter of N-sized arrays.)

With the HIP ops the exchange mode is ONE library call per pass (humid_dedup_run_exchange): the stage
sequence runs inside libhumid_hip.so and this module only supplies the two humid_comm callbacks
(HipStageOps.run_exchange).  The stage-by-stage form below drives the same entry points from Python
(HUMID_PY_ORCHESTRATION=1; the oracle-backed CPU ops of the gloo tests always take it).

The compute is behind an `ops` object: HipStageOps (libhumid_hip.so through the C ABI) in
production; the CPU tests drive the same orchestration over gloo with an oracle-backed ops
object that lives under tests/ (this package never imports the oracle).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from .api import Context, HumidError

HIST_BITS = 12


class _DevArray:
    """zero-copy view of ctx-owned device memory for torch (CUDA array interface)."""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr,
                                         "data": (int(ptr), False), "version": 2}


def _wrap(ptr, n, typestr, dtype, device):
    if n == 0 or not ptr:
        return torch.empty(0, dtype=dtype, device=device)
    return torch.as_tensor(_DevArray(ptr, n, typestr), device=device).view(dtype)


class HipStageOps(Context):
    """The stage entry points of include/humid_hip.h on this rank's GPU."""

    def __init__(self, device: int):
        self.device = torch.device("cuda", device)
        # The context works on a torch stream of its own.  ShardedDedup.run makes it the current
        # stream, so torch ops, collectives and the library's kernels are ordered by the stream
        # itself; a caller on another stream is ordered through events (_enter / _exit) -- no host
        # synchronisation either way.
        self.tstream = torch.cuda.Stream(self.device)
        super().__init__(device=device, stream=self.tstream.cuda_stream)

    def _enter(self):
        cur = torch.cuda.current_stream(self.device)
        if cur != self.tstream:
            self.tstream.wait_stream(cur)

    def _exit(self):
        cur = torch.cuda.current_stream(self.device)
        if cur != self.tstream:
            cur.wait_stream(self.tstream)

    def _call(self, fn, *args):
        """one stage entry point, ordered after the caller's stream and before its next op"""
        self._enter()
        self._check(fn(self._h, *args))
        self._exit()

    @staticmethod
    def _p(t):
        return C.c_void_p(t.data_ptr()) if t is not None and t.numel() else None

    def histogram(self, g_w, g_f, word_nt, bits):
        hist = torch.zeros(1 << bits, dtype=torch.int32, device=self.device)
        self._call(self._lib.humid_stage_histogram, self._p(g_w), self._p(g_f), g_f.numel(), word_nt, bits,
                   C.c_void_p(hist.data_ptr()))             # (g_f: one flag per read; g_w has two entries per read beyond 32 nt)
        return hist

    def count_dense(self, g_w, g_f, word_nt, lo, hi, shard_begin):
        """compact this rank's reads, count them with the LDS tables; also the send split sizes.
        g_f None (with the full value range): every read is owned, no compaction pass"""
        n = len(shard_begin) - 1
        sb = (C.c_uint64 * (n + 1))(*shard_begin)
        counts = (C.c_uint64 * n)()
        nu, ns = C.c_uint64(), C.c_uint64()
        self.set_option("count_mode", 0)
        self._wpr = 2 if word_nt > 32 else 1
        self._call(self._lib.humid_stage_count_dense, self._p(g_w), self._p(g_f), g_w.numel() // self._wpr, word_nt,
                   C.c_uint64(lo), C.c_uint64(hi), sb, n, counts, C.byref(nu), C.byref(ns))
        self._u = nu.value
        return nu.value, ns.value, [int(x) for x in counts]

    def map_dense(self, l_cid, l_ismax):
        pp = C.c_void_p()
        n = C.c_uint64()
        self._call(self._lib.humid_stage_map_dense, self._p(l_cid), self._p(l_ismax), C.byref(pp), C.byref(n))
        return _wrap(pp.value, n.value, "<i4", torch.int32, self.device)

    def count(self, g_w, g_f, word_nt, lo, hi, expected):
        self.set_option("count_mode", 1)      # partial range + slot_of_read over N: global table
        nu, ns = C.c_uint64(), C.c_uint64()
        self._wpr = 1
        self._n_reads = g_w.numel()
        self._call(self._lib.humid_stage_count, self._p(g_w), self._p(g_f), g_w.numel(), word_nt,
                   C.c_uint64(lo), C.c_uint64(hi), expected, C.byref(nu), C.byref(ns))
        self._u = nu.value
        return nu.value, ns.value

    def unique(self):
        pw, pc, pf = C.c_void_p(), C.c_void_p(), C.c_void_p()
        self._call(self._lib.humid_stage_unique, C.byref(pw), C.byref(pc), C.byref(pf))
        w = _wrap(pw.value, self._u * getattr(self, "_wpr", 1), "<i8", torch.int64, self.device)    # (two entries per word beyond 32 nt)
        c = _wrap(pc.value, self._u, "<i4", torch.int32, self.device)
        return w, c

    def graph(self, g_word, g_cnt, word_nt, distance, method):
        pc, pm = C.c_void_p(), C.c_void_p()
        s = _lib.HumidSummary()
        n = g_cnt.numel()
        self._call(self._lib.humid_stage_graph, self._p(g_word), self._p(g_cnt), n, word_nt, distance,
                   method, C.byref(pc), C.byref(pm), C.byref(s))
        cid = _wrap(pc.value, n, "<i4", torch.int32, self.device)
        ismax = _wrap(pm.value, n, "|u1", torch.uint8, self.device)
        return cid, ismax, s.asdict()

    def pairs(self, g_word, word_nt, distance, part_rank, part_world):
        """this rank's share of the neighbour pairs: int64 tensor of (smaller << 32 | larger)"""
        pe = C.c_void_p()
        ne = C.c_uint64()
        self._call(self._lib.humid_stage_pairs, self._p(g_word), g_word.numel(), word_nt, distance,
                   part_rank, part_world, C.byref(pe), C.byref(ne))
        return _wrap(pe.value, ne.value, "<i8", torch.int64, self.device)

    def graph_edges(self, g_word, g_cnt, edges, word_nt, distance, method):
        pc, pm = C.c_void_p(), C.c_void_p()
        s = _lib.HumidSummary()
        n = g_cnt.numel()
        self._call(self._lib.humid_stage_graph_edges, self._p(g_word), self._p(g_cnt), n, self._p(edges),
                   edges.numel(), word_nt, distance, method, C.byref(pc), C.byref(pm), C.byref(s))
        cid = _wrap(pc.value, n, "<i4", torch.int32, self.device)
        ismax = _wrap(pm.value, n, "|u1", torch.uint8, self.device)
        return cid, ismax, s.asdict()

    def owned_results(self, l_cid, l_ismax, shard_begin):
        """dense packed results of the reads this rank counted + send split sizes per shard"""
        n = len(shard_begin) - 1
        sb = (C.c_uint64 * (n + 1))(*shard_begin)
        counts = (C.c_uint64 * n)()
        pp = C.c_void_p()
        self._call(self._lib.humid_stage_owned_results, self._p(l_cid), self._p(l_ismax), sb, n,
                   C.byref(pp), counts)
        counts = [int(x) for x in counts]
        return _wrap(pp.value, sum(counts), "<i4", torch.int32, self.device), counts

    def owner_perm(self, d_w, d_f, ranges, word_nt=32):
        """owner-major stable order of this rank's own reads + receive split sizes per owner
        (word_nt > 32: two entries per read in d_w, the ranges are ranges of heads)"""
        P = len(ranges)
        lo = (C.c_uint64 * P)(*[r[0] for r in ranges])
        hi = (C.c_uint64 * P)(*[r[1] for r in ranges])
        counts = (C.c_uint64 * P)()
        pp = C.c_void_p()
        n = d_f.numel()
        self._call(self._lib.humid_stage_owner_perm_wide, self._p(d_w), self._p(d_f), n, word_nt, lo, hi, P,
                   C.byref(pp), counts)
        counts = [int(x) for x in counts]
        return _wrap(pp.value, n, "<i4", torch.int32, self.device), counts

    def scatter(self, perm, recv, out_cid, out_keep):
        self._call(self._lib.humid_stage_scatter, self._p(perm), self._p(recv), recv.numel(),
                   out_cid.numel(), self._p(out_cid), self._p(out_keep))

    max_ranks_dense = 16
    wide_allgather = True       # the all-gather mode's stages take two-word words (33 .. 64 nt)

    # ---- exchange mode ----
    def plan_info(self, word_nt, distance, plan_unique):
        nc, pb = C.c_uint32(), C.c_uint32()
        self._check(self._lib.humid_stage_plan_info(self._h, word_nt, distance, plan_unique,
                                                    C.byref(nc), C.byref(pb)))
        return nc.value, pb.value

    def route_words(self, d_w, n_send):
        """words of the usable reads in the owner-major order of the preceding owner_perm"""
        pr = C.c_void_p()
        self._call(self._lib.humid_stage_route_words, self._p(d_w), n_send, C.byref(pr))
        return _wrap(pr.value, n_send, "<i8", torch.int64, self.device)

    def route(self, d_w, d_f, ranges, send_counts):
        """stable owner-major routing without a host wait: (routed words int64[sum(send_counts)],
        perm int32[n]: routed position -> read index); views of ctx memory"""
        P = len(ranges)
        lo = (C.c_uint64 * P)(*[r[0] for r in ranges])
        hi = (C.c_uint64 * P)(*[r[1] for r in ranges])
        sc = (C.c_uint64 * P)(*send_counts)
        pr, pp = C.c_void_p(), C.c_void_p()
        self._call(self._lib.humid_stage_route, self._p(d_w), self._p(d_f), d_w.numel(), lo, hi, P, sc,
                   C.byref(pr), C.byref(pp))
        n_send = sum(send_counts)
        return (_wrap(pr.value, n_send, "<i8", torch.int64, self.device),
                _wrap(pp.value, d_w.numel(), "<i4", torch.int32, self.device))

    def route_check(self):
        self._call(self._lib.humid_stage_route_check)

    def combo_route(self, l_word, l_cnt, id_base, word_nt, distance, plan_unique, combo, n_ranks):
        """(word, id | count << 32) items of the local unique array in destination-major order:
        int64[n, 2]"""
        pi = C.c_void_p()
        counts = (C.c_uint64 * n_ranks)()
        n = l_word.numel()
        self._call(self._lib.humid_stage_combo_route, self._p(l_word), self._p(l_cnt), n, id_base, word_nt,
                   distance, plan_unique, combo, n_ranks, C.byref(pi), counts)
        return _wrap(pi.value, 2 * n, "<i8", torch.int64, self.device).view(-1, 2), [int(x) for x in counts]

    def pairs_keyed(self, items, interleaved, id_base, l_cnt, word_nt, distance, plan_unique, combo):
        """pair records int64[E, 2]: (smaller id << 32 | larger id, count(smaller) | count(larger) << 32);
        a VIEW of ctx memory that the next call overwrites"""
        pe = C.c_void_p()
        ne = C.c_uint64()
        self._call(self._lib.humid_stage_pairs_keyed, self._p(items), items.shape[0], int(interleaved),
                   id_base, self._p(l_cnt), word_nt, distance, plan_unique, combo, C.byref(pe), C.byref(ne))
        return _wrap(pe.value, 2 * ne.value, "<i8", torch.int64, self.device).view(-1, 2)

    def compact_nodes(self, records):
        """pair records int64[E, 2] -> (nodes int32[M] ascending, compact edges int64[E], counts int32[M])"""
        pn, pc, pk = C.c_void_p(), C.c_void_p(), C.c_void_p()
        nn = C.c_uint64()
        e = records.shape[0]
        self._call(self._lib.humid_stage_compact_nodes, self._p(records), e, 2, C.byref(pn), C.byref(nn),
                   C.byref(pc), C.byref(pk))
        return (_wrap(pn.value, nn.value, "<i4", torch.int32, self.device),
                _wrap(pc.value, e, "<i8", torch.int64, self.device),
                _wrap(pk.value, nn.value, "<i4", torch.int32, self.device))

    def exchange_ids(self, nodes, ccid, cismax, n_clusters, id_base, u_local):
        """cluster id (int32) + maxLeaf flag (uint8) of the local unique words; views of ctx memory"""
        pc, pm = C.c_void_p(), C.c_void_p()
        m = nodes.numel() if nodes is not None else 0
        self._call(self._lib.humid_stage_exchange_ids, self._p(nodes), self._p(ccid), self._p(cismax), m,
                   n_clusters, id_base, u_local, C.byref(pc), C.byref(pm))
        return (_wrap(pc.value, u_local, "<i4", torch.int32, self.device),
                _wrap(pm.value, u_local, "|u1", torch.uint8, self.device))

    def pairs_edit(self, g_word, word_nt, distance, part_rank, part_world):
        """this rank's share of the Levenshtein neighbour pairs (may repeat pairs); a ctx view"""
        pe = C.c_void_p()
        ne = C.c_uint64()
        self._call(self._lib.humid_stage_pairs_edit, self._p(g_word), g_word.numel(), word_nt, distance,
                   part_rank, part_world, C.byref(pe), C.byref(ne))
        return _wrap(pe.value, ne.value, "<i8", torch.int64, self.device)

    def unique_edges(self, edges, n_unique):
        pe = C.c_void_p()
        ne = C.c_uint64()
        self._call(self._lib.humid_stage_unique_edges, self._p(edges), edges.numel(), n_unique,
                   C.byref(pe), C.byref(ne))
        return _wrap(pe.value, ne.value, "<i8", torch.int64, self.device)

    def kernel_ms(self):
        """HIP-event ms of the dominant kernels of the last count_dense / map_dense pair"""
        a, b, m = C.c_float(), C.c_float(), C.c_uint32()
        self._call(self._lib.humid_stage_kernel_ms, C.byref(a), C.byref(b), C.byref(m))
        return dict(ms_k_insert=a.value, ms_k_map=b.value, count_mode_used=m.value & 0xff, records8=bool(m.value >> 8))

    def run_exchange(self, dist, d_w, d_f, d_cid, d_keep, word_nt, distance, method):
        """the whole exchange-mode pass in ONE library call (humid_dedup_run_exchange): the stage sequence
        runs inside the library, torch.distributed only moves the bytes (humid_comm callbacks)"""
        world, rank = dist.get_world_size(), dist.get_rank()
        dev = self.device
        err = []

        def host_all_gather(_user, mine, nbytes, out):
            try:
                if world == 1:                               # a group of one: the table is the rank's own entry
                    C.memmove(out, mine, nbytes)
                    return 0
                src = torch.frombuffer((C.c_uint8 * nbytes).from_address(mine), dtype=torch.uint8)
                inp = src.to(dev) if dev.type == "cuda" else src.clone()
                allb = torch.empty(world * nbytes, dtype=torch.uint8, device=inp.device)
                _all_gather_flat(dist, allb, inp, world)
                host = allb.cpu().numpy()                    # (kept alive until the bytes are copied out)
                C.memmove(out, host.ctypes.data, world * nbytes)
                return 0
            except Exception as e:  # pragma: no cover  (reported through HUMID_E_COMM)
                err.append(e)
                return -1

        def exchange(_user, d_send, so, sb, d_recv, ro, rb, all_gather, _stream):
            try:
                so, sb = [so[q] for q in range(world)], [sb[q] for q in range(world)]
                ro, rb = [ro[q] for q in range(world)], [rb[q] for q in range(world)]
                out = self._bytes(d_recv, ro[-1] + rb[-1])
                if all_gather:                              # the same bytes to everybody
                    inp = self._bytes(d_send, sb[0])
                    if all(x == rb[0] for x in rb):
                        _all_gather_flat(dist, out, inp, world)
                    else:
                        # different lengths, all known (the library gathered them): grouped point-to-point
                        # messages straight into place where the backend has them, else the padded form
                        try:
                            if _bounce(dist, inp):
                                raise NotImplementedError
                            dist.all_to_all([out[ro[q]:ro[q] + rb[q]] for q in range(world)], [inp] * world)
                        except (RuntimeError, NotImplementedError, AttributeError, ValueError):
                            got, _ = _all_gather_var(dist, inp, world)
                            out.copy_(got)
                else:
                    inp = self._bytes(d_send, so[-1] + sb[-1])
                    _all_to_all_v(dist, out, inp, rb, sb, world, rank)
                return 0
            except Exception as e:  # pragma: no cover
                err.append(e)
                return -1

        shm = getattr(self, "shm", None)
        if shm:          # ranks are processes of one node: the small host tables go through shared memory
            gather = _lib.HOST_ALL_GATHER_FN(C.cast(self._lib.humid_shm_all_gather, C.c_void_p).value)
        else:
            gather = _lib.HOST_ALL_GATHER_FN(host_all_gather)
        cm = _lib.HumidComm(shm, rank, world, gather, _lib.EXCHANGE_FN(exchange))
        s = _lib.HumidSummary()
        info = _lib.HumidExchangeInfo()
        self._enter()
        rc = self._lib.humid_dedup_run_exchange(self._h, C.byref(cm), self._p(d_w), self._p(d_f), d_w.shape[0], word_nt,
                                                distance, method, self._p(d_cid), self._p(d_keep), C.byref(s), C.byref(info))
        if err:
            raise err[0]
        self._check(rc)
        self._exit()
        return s.asdict()

    def open_shm(self, dist):
        """host_all_gather through a shared-memory segment when every rank of `dist` is a process of this
        node (humid_shm_*): a few microseconds per gather instead of a collective's launches and waits.
        Quietly stays with the collective when the ranks are spread over nodes, are threads of one process
        (the test stand-in) or HUMID_NO_SHM is set."""
        import os
        import socket
        self.shm = None
        if os.environ.get("HUMID_NO_SHM") or not hasattr(dist, "broadcast_object_list"):
            return False
        try:
            world, rank = dist.get_world_size(), dist.get_rank()
            hosts = [None] * world
            dist.all_gather_object(hosts, (socket.gethostname(), os.getpid()))
            if len({h for h, _ in hosts}) != 1 or len({p for _, p in hosts}) != world:
                return False
            name = ["/humid_%d_%s" % (os.getpid(), os.urandom(4).hex())] if rank == 0 else [None]
            dist.broadcast_object_list(name, src=0)
            h = C.c_void_p()
            rc = self._lib.humid_shm_open(C.byref(h), name[0].encode(), rank, world, 1 << 16)
            oks = [None] * world
            dist.all_gather_object(oks, rc == 0)
            if not all(oks):                       # all or nobody
                if rc == 0:
                    self._lib.humid_shm_close(h)
                return False
            self.shm = h.value
            return True
        except Exception:  # pragma: no cover  (an optional fast path)
            self.shm = None
            return False

    def close(self):
        if getattr(self, "shm", None):
            self._lib.humid_shm_close(C.c_void_p(self.shm))
            self.shm = None
        super().close()

    def _bytes(self, ptr, n):
        """torch uint8 view of ctx-owned (or caller) device memory; views of the persistent buffers are
        kept: the pointers stay the same from pass to pass"""
        if n == 0 or not ptr:
            return torch.empty(0, dtype=torch.uint8, device=self.device)
        key = (int(ptr), int(n))
        cache = self.__dict__.setdefault("_views", {})
        t = cache.get(key)
        if t is None:
            if len(cache) > 64:
                cache.clear()
            t = cache[key] = _wrap(ptr, n, "|u1", torch.uint8, self.device)
        return t

    def map(self, l_cid, l_ismax, out_cid, out_keep):
        self._call(self._lib.humid_stage_map, self._p(l_cid), self._p(l_ismax), out_cid.numel(),
                   self._p(out_cid), self._p(out_keep))


# ------------------------------------------------------------------------------------------
# collectives with a fallback for backends (gloo, CPU tests) that lack the tensor forms
# ------------------------------------------------------------------------------------------
def _bounce(dist, t):
    """gloo moves host memory only: device tensors take a round trip through the host (multi-process
    rehearsals on a box without RCCL peers; never the case under nccl)"""
    try:
        return bool(t.is_cuda) and dist.get_backend() == "gloo"
    except Exception:
        return False


def _all_reduce_sum(dist, t):
    if _bounce(dist, t):
        h = t.cpu()
        dist.all_reduce(h)
        t.copy_(h)
    else:
        dist.all_reduce(t)


def _all_gather_flat(dist, out, inp, world):
    if _bounce(dist, inp):
        parts = [torch.empty(inp.shape, dtype=inp.dtype) for _ in range(world)]
        dist.all_gather(parts, inp.cpu())
        out.copy_(torch.cat([p.reshape(-1) for p in parts]).view_as(out))
        return
    try:
        dist.all_gather_into_tensor(out, inp)
    except (RuntimeError, NotImplementedError):
        parts = [torch.empty_like(inp) for _ in range(world)]
        dist.all_gather(parts, inp)
        out.copy_(torch.cat(parts))


def _reduce_scatter_sum(dist, out, inp, world, rank):
    try:
        if _bounce(dist, inp):
            raise NotImplementedError
        dist.reduce_scatter_tensor(out, inp, op=dist.ReduceOp.SUM)
    except (RuntimeError, NotImplementedError):
        tmp = inp.clone()
        if tmp.dtype == torch.uint8:        # gloo has no uint8 sum
            tmp = tmp.to(torch.int32)
        _all_reduce_sum(dist, tmp)
        n = out.numel()
        out.copy_(tmp[rank * n:(rank + 1) * n].to(out.dtype))


def _all_to_all_v(dist, out, inp, out_splits, in_splits, world, rank):
    """variable all-to-all along dim 0 (rows of a contiguous [n, k] tensor count as one element);
    emulated with an all-gather where the backend (gloo) has no all_to_all"""
    if inp.dim() == 2:
        k = inp.shape[1]
        return _all_to_all_v(dist, out.view(-1), inp.reshape(-1), [x * k for x in out_splits],
                             [x * k for x in in_splits], world, rank)
    if not _bounce(dist, inp):
        try:
            dist.all_to_all_single(out, inp, output_split_sizes=out_splits, input_split_sizes=in_splits)
            return
        except (RuntimeError, NotImplementedError, ValueError):
            pass
    dev = inp.device
    meta = torch.tensor(in_splits, dtype=torch.int64, device=dev)
    metas = torch.empty(world * world, dtype=torch.int64, device=dev)
    _all_gather_flat(dist, metas, meta, world)
    metas = metas.cpu().view(world, world)            # metas[src][dst]
    totals = metas.sum(dim=1).tolist()
    m = max(max(totals), 1)
    pad = torch.zeros(m, dtype=inp.dtype, device=dev)
    pad[:inp.numel()] = inp
    allb = torch.empty(world * m, dtype=inp.dtype, device=dev)
    _all_gather_flat(dist, allb, pad, world)
    o = 0
    for src in range(world):
        off = int(metas[src, :rank].sum())
        cnt = int(metas[src, rank])
        out[o:o + cnt] = allb[src * m + off: src * m + off + cnt]
        o += cnt


def _all_gather_var(dist, t, world, fill=0):
    """all-gather of 1-D tensors of different lengths -> (concatenation in rank order, lengths)"""
    dev = t.device
    n = torch.tensor([t.numel()], dtype=torch.int64, device=dev)
    ns = torch.empty(world, dtype=torch.int64, device=dev)
    _all_gather_flat(dist, ns, n, world)
    ns = ns.cpu().tolist()
    m = max(max(ns), 1)
    pad = torch.full((m,), fill, dtype=t.dtype, device=dev)
    pad[:t.numel()] = t
    allb = torch.empty(world * m, dtype=t.dtype, device=dev)
    _all_gather_flat(dist, allb, pad, world)
    if all(x == m for x in ns):
        return allb, ns
    return torch.cat([allb[q * m:q * m + ns[q]] for q in range(world)]), ns


def _exchange_ids_torch(nodes, ccid, cismax, n_clusters, goff, u_local, dev):
    """humid_stage_exchange_ids restated with torch ops (ops objects without the entry point: the
    CPU stand-in of the gloo tests).  Cluster id = 1 + cluster-creating leaves before it in the whole
    walk = singletons before it + compact creators before it."""
    i64 = dict(dtype=torch.int64, device=dev)
    g = goff + torch.arange(u_local, **i64)
    if nodes is None or nodes.numel() == 0:
        return (1 + g).to(torch.int32), torch.ones(u_local, dtype=torch.uint8, device=dev)
    M = nodes.numel()
    nodes64 = nodes.long() & 0xffffffff
    s0, s1 = torch.searchsorted(nodes64, torch.tensor([goff, goff + u_local], **i64)).tolist()
    li = nodes64[s0:s1] - goff                                # local indices of this rank's endpoints
    ccid64 = ccid.long()
    creator = torch.full((n_clusters,), M, **i64).scatter_reduce_(0, ccid64 - 1, torch.arange(M, **i64), "amin")
    creator_g = nodes64[creator]                              # ascending: ids follow creators
    base_id = creator_g - creator + torch.arange(n_clusters, **i64)   # global cluster id - 1
    is_c = torch.zeros(u_local, **i64)
    is_c[li] = 1
    nb = s0 + torch.cumsum(is_c, 0) - is_c
    cr = torch.searchsorted(creator_g, torch.tensor([goff, goff + u_local], **i64)).tolist()
    is_cr = torch.zeros(u_local, **i64)
    is_cr[creator_g[cr[0]:cr[1]] - goff] = 1
    cb = cr[0] + torch.cumsum(is_cr, 0) - is_cr
    l_cid = 1 + g - nb + cb
    l_cid[li] = 1 + base_id[ccid64[s0:s1] - 1]
    l_ismax = torch.ones(u_local, dtype=torch.uint8, device=dev)
    l_ismax[li] = cismax[s0:s1]
    return l_cid.to(torch.int32), l_ismax


def _order_hint(hist, rng, word_nt, bits):
    """count_order for humid_stage_count_dense from the global top-bits histogram: 1 = the words of
    this value range are spread evenly (word-ordered buckets fit their LDS tables), 0 = clearly
    not, -1 = let the library sample"""
    lo, hi, _ = rng
    if lo > hi:
        return -1
    shift = 2 * word_nt - bits
    b0, b1 = lo >> shift, min(hi >> shift, len(hist) - 1)
    h = hist[b0:b1 + 1].astype(np.float64)
    if len(h) < 4 or h.sum() < 65536:
        return -1
    ratio = float(h.max() / h.mean())
    return 1 if ratio <= 1.25 else (0 if ratio > 2.5 else -1)


def splitters_from_hist(hist: np.ndarray, world: int, word_nt: int, bits: int):
    """P ordered, disjoint, covering value ranges with balanced usable-read counts.
    Returns [(lo, hi_inclusive, expected_reads)] -- identical on every rank."""
    shift = 2 * word_nt - bits
    cum = np.cumsum(hist.astype(np.int64))
    total = int(cum[-1]) if len(cum) else 0
    n_bins = len(hist)
    bounds = [0]
    for k in range(1, world):
        b = int(np.searchsorted(cum, (total * k + world - 1) // world, side="left")) + 1
        bounds.append(min(max(b, bounds[-1]), n_bins))
    bounds.append(n_bins)
    out = []
    top = (1 << 64) - 1
    for r in range(world):
        b0, b1 = bounds[r], bounds[r + 1]
        if b1 <= b0:
            out.append((1, 0, 0))              # empty range
            continue
        lo = b0 << shift
        hi = top if r == world - 1 else (b1 << shift) - 1
        exp = int(cum[b1 - 1] - (cum[b0 - 1] if b0 > 0 else 0))
        out.append((lo, min(hi, top), exp))
    return out


class ShardedDedup:
    """Global deduplication of a read set sharded over the ranks of the default process group."""

    def __init__(self, device: int = 0, word_nt: int = 24, distance: int = 1, method: int = 0,
                 ops=None, dist=None, dense_return: bool = True, partition_search: bool = True,
                 mode: str = None, edit: bool = False):
        # -e: distance <= 1 IS the Hamming search (equal-length words); 2 to 5: the joins of the shifted-segment
        # search are dealt out over the ranks -- inside the library's exchange pass (round 3: the unique words
        # are all-gathered there), or stage by stage in the all-gather mode below
        self.edit = bool(edit) and distance >= 2
        self.edit_in_library = False
        # 33 <= word_nt <= 64 (two int64 per read, tensors of shape [n, 2]): the library's exchange pass only
        import os
        import torch.distributed as tdist
        self.mode = mode or os.environ.get("HUMID_SHARD_MODE", "exchange")
        if self.mode not in ("exchange", "allgather"):
            raise ValueError("mode must be 'exchange' or 'allgather'")
        explicit_mode = mode or os.environ.get("HUMID_SHARD_MODE")
        self.dist = dist or tdist
        self.world = self.dist.get_world_size()
        self.rank = self.dist.get_rank()
        self.word_nt, self.distance, self.method = word_nt, distance, method
        self.ops = ops if ops is not None else HipStageOps(device)
        self.bits = min(HIST_BITS, 2 * word_nt)
        self.dense_return = dense_return
        self.partition_search = partition_search
        self._n_max = None
        self.trace = {} if os.environ.get("HUMID_SHARD_TRACE") else None
        # HUMID_PY_ORCHESTRATION=1 (or a trace): the stage-by-stage Python form of the exchange mode
        # below instead of the library's single call -- same entry points, same results
        self.py_orchestration = bool(os.environ.get("HUMID_PY_ORCHESTRATION")) or self.trace is not None
        if self.edit:
            if hasattr(self.ops, "run_exchange") and not self.py_orchestration and explicit_mode != "allgather":
                self.mode = "exchange"
                self.edit_in_library = True
                self.ops.set_option("edit_distance", 1)
            else:
                self.mode = "allgather"
        if word_nt > 32 and self.world > getattr(self.ops, "max_ranks_dense", 16):
            # refused here, on every rank alike and before any collective (both modes' two-word stages
            # return results as dense per-shard streams, which the library lays out for at most 16 ranks)
            raise NotImplementedError("words longer than 32 nt: at most %d ranks" % getattr(self.ops, "max_ranks_dense", 16))
        self.shm_used = False
        if self.world > 1 and hasattr(self.ops, "open_shm") and not self.py_orchestration:
            self.shm_used = self.ops.open_shm(self.dist)

    def run(self, d_w, d_f, d_cid, d_keep):
        """d_w int64[n_local] packed words, d_f uint8[n_local]; writes d_cid int32[n_local] and
        d_keep uint8[n_local] (torch tensors on this rank's device).  Returns a summary dict."""
        ts = getattr(self.ops, "tstream", None)
        if ts is None:
            return self._run(d_w, d_f, d_cid, d_keep)
        # the library's stream becomes the current stream for the whole pass: torch ops, RCCL
        # collectives and the stage kernels are then ordered by the stream, without host syncs
        cur = torch.cuda.current_stream(d_w.device)
        ts.wait_stream(cur)
        with torch.cuda.stream(ts):
            summ = self._run(d_w, d_f, d_cid, d_keep)
        cur.wait_stream(ts)
        return summ

    def _run(self, d_w, d_f, d_cid, d_keep):
        if self.word_nt > 32:
            # two-word words: the library's exchange pass, or -- asked for by name -- the all-gather mode stage by stage
            if self.mode == "allgather" and getattr(self.ops, "wide_allgather", False) and not self.edit:
                self.mode_used = "allgather"
                return self._run_allgather(d_w, d_f, d_cid, d_keep)
            if not hasattr(self.ops, "run_exchange") or self.py_orchestration or (self.edit and not self.edit_in_library):
                raise NotImplementedError("words longer than 32 nt: the library's exchange pass or the all-gather mode (HIP ops)")
            self.mode_used = "exchange"
            self.summary = self.ops.run_exchange(self.dist, d_w, d_f, d_cid, d_keep, self.word_nt, self.distance, self.method)
            return self.summary
        if self.mode == "exchange" and hasattr(self.ops, "combo_route") and \
                self.world <= getattr(self.ops, "max_ranks_dense", 0):
            _, pbits = self.ops.plan_info(self.word_nt, self.distance, 1)
            if pbits >= 1:          # d >= n has no prefix to cut the value ranges at
                self.mode_used = "exchange"
                return self._run_exchange(d_w, d_f, d_cid, d_keep, min(self.bits, pbits))
        self.mode_used = "allgather"
        return self._run_allgather(d_w, d_f, d_cid, d_keep)

    def _run_exchange(self, d_w, d_f, d_cid, d_keep, bits):
        if hasattr(self.ops, "run_exchange") and not self.py_orchestration:
            summ = self.ops.run_exchange(self.dist, d_w, d_f, d_cid, d_keep, self.word_nt, self.distance, self.method)
            self.summary = summ
            return summ
        dist, P, r, ops = self.dist, self.world, self.rank, self.ops
        dev = d_w.device
        n_local = d_w.numel()
        i64 = dict(dtype=torch.int64, device=dev)
        trace = self.trace
        if trace is not None:
            import time
            if dev.type == "cuda":
                torch.cuda.synchronize()
            t_last = [time.perf_counter()]

        def mark(name):                                   # HUMID_SHARD_TRACE=1: per-phase wall ms
            if trace is None:
                return
            if dev.type == "cuda":
                torch.cuda.synchronize()
            now = time.perf_counter()
            trace[name] = trace.get(name, 0.0) + 1e3 * (now - t_last[0])
            t_last[0] = now
        # ---- 1. per-rank histograms of the top word bits, all-gathered (P x 16 KB): their sum gives
        #         balanced ordered value ranges (cut at prefix boundaries), and since the ranges are
        #         cut at bin boundaries the histograms also say how many reads every rank sends to
        #         every owner -- no second exchange of counts, no host wait in the routing ----
        hist = ops.histogram(d_w, d_f, self.word_nt, bits)
        n_bins = hist.numel()
        all_h = torch.empty(P * n_bins, dtype=hist.dtype, device=dev)
        _all_gather_flat(dist, all_h, hist, P)
        all_host = all_h.cpu().numpy().reshape(P, n_bins).astype(np.int64)
        hist_host = all_host.sum(axis=0)
        ranges = splitters_from_hist(hist_host, P, self.word_nt, bits)
        shift = 2 * self.word_nt - bits
        cum = np.concatenate([np.zeros((P, 1), np.int64), np.cumsum(all_host, axis=1)], axis=1)   # [P, n_bins + 1]

        def in_range(src, q):                             # usable reads of rank src in the range of owner q
            lo, hi, _ = ranges[q]
            if lo > hi:
                return 0
            b0, b1 = lo >> shift, min(hi >> shift, n_bins - 1) + 1
            return int(cum[src, b1] - cum[src, b0])
        send_counts = [in_range(r, q) for q in range(P)]
        recv_counts = [in_range(q, r) for q in range(P)]
        lo_r, hi_r = ranges[r][0], ranges[r][1]
        if lo_r > hi_r:
            lo_r, hi_r = 0, (1 << 64) - 1                 # empty range: nothing arrives
        if hasattr(ops, "set_option"):
            # word-ordered LDS buckets need words that are uniform over this rank's range; the global
            # histogram already says so (saves the library's own sampling pass and its host wait)
            ops.set_option("count_order", _order_hint(hist_host, ranges[r], self.word_nt, bits))
        mark("1_ranges")
        # ---- 2. usable words -> owner of their range (stable: input order inside every block) ----
        n_send = sum(send_counts)
        if hasattr(ops, "route"):
            send_w, perm = ops.route(d_w, d_f, ranges, send_counts)
        else:
            perm, sc2 = ops.owner_perm(d_w, d_f, ranges)              # owner-major, filtered reads last
            assert sc2 == send_counts
            send_w = d_w[perm[:n_send].long()] if n_send else torch.empty(0, **i64)
        n_recv = sum(recv_counts)
        recv_w = torch.empty(n_recv, **i64)
        _all_to_all_v(dist, recv_w, send_w, recv_counts, send_counts, P, r)
        mark("2_route_words")
        # ---- 3. exact counts of the received words (all usable, all in this rank's range) ----
        u_local, usable_local, _ = ops.count_dense(recv_w, None, self.word_nt, lo_r, hi_r, [0, n_recv])
        metas = torch.empty(3 * P, **i64)
        _all_gather_flat(dist, metas, torch.tensor([u_local, usable_local, n_local], **i64), P)
        metas = metas.cpu().view(P, 3)
        u_all = metas[:, 0].tolist()
        u_total, goff = sum(u_all), sum(u_all[:r])
        summ = dict(total=int(metas[:, 2].sum()), usable=int(metas[:, 1].sum()), unique=u_total,
                    clusters=u_total, edges=0, nonsingle=0)
        if u_total >= (1 << 32) - 1:
            raise HumidError(-5, "more than 2^32-2 unique words in total")
        lw, lc = ops.unique() if u_local else (torch.empty(0, **i64), torch.empty(0, dtype=torch.int32, device=dev))
        mark("3_count")
        # ---- 4. neighbour pairs in global unique indices, each with the counts of its endpoints ----
        e_parts = []
        if self.distance > 0 and u_total > 1:
            n_combos, _ = ops.plan_info(self.word_nt, self.distance, u_total)
            if u_local > 1:
                e_parts.append(ops.pairs_keyed(lw, False, goff, lc, self.word_nt, self.distance, u_total, 0).clone())
            for cb in range(1, n_combos):
                items, sc = ops.combo_route(lw, lc, goff, self.word_nt, self.distance, u_total, cb, P)
                cm = torch.empty(P * P, **i64)
                _all_gather_flat(dist, cm, torch.tensor(sc, **i64), P)
                rc = cm.cpu().view(P, P)[:, r].tolist()
                got = torch.empty((sum(rc), 2), **i64)
                _all_to_all_v(dist, got, items, rc, sc, P, r)
                if got.shape[0] > 1:
                    e_parts.append(ops.pairs_keyed(got, True, 0, None, self.word_nt, self.distance, u_total, cb).clone())
        e_loc = torch.cat(e_parts) if e_parts else torch.empty((0, 2), **i64)
        e_flat, _ = _all_gather_var(dist, e_loc.view(-1), P)
        e_all = e_flat.view(-1, 2)
        mark("4_pairs")
        # ---- 5. compact graph over the pairs' endpoints; ids by closed-form prefix counts ----
        if e_all.shape[0]:
            # ascending global indices of the leaves with neighbours, edges over positions in that
            # list, endpoint counts (they travelled with the pairs): the singletons never enter
            nodes, cedges, cnt_c = ops.compact_nodes(e_all)
            M = nodes.numel()
            ccid, cismax, gs = ops.graph_edges(nodes, cnt_c, cedges, self.word_nt, self.distance, self.method)
            C_c = int(gs["clusters"])
            summ.update(clusters=u_total - M + C_c, edges=int(e_all.shape[0]), nonsingle=M)
            for k, v in gs.items():
                if k.startswith("ms_"):
                    summ[k] = v
        else:
            nodes = ccid = cismax = None
            C_c = 0
        if hasattr(ops, "exchange_ids"):
            l_cid, l_ismax = ops.exchange_ids(nodes, ccid, cismax, C_c, goff, u_local)
        else:
            l_cid, l_ismax = _exchange_ids_torch(nodes, ccid, cismax, C_c, goff, u_local, dev)
        if summ["clusters"] >= (1 << 31):
            raise HumidError(-5, "cluster ids exceed 31 bits")
        mark("5_clusters")
        # ---- 6. per-read results at the owner, back to the home shards ----
        packed = ops.map_dense(l_cid, l_ismax)
        ret = torch.empty(n_send, dtype=torch.int32, device=dev)
        _all_to_all_v(dist, ret, packed, send_counts, recv_counts, P, r)
        ops.scatter(perm, ret, d_cid, d_keep)
        mark("6_results")
        self.summary = summ
        return summ

    def _run_allgather(self, d_w, d_f, d_cid, d_keep):
        dist, P, r = self.dist, self.world, self.rank
        if hasattr(self.ops, "set_option"):
            self.ops.set_option("count_order", -1)
        dev = d_w.device
        # words of 33 .. 64 nt: two int64 per read / unique word; value ranges are ranges of HEADS (the top 64 bits of
        # a word's value), i.e. the histogram and the splitters are those of 32-nt words
        wpr = 2 if self.word_nt > 32 else 1
        range_nt = 32 if wpr == 2 else self.word_nt
        if wpr == 2 and (self.edit or not self.dense_return or P > getattr(self.ops, "max_ranks_dense", 0)):
            raise NotImplementedError("words longer than 32 nt in the all-gather mode: dense return, <= 16 ranks, no -e")
        n_local = d_f.numel()
        if self._n_max is None:                      # shard sizes are fixed per instance
            t = torch.tensor([n_local], dtype=torch.int64, device=dev)
            sizes = torch.empty(P, dtype=torch.int64, device=dev)
            _all_gather_flat(dist, sizes, t, P)
            self._sizes = sizes.cpu().tolist()
            self._n_max = max(self._sizes)
        n_max = self._n_max
        # ---- 1. all-gather of the packed words (padding reads are flagged filtered) ----
        if n_local == n_max:
            pw, pf = d_w, d_f
        else:
            pw = torch.zeros(n_max * wpr, dtype=torch.int64, device=dev)
            pf = torch.ones(n_max, dtype=torch.uint8, device=dev)
            pw[:n_local * wpr] = d_w.reshape(-1)
            pf[:n_local] = d_f
        pw = pw.reshape(-1)
        g_w = torch.empty(P * n_max * wpr, dtype=torch.int64, device=dev)
        g_f = torch.empty(P * n_max, dtype=torch.uint8, device=dev)
        _all_gather_flat(dist, g_w, pw, P)
        _all_gather_flat(dist, g_f, pf, P)
        # ---- 2. balanced ordered ranges (replicated, deterministic) ----
        hist = self.ops.histogram(g_w, g_f, self.word_nt, self.bits)
        ranges = splitters_from_hist(hist.cpu().numpy(), P, range_nt, self.bits)
        lo, hi, exp = ranges[r]
        # ---- 3. exact counts of this rank's range ----
        shard_begin = [q * n_max for q in range(P + 1)]
        use_dense = (self.dense_return and hasattr(self.ops, "count_dense")
                     and P <= getattr(self.ops, "max_ranks_dense", 0))
        send_counts = None
        if use_dense:
            u_local, usable_local, send_counts = self.ops.count_dense(g_w, g_f, self.word_nt, lo, hi, shard_begin)
        else:
            u_local, usable_local = self.ops.count(g_w, g_f, self.word_nt, lo, hi, max(exp, 1))
        meta = torch.tensor([u_local, usable_local], dtype=torch.int64, device=dev)
        metas = torch.empty(2 * P, dtype=torch.int64, device=dev)
        _all_gather_flat(dist, metas, meta, P)
        metas = metas.cpu().view(P, 2)
        u_all = metas[:, 0].tolist()
        usable = int(metas[:, 1].sum())
        u_max, u_total = max(u_all), sum(u_all)
        goff = sum(u_all[:r])
        total_reads = sum(self._sizes)
        summ = dict(total=total_reads, usable=usable, unique=u_total, clusters=0, edges=0, nonsingle=0)
        if u_total > 0:
            # ---- 4. all-gather of the per-range unique arrays -> global walk order ----
            lw, lc = self.ops.unique()
            sw = torch.zeros(u_max * wpr, dtype=torch.int64, device=dev)
            sc = torch.zeros(u_max, dtype=torch.int32, device=dev)
            sw[:u_local * wpr] = lw
            sc[:u_local] = lc
            aw = torch.empty(P * u_max * wpr, dtype=torch.int64, device=dev)
            ac = torch.empty(P * u_max, dtype=torch.int32, device=dev)
            _all_gather_flat(dist, aw, sw, P)
            _all_gather_flat(dist, ac, sc, P)
            if all(u == u_max for u in u_all):
                gw, gc = aw, ac
            else:
                gw = torch.cat([aw[q * u_max * wpr:(q * u_max + u_all[q]) * wpr] for q in range(P)])
                gc = torch.cat([ac[q * u_max:q * u_max + u_all[q]] for q in range(P)])
            # ---- 5. neighbours + clusters over the global unique array ----
            if self.edit:
                # Levenshtein neighbours: every rank runs its share of the joins over the whole unique
                # array; shares can repeat a pair, the gathered list is made unique
                e_loc = self.ops.pairs_edit(gw, self.word_nt, self.distance, r, P).clone()
                e_raw, _ = _all_gather_var(dist, e_loc, P)
                e_all = self.ops.unique_edges(e_raw, gw.numel())
                cid_g, ismax_g, gs = self.ops.graph_edges(gw, gc, e_all, self.word_nt, self.distance,
                                                          self.method)
            elif self.partition_search and hasattr(self.ops, "pairs") and wpr == 1:
                # every rank searches its share of the pairs; the shares are all-gathered (tiny:
                # ~2 % of N pairs) and every rank builds the graph from the same complete list
                e_loc = self.ops.pairs(gw, self.word_nt, self.distance, r, P)
                ne = torch.tensor([e_loc.numel()], dtype=torch.int64, device=dev)
                nes = torch.empty(P, dtype=torch.int64, device=dev)
                _all_gather_flat(dist, nes, ne, P)
                nes = nes.cpu().tolist()
                e_max = max(max(nes), 1)
                pe = torch.zeros(e_max, dtype=torch.int64, device=dev)
                pe[:e_loc.numel()] = e_loc
                ae = torch.empty(P * e_max, dtype=torch.int64, device=dev)
                _all_gather_flat(dist, ae, pe, P)
                e_all = torch.cat([ae[q * e_max:q * e_max + nes[q]] for q in range(P)])
                cid_g, ismax_g, gs = self.ops.graph_edges(gw, gc, e_all, self.word_nt, self.distance,
                                                          self.method)
            else:
                cid_g, ismax_g, gs = self.ops.graph(gw, gc, self.word_nt, self.distance, self.method)
            for k in ("clusters", "edges", "nonsingle"):
                summ[k] = int(gs[k])
            for k, v in gs.items():
                if k.startswith("ms_"):
                    summ[k] = v
            dense = self.dense_return and P <= getattr(self.ops, "max_ranks_dense", 0)
            if dense:
                # ---- 6. owners emit dense per-shard result streams; one all-to-all; home ranks
                #         scatter them (both sides derive the split sizes from the value ranges)
                if use_dense:
                    packed = self.ops.map_dense(cid_g[goff:goff + u_local], ismax_g[goff:goff + u_local])
                else:
                    packed, send_counts = self.ops.owned_results(cid_g[goff:goff + u_local],
                                                                 ismax_g[goff:goff + u_local], shard_begin)
                perm, recv_counts = (self.ops.owner_perm(d_w.reshape(-1), d_f, ranges, self.word_nt) if wpr == 2
                                     else self.ops.owner_perm(d_w, d_f, ranges))
                recv = torch.empty(sum(recv_counts), dtype=torch.int32, device=dev)
                _all_to_all_v(dist, recv, packed, recv_counts, send_counts, P, r)
                self.ops.scatter(perm, recv, d_cid, d_keep)
                self.summary = summ
                return summ
            # ---- 6'. fallback: N-sized result arrays + reduce-scatter (sum) to the shards ----
            g_cid = torch.zeros(P * n_max, dtype=torch.int32, device=dev)
            g_keep = torch.zeros(P * n_max, dtype=torch.uint8, device=dev)
            self.ops.map(cid_g[goff:goff + u_local], ismax_g[goff:goff + u_local], g_cid, g_keep)
        else:
            g_cid = torch.zeros(P * n_max, dtype=torch.int32, device=dev)
            g_keep = torch.zeros(P * n_max, dtype=torch.uint8, device=dev)
        o_cid = torch.empty(n_max, dtype=torch.int32, device=dev)
        o_keep = torch.empty(n_max, dtype=torch.uint8, device=dev)
        _reduce_scatter_sum(dist, o_cid, g_cid, P, r)
        _reduce_scatter_sum(dist, o_keep, g_keep, P, r)
        d_cid.copy_(o_cid[:n_local])
        d_keep.copy_(o_keep[:n_local])
        self.summary = summ
        return summ


__all__ = ["ShardedDedup", "HipStageOps", "splitters_from_hist", "HumidError"]
