// kernels_xchg.hip.h -- the multi-GPU exchange pass with OWNER-LOCAL clustering (round 3)
// Part of libhumid_hip.so (see humid_hip.hip for the pipeline and the C ABI).  Device code for
// gfx950 only; included once, in this order, by humid_hip.hip.
//
// Rounds 1-2 all-gathered every neighbour pair and let every rank cluster the graph of ALL ranks' pairs:
// per-rank work that grows with the number of ranks.  Here a pair goes to the ONE rank that owns both of
// its ends (value ranges are contiguous slices of the walk order, so "owner" is a comparison with P
// boundaries); only pairs whose ends have different owners -- a substitution in the first log4(P)
// nucleotides: ~8 % at 8 ranks and d = 1 -- go to everybody, together with the interior pairs of the
// components they touch.  A rank then clusters its own components plus the (replicated) crossing ones in
// one compact graph over global unique indices; cluster ids stay closed-form (creators before a leaf =
// the lower ranks' creator totals + the owner's creators before it).
#ifndef HUMID_KERNELS_XCHG_HIP_H
#define HUMID_KERNELS_XCHG_HIP_H

#include "common.hip.h"
#include "kernels_graph.hip.h"
#include "kernels_cgraph.hip.h"
#include "kernels_map.hip.h"

// pair RECORD = {smaller global id << 32 | larger global id, count(smaller) | count(larger) << 32}
// append regions of records: the geometry of EdgeRegs (kernels_cgraph.hip.h) with 16-byte elements
struct RecRegs {
  ulonglong2 *e;                 // ER_REGIONS regions of cap_r records
  u32 cap_r;
  u32 *cur;                      // cur[r * ER_STRIDE]: records appended to (or wanted by) region r
  const ulonglong2 *far;         // region ER_REGIONS: a dense list (combinations that went through the large-bucket tiles)
  u32 n_far;
};
__device__ __forceinline__ u32 rr_count(const RecRegs &rr, u32 r) {
  if (r == ER_REGIONS) return rr.n_far;
  const u32 c = rr.cur[r * ER_STRIDE];
  return c < rr.cap_r ? c : rr.cap_r;
}
__device__ __forceinline__ const ulonglong2 *rr_at(const RecRegs &rr, u32 r, u32 k) {
  return r == ER_REGIONS ? rr.far + k : rr.e + (size_t)r * rr.cap_r + k;
}

// k_pairs_append (kernels_cgraph.hip.h) writing RECORDS: position p of the array the walked order refers to
// has the global id id_of ? id_of[p] : id_base + p and the count cnt_of[p].  PA_PPT positions per thread as there
// (a workgroup takes PA_PPT x 256 consecutive positions; the first two words of all a thread's walks are requested
// together, and a launch has a quarter of the workgroups).
template <bool PASS0, class WT>
__global__ void __launch_bounds__(256)
k_pairs_records(const WT *__restrict__ W, const u32 *__restrict__ V, u32 n, WT mask, EarlierMasksT<WT> em, u32 cb,
                u32 distance, u32 walk_max, const u32 *__restrict__ id_of, u32 id_base, const u32 *__restrict__ cnt_of,
                RecRegs rr, ull *big, u32 *overflow) {
  HUMID_GUARD_LAST_VGPR();
  __shared__ u32 lds[8];
  __shared__ u32 s_base;
  const u32 i0 = blockIdx.x * (PA_PPT * 256u) + threadIdx.x;
  u32 found[PA_PPT], first_off[PA_PPT], jend[PA_PPT];
  WT wi[PA_PPT], w1[PA_PPT];
#pragma unroll
  for (u32 q = 0; q < PA_PPT; q++) {
    const u32 i = i0 + q * 256u;
    found[q] = 0; first_off[q] = 0; jend[q] = 0;
    if (i < n) {
      wi[q] = W[i];
      w1[q] = W[i + 1 < n ? i + 1 : i];                // requested together with W[i]: most walks end at this word
    }
  }
  u32 total_found = 0;
#pragma unroll
  for (u32 q = 0; q < PA_PPT; q++) {
    const u32 i = i0 + q * 256u;
    if (i >= n) continue;
    jend[q] = (walk_max && n - i > walk_max + 1) ? i + walk_max + 1 : n;
    u32 j = i + 1;
    for (; j < jend[q]; j++) {
      const WT x = w_xor(wi[q], j == i + 1 ? w1[q] : W[j]);
      if (w_hits(x, mask)) break;
      if (w_mismatch(x) > distance) continue;
      bool first = true;
#pragma unroll
      for (u32 t = 0; t < MAX_COMBOS; t++)
        first = first && !(t < cb && !w_hits(x, em.m[t]));
      if (!first) continue;
      if (!found[q]) first_off[q] = j - i;
      found[q]++;
    }
    if (big && j == jend[q] && jend[q] < n && !w_hits(w_xor(wi[q], W[jend[q]]), mask)) atomicOr(big, 1ull << cb);
    total_found += found[q];
  }
  const u32 lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  u32 incl = total_found;
  incl = wave_incl_scan(incl);
  if (lane == 63) lds[wv] = incl;
  __syncthreads();
  u32 before = 0, total = 0;
#pragma unroll
  for (u32 k = 0; k < 4; k++) { if (k < wv) before += lds[k]; total += lds[k]; }
  if (total == 0) return;                              // (uniform)
  const u32 region = blockIdx.x % ER_REGIONS;
  if (threadIdx.x == 0) s_base = atomicAdd(&rr.cur[region * ER_STRIDE], total);
  __syncthreads();
  if (!total_found) return;
  u32 at = s_base + before + incl - total_found;
  if (at + total_found > rr.cap_r) { *overflow = 1; if (at >= rr.cap_r) return; }
  ulonglong2 *out = rr.e + (size_t)region * rr.cap_r;
#pragma unroll
  for (u32 q = 0; q < PA_PPT; q++) {
    if (!found[q]) continue;
    const u32 i = i0 + q * 256u;
    const u32 pi = PASS0 ? i : V[i];
    const u32 ida = id_of ? id_of[pi] : id_base + pi, ca = cnt_of[pi];
    auto emit = [&](u32 j) {
      if (at >= rr.cap_r) return;
      const u32 pj = PASS0 ? j : V[j];
      const u32 idb = id_of ? id_of[pj] : id_base + pj, cb2 = cnt_of[pj];
      out[at++] = ida < idb ? make_ulonglong2(((u64)ida << 32) | idb, (u64)ca | ((u64)cb2 << 32))
                            : make_ulonglong2(((u64)idb << 32) | ida, (u64)cb2 | ((u64)ca << 32));
    };
    if (found[q] == 1) { emit(i + first_off[q]); continue; }
    for (u32 j = i + first_off[q]; j < jend[q]; j++) {
      const WT x = w_xor(wi[q], W[j]);
      if (w_hits(x, mask)) break;
      if (w_mismatch(x) > distance) continue;
      bool first = true;
#pragma unroll
      for (u32 t = 0; t < MAX_COMBOS; t++)
        first = first && !(t < cb && !w_hits(x, em.m[t]));
      if (first) emit(j);
    }
  }
}

// the fullest region's cursor -> out[0], the records in all regions (cursors cut to the room) -> out[1]
static __global__ void k_rec_regions_max(RecRegs rr, u32 *__restrict__ out) {
  HUMID_GUARD_LAST_VGPR();
  u32 c = rr.cur[threadIdx.x * ER_STRIDE];
  u32 t = c < rr.cap_r ? c : rr.cap_r;
#pragma unroll
  for (u32 d = 32; d >= 1; d >>= 1) { const u32 y = __shfl_xor(c, d); c = y > c ? y : c; t += __shfl_xor(t, d); }
  if (threadIdx.x == 0) { out[0] = c; out[1] = t; }
}

// ---- where a record goes ---------------------------------------------------------------------------
// id_begin[q] = first global unique index of rank q (ascending; id_begin[n_ranks] = all unique words).
// Both ends with one owner: that owner (an "interior" pair); else destination n_ranks: everybody.
struct IdRanges {
  u32 b[MAX_RANKS + 1];
};
__device__ __forceinline__ u32 id_owner(const IdRanges &rg, u32 n_ranks, u32 id) {
  u32 o = 0;
#pragma unroll
  for (u32 q = 1; q < MAX_RANKS; q++) o += (q < n_ranks && id >= rg.b[q]) ? 1u : 0u;
  return o;
}
__device__ __forceinline__ u32 rec_dest(const IdRanges &rg, u32 n_ranks, u64 e) {
  const u32 oa = id_owner(rg, n_ranks, (u32)(e >> 32)), ob = id_owner(rg, n_ranks, (u32)e);
  return oa == ob ? oa : n_ranks;
}
// counts per destination (n_ranks + 1 of them) over all regions; grid: x over a region, y = region
static __global__ void __launch_bounds__(256)
k_rec_dest_count(RecRegs rr, IdRanges rg, u32 n_ranks, u32 *__restrict__ counts) {
  HUMID_GUARD_LAST_VGPR();
  __shared__ u32 h[MAX_RANKS + 1];
  if (threadIdx.x <= MAX_RANKS) h[threadIdx.x] = 0;
  __syncthreads();
  const u32 r = blockIdx.y, n_r = rr_count(rr, r);
  for (u32 k = blockIdx.x * blockDim.x + threadIdx.x; k < n_r; k += gridDim.x * blockDim.x)
    atomicAdd(&h[rec_dest(rg, n_ranks, rr_at(rr, r, k)->x)], 1u);
  __syncthreads();
  if (threadIdx.x <= n_ranks && h[threadIdx.x]) atomicAdd(&counts[threadIdx.x], h[threadIdx.x]);
}
// the records in destination-major order: base[d] = first slot of destination d (host: prefix of the counts),
// cursor[d] zeroed; one global atomic per workgroup and destination
static __global__ void __launch_bounds__(256)
k_rec_dest_scatter(RecRegs rr, IdRanges rg, u32 n_ranks, IdRanges base, u32 *cursor, ulonglong2 *__restrict__ out) {
  HUMID_GUARD_LAST_VGPR();
  __shared__ u32 h[MAX_RANKS + 1], hb[MAX_RANKS + 1];
  const u32 r = blockIdx.y, n_r = rr_count(rr, r);
  for (u32 k0 = blockIdx.x * blockDim.x; k0 < n_r; k0 += gridDim.x * blockDim.x) {   // (whole workgroups stay in the loop)
    if (threadIdx.x <= MAX_RANKS) h[threadIdx.x] = 0;
    __syncthreads();
    const u32 k = k0 + threadIdx.x;
    ulonglong2 rec;
    u32 d = 0, rank = 0;
    const bool ok = k < n_r;
    if (ok) {
      rec = *rr_at(rr, r, k);
      d = rec_dest(rg, n_ranks, rec.x);
      rank = atomicAdd(&h[d], 1u);
    }
    __syncthreads();
    if (threadIdx.x <= n_ranks) hb[threadIdx.x] = h[threadIdx.x] ? atomicAdd(&cursor[threadIdx.x], h[threadIdx.x]) : 0u;
    __syncthreads();
    if (ok) out[base.b[d] + hb[d] + rank] = rec;
    __syncthreads();
  }
}

// ---- which of a rank's interior pairs belong to components a crossing pair touches ----------------
// forest over this rank's own leaves (local index = global id - id0) from its interior pairs
static __global__ void __launch_bounds__(256)
k_union_records(const ulonglong2 *__restrict__ recs, u32 n, u32 id0, u32 n_local, u32 *parent, bool join_by_count, u32 *bad) {
  HUMID_GUARD_LAST_VGPR();
  const u32 k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const ulonglong2 r = recs[k];
  const u32 a = (u32)(r.x >> 32) - id0, b = (u32)r.x - id0;
  if (a >= n_local || b >= n_local) { *bad = 1; return; }
  const u32 ca = (u32)r.y, cb = (u32)(r.y >> 32);
  if (!join_by_count || at_least_double(ca, cb) || at_least_double(cb, ca)) uf_union(parent, a, b);
}
// every end of a crossing pair that this rank owns flags the root of its component (pairs the clustering
// never crosses -- counts within a factor of two, directional method -- flag nothing)
static __global__ void __launch_bounds__(256)
k_flag_crossing(const ulonglong2 *__restrict__ recs, u32 n, u32 id0, u32 n_local, const u32 *parent, u8 *__restrict__ flag,
                bool join_by_count) {
  HUMID_GUARD_LAST_VGPR();
  const u32 k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const ulonglong2 r = recs[k];
  const u32 ca = (u32)r.y, cb = (u32)(r.y >> 32);
  if (join_by_count && !at_least_double(ca, cb) && !at_least_double(cb, ca)) return;
  const u32 a = (u32)(r.x >> 32) - id0, b = (u32)r.x - id0;
  if (a < n_local) flag[uf_find(parent, a)] = 1;
  if (b < n_local) flag[uf_find(parent, b)] = 1;
}
// the interior pairs of flagged components, appended to `out` (one global atomic per workgroup); COUNT: only counted
template <bool COUNT>
__global__ void __launch_bounds__(256)
k_select_flagged(const ulonglong2 *__restrict__ recs, u32 n, u32 id0, const u32 *parent, const u8 *__restrict__ flag,
                 u32 *cursor, ulonglong2 *__restrict__ out) {
  HUMID_GUARD_LAST_VGPR();
  __shared__ u32 lds[8];
  __shared__ u32 s_base;
  for (u32 k0 = blockIdx.x * blockDim.x; k0 < n; k0 += gridDim.x * blockDim.x) {
    const u32 k = k0 + threadIdx.x;
    ulonglong2 r;
    bool sel = false;
    if (k < n) {
      r = recs[k];
      sel = flag[uf_find(parent, (u32)(r.x >> 32) - id0)] != 0;
    }
    u32 tot;
    const u32 rank = block_rank(sel, lds, &tot);
    if (threadIdx.x == 0) s_base = tot ? atomicAdd(cursor, tot) : 0u;
    __syncthreads();
    if (!COUNT && sel) out[s_base + rank] = r;
    __syncthreads();
  }
}

// ---- several record arrays as ONE source of the compact graph ------------------------------------
#define REC_SEGS 4
struct RecSegs {
  const ulonglong2 *p[REC_SEGS];
  u32 n[REC_SEGS];
  u32 first[REC_SEGS + 1];       // first[s] = records of the segments before s (positions in the compact pair list)
};
static __global__ void __launch_bounds__(256)
k_mark_segs(RecSegs sg, u32 n_ids, u32 *bits, u32 *bad) {
  HUMID_GUARD_LAST_VGPR();
  const u32 s = blockIdx.y;
  const u32 n = sg.n[s];
  for (u32 k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
    const u64 e = sg.p[s][k].x;
    const u32 a = (u32)(e >> 32), b = (u32)e;
    if (a >= n_ids || b >= n_ids || a == b) { *bad = 1; continue; }
    atomicOr(&bits[a >> 5], 1u << (a & 31));
    atomicOr(&bits[b >> 5], 1u << (b & 31));
  }
}
static __global__ void __launch_bounds__(256)
k_segs_relabel(RecSegs sg, u32 n_ids, BitRank br, u64 *__restrict__ cpairs, u32 *__restrict__ ncnt, u32 *deg, u32 *parent,
               bool join_by_count) {
  HUMID_GUARD_LAST_VGPR();
  const u32 s = blockIdx.y;
  const u32 n = sg.n[s];
  for (u32 k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
    const ulonglong2 r = sg.p[s][k];
    const u32 ia = (u32)(r.x >> 32), ib = (u32)r.x;
    if (ia >= n_ids || ib >= n_ids || ia == ib) { cpairs[sg.first[s] + k] = 0; continue; }   // (reported by k_mark_segs)
    const u32 a = br_rank(br, ia), b = br_rank(br, ib);
    const u32 ca = (u32)r.y, cb = (u32)(r.y >> 32);
    cpairs[sg.first[s] + k] = ((u64)a << 32) | b;
    ncnt[a] = ca;
    ncnt[b] = cb;
    atomicAdd(&deg[a], 1u);
    atomicAdd(&deg[b], 1u);
    if (!join_by_count || at_least_double(ca, cb) || at_least_double(cb, ca)) uf_union(parent, a, b);
  }
}

// ---- cluster ids ------------------------------------------------------------------------------------
// A CROSSING cluster: a cluster of a component that a crossing pair touches.  Its creator may belong to
// another rank, whose creator count before it this rank cannot know -- the owner computes the id and the
// ranks exchange them.  xroot[root] = 1 for the components of the replicated part: every end of a crossing
// pair (segment `seg`) flags the root of its compact node.
// (Only pairs the clustering can cross flag anything, exactly as in k_flag_crossing at the owners: the set of
// crossing clusters must come out the same on every rank, and it is the components of such pairs -- with all
// their pairs -- that every rank holds.)
static __global__ void __launch_bounds__(256)
k_flag_xroots(RecSegs sg, u32 seg, u32 n_ids, BitRank br, const u32 *__restrict__ parent, u8 *__restrict__ xroot, bool join_by_count) {
  HUMID_GUARD_LAST_VGPR();
  const u32 n = sg.n[seg];
  for (u32 k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
    const ulonglong2 r = sg.p[seg][k];
    const u64 e = r.x;
    const u32 ia = (u32)(e >> 32), ib = (u32)e;
    if (ia >= n_ids || ib >= n_ids || ia == ib) continue;
    const u32 ca = (u32)r.y, cb = (u32)(r.y >> 32);
    if (join_by_count && !at_least_double(ca, cb) && !at_least_double(cb, ca)) continue;
    xroot[parent[br_rank(br, ia)]] = 1;                 // parent: flattened by k_comp_stats
    xroot[parent[br_rank(br, ib)]] = 1;
  }
}
// the creators of the crossing clusters, as a bitmap over global ids
static __global__ void __launch_bounds__(256)
k_xcreator_bits(const u32 *__restrict__ cl_of, const u32 *__restrict__ parent, const u8 *__restrict__ xroot,
                const u32 *__restrict__ nodes, u32 m, u32 *bits) {
  HUMID_GUARD_LAST_VGPR();
  const u32 c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= m || cl_of[c] != c + 1 || !xroot[parent[c]]) return;
  const u32 u = nodes[c];
  atomicOr(&bits[u >> 5], 1u << (u & 31));
}
// cluster id of a creator this rank owns: 1 + the creators of the lower ranks + its own creators before it
__device__ __forceinline__ u32 own_creator_id(u32 g, u32 id0, u32 creators_before_rank, const BitRank &noncreator, u32 nc_before_rank) {
  return 1u + creators_before_rank + (g - id0) - (br_rank(noncreator, g) - nc_before_rank);
}
// xcid[k] = id of the k-th crossing creator if this rank owns it, else 0 (the ranks' arrays are all-gathered
// and merged by maximum)
static __global__ void __launch_bounds__(256)
k_xcreator_ids(BitRank xc, u32 n_ids_words, u32 id0, u32 n_local, u32 creators_before_rank, BitRank noncreator,
               u32 *__restrict__ xcid) {
  HUMID_GUARD_LAST_VGPR();
  const u32 w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= n_ids_words) return;
  u32 x = xc.bits[w];
  if (!x) return;
  u32 k = br_rank(xc, w << 5);
  const u32 nc0 = br_rank(noncreator, id0);
  while (x) {
    const u32 bit = (u32)__ffs((int)x) - 1u;
    x &= x - 1u;
    const u32 g = (w << 5) | bit;
    xcid[k] = (g >= id0 && g - id0 < n_local) ? own_creator_id(g, id0, creators_before_rank, noncreator, nc0) : 0u;
    k++;
  }
}
static __global__ void k_max_rows(const u32 *__restrict__ rows, u32 n_rows, u32 n, u32 *__restrict__ out) {
  HUMID_GUARD_LAST_VGPR();
  const u32 k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  u32 m = 0;
  for (u32 q = 0; q < n_rows; q++) { const u32 v = rows[(size_t)q * n + k]; m = v > m ? v : m; }
  out[k] = m;
}
// cluster id and maxLeaf flag of this rank's own unique words (local index u, global id0 + u), and their
// degree (for neigh.dat).  xcid_all: the merged ids of the crossing creators (null: no crossing pair anywhere).
static __global__ void __launch_bounds__(256)
k_own_results(BitRank in_graph, BitRank noncreator, BitRank xc, const u32 *__restrict__ xcid_all, const u32 *__restrict__ nodes,
              const u32 *__restrict__ cl_of, const u32 *__restrict__ maxleaf, const u32 *__restrict__ deg, u32 id0, u32 n_local,
              u32 creators_before_rank, u32 *__restrict__ l_cid, u8 *__restrict__ l_ismax, u32 *__restrict__ l_deg,
              const u32 *__restrict__ s_first, const u32 *__restrict__ s_slot, u64 *__restrict__ slot_out, bool whole_graph = false) {
  HUMID_GUARD_LAST_VGPR();
  const u32 u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= n_local) return;
  const u32 me = id0 + u;
  u32 g = me, dg = 0;
  bool mx = true;
  if (br_test(in_graph, me)) {
    const u32 c = br_rank(in_graph, me);
    const u32 cc = cl_of[c] - 1u;
    g = nodes[cc];
    mx = maxleaf[cc] == c;
    dg = deg[c];
  }
  u32 id;
  // whole_graph: this rank holds the graph of ALL ranks (the edit-distance road): every creator's id is closed-form
  if (whole_graph) id = 1u + g - br_rank(noncreator, g);
  else if (xcid_all && br_test(xc, g)) id = xcid_all[br_rank(xc, g)];
  else id = own_creator_id(g, id0, creators_before_rank, noncreator, br_rank(noncreator, id0));
  l_cid[u] = id;
  l_ismax[u] = mx ? 1 : 0;
  if (l_deg) l_deg[u] = dg;
  if (slot_out) slot_out[s_slot[u]] = ((u64)(mx ? s_first[u] : NONE32) << 32) | id;   // (k_slot_results of the result return, done here)
}
// non-creators and nodes among this rank's own leaves [id0, id0 + n_local): out[0], out[1]
static __global__ void k_own_totals(BitRank noncreator, BitRank in_graph, u32 id0, u32 n_local, u32 *__restrict__ out) {
  HUMID_GUARD_LAST_VGPR();
  if (threadIdx.x == 0) {
    out[0] = n_local ? br_rank(noncreator, id0 + n_local - 1) + (br_test(noncreator, id0 + n_local - 1) ? 1u : 0u) - br_rank(noncreator, id0) : 0u;
    out[1] = n_local ? br_rank(in_graph, id0 + n_local - 1) + (br_test(in_graph, id0 + n_local - 1) ? 1u : 0u) - br_rank(in_graph, id0) : 0u;
  }
}

#endif  // HUMID_KERNELS_XCHG_HIP_H
