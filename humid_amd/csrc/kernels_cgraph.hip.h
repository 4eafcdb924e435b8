// kernels_cgraph.hip.h -- the neighbour graph over the leaves that HAVE neighbours only (round 3)
// Part of libhumid_hip.so (see humid_hip.hip for the pipeline and the C ABI).  Device code for
// gfx950 only; included once, in this order, by humid_hip.hip.
//
// On UMI data about 85 % of the unique words have no neighbour at all.  Rounds 1-2 ran every graph and
// cluster pass over all U unique words (degrees, union-find parents, component sizes, cluster arrays,
// creator flags: ~370 B of HBM traffic per unique word).  Here the pair search APPENDS the pairs it
// finds (64 append regions, one global atomic per workgroup that found something) and marks both ends
// in a bitmap of U bits; a rank structure over the bitmap (set bits before every 256th leaf) turns a
// walk index into a COMPACT node index in two cached loads.  Degrees, CSR rows, union-find, the
// findClusters loop and its outputs then live on the M compact nodes (M ~ 0.15 U); leaves outside the
// graph are their own cluster and maxLeaf (src/humid.cc:176-189 with an empty neighbour list), and the
// cluster id of any leaf is closed-form: 1 + the creator's walk index - the graph nodes before it that
// did NOT create a cluster (a second bitmap + rank structure).  Walk order is preserved by the
// compaction (ranks are monotone), so "members ascending", "lists ascending" and "ids in order of the
// creating leaf" mean the same thing on compact indices.
#ifndef HUMID_KERNELS_CGRAPH_HIP_H
#define HUMID_KERNELS_CGRAPH_HIP_H

#include "common.hip.h"
#include "kernels_graph.hip.h"
#include "kernels_cluster.hip.h"

// ---- append regions of the pair list ------------------------------------------------------------
#define ER_REGIONS 64u
#define ER_STRIDE 32u            // u32 between two cursors: 128 B, one cache line each
struct EdgeRegs {
  u64 *e;                        // ER_REGIONS regions of cap_r pairs (smaller walk index << 32 | larger)
  u32 cap_r;
  u32 *cur;                      // cur[r * ER_STRIDE]: pairs appended to (or wanted by) region r
  const u64 *far;                // region ER_REGIONS: the pairs of buckets beyond the bounded walk (k_pairs_tiles)
  u32 n_far;
};
__device__ __forceinline__ u32 er_count(const EdgeRegs &er, u32 r) {
  if (r == ER_REGIONS) return er.n_far;
  const u32 c = er.cur[r * ER_STRIDE];
  return c < er.cap_r ? c : er.cap_r;
}
__device__ __forceinline__ u64 *er_at(const EdgeRegs &er, u32 r, u32 k) {
  return r == ER_REGIONS ? (u64 *)er.far + k : er.e + (size_t)r * er.cap_r + k;
}

// ---- rank structure over a bitmap ------------------------------------------------------------------
// bits: one bit per leaf (walk index), padded to a multiple of 8 words; blk[b] = set bits before leaf
// 256 b (an exclusive scan of the per-block popcounts), blk[n_blk] = total
struct BitRank {
  const u32 *bits;
  const u32 *blk;
};
__device__ __forceinline__ bool br_test(const BitRank &br, u32 u) { return (br.bits[u >> 5] >> (u & 31)) & 1u; }
// set bits before leaf u (exclusive)
__device__ __forceinline__ u32 br_rank(const BitRank &br, u32 u) {
  const u32 w = u >> 5, b0 = w & ~7u;
  const uint4 *p = (const uint4 *)(br.bits + b0);
  const uint4 x = p[0], y = p[1];
  const u32 v[8] = {x.x, x.y, x.z, x.w, y.x, y.y, y.z, y.w};
  u32 r = br.blk[u >> 8];
  const u32 k = w - b0;
#pragma unroll
  for (u32 q = 0; q < 8; q++) {
    const u32 m = q < k ? 0xffffffffu : (q == k ? ((1u << (u & 31)) - 1u) : 0u);
    r += (u32)__popc(v[q] & m);
  }
  return r;
}

// the same as the input of a scan (prims.hip.h): item b = set bits of block b, item n_blk = 0 -- the rank
// blocks come out of ONE launch, without the array of counts in between
struct BitsBlockIn {
  const u32 *bits;
  u32 n_blk;
  __device__ __forceinline__ u32 operator()(u64 b) const {
    if (b >= n_blk) return 0u;
    const uint4 *p = (const uint4 *)(bits + 8 * (size_t)b);
    const uint4 x = p[0], y = p[1];
    return (u32)(__popc(x.x) + __popc(x.y) + __popc(x.z) + __popc(x.w) + __popc(y.x) + __popc(y.y) + __popc(y.z) + __popc(y.w));
  }
};

// set bits of every block of 256 leaves (8 words); cnt[n_blk] = 0 is the scan's sentinel
static __global__ void __launch_bounds__(256)
k_bits_blocks(const u32 *__restrict__ bits, u32 n_blk, u32 *__restrict__ cnt) {
  HUMID_GUARD_LAST_VGPR();
  const u32 b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b > n_blk) return;
  if (b == n_blk) { cnt[b] = 0; return; }
  const uint4 *p = (const uint4 *)(bits + 8 * (size_t)b);
  const uint4 x = p[0], y = p[1];
  cnt[b] = (u32)(__popc(x.x) + __popc(x.y) + __popc(x.z) + __popc(x.w) + __popc(y.x) + __popc(y.y) + __popc(y.z) + __popc(y.w));
}

// zeroes a handful of small scratch arrays in ONE launch (bitmaps, cursors, counters): what used to be
// one fillBuffer launch each
#define ZERO_MAX 8
struct ZeroList {
  u32 *p[ZERO_MAX];
  u32 n[ZERO_MAX];               // in u32
};
static __global__ void __launch_bounds__(256)
k_zero_many(ZeroList z) {
  HUMID_GUARD_LAST_VGPR();
  const u32 stride = gridDim.x * blockDim.x;
#pragma unroll
  for (u32 q = 0; q < ZERO_MAX; q++) {
    u32 *p = z.p[q];
    const u32 n = z.n[q];
    for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = 0;
  }
}

// ---- pair search that appends ----------------------------------------------------------------------
// k_pairs (kernels_graph.hip.h) in one pass: a position walks its bucket once, counting its pairs and
// remembering the first; the workgroup reserves room for all its pairs with ONE atomic on the cursor of
// its region (blockIdx % 64: the ~10^4 workgroups of a launch spread their atomics over 64 lines); a
// position with one pair -- nearly all that have any -- writes it from memory, the others walk again.
// Both ends are marked in `bits`.  A region that is full keeps counting (the cursor says how much room
// the search wants) and sets *overflow.
// PA_PPT positions per thread (a workgroup takes PA_PPT x 256 consecutive positions, thread t the positions t, t + 256,
// ...): the first two words of all its walks are requested together, and a launch has a quarter of the workgroups --
// with one position per thread the 10^4 workgroups of ~3 us each were bound by how fast workgroups can be started.
#define PA_PPT 4u
template <bool PASS0, class WT>
__global__ void __launch_bounds__(256)
k_pairs_append(const WT *__restrict__ W, const u32 *__restrict__ V, u32 n, WT mask, EarlierMasksT<WT> em, u32 cb,
               u32 distance, u32 walk_max, EdgeRegs er, u32 *bits, ull *big, u32 *overflow,
               const u32 *__restrict__ n_valid = nullptr) {
  HUMID_GUARD_LAST_VGPR();
  __shared__ u32 lds[8];
  __shared__ u32 s_base;
  PH_DECL;
  PH(0);
  // n_valid: the walked order came from a padded grouping; had a coarse bin been full, words were dropped and the
  // order ends at *n_valid (what lies behind it was never written; the caller discards this search)
  if (n_valid && *n_valid < n) n = *n_valid;
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
      if (w_hits(x, mask)) break;                    // left the bucket
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
  PH(1);
  // room for the workgroup's pairs: exclusive position of this thread's, one atomic per workgroup
  const u32 lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  u32 incl = total_found;
  incl = wave_incl_scan(incl);
  if (lane == 63) lds[wv] = incl;
  __syncthreads();
  u32 before = 0, total = 0;
#pragma unroll
  for (u32 k = 0; k < 4; k++) { if (k < wv) before += lds[k]; total += lds[k]; }
  PH(2);
  if (total == 0) { PH(3); PH(4); PH_END(5, 4, (blockIdx.x & 15u) == 5u); return; }                              // (uniform)
  const u32 region = blockIdx.x % ER_REGIONS;
  if (threadIdx.x == 0) s_base = atomicAdd(&er.cur[region * ER_STRIDE], total);
  __syncthreads();
  PH(3);
  if (threadIdx.x == 0) { PH(4); PH_END(5, 4, (blockIdx.x & 15u) == 5u); }   // 1 walks | 2 block scan | 3 region cursor | (4 -)
  if (!total_found) return;
  u32 at = s_base + before + incl - total_found;
  if (at + total_found > er.cap_r) { *overflow = 1; if (at >= er.cap_r) return; }
  u64 *out = er.e + (size_t)region * er.cap_r;
#pragma unroll
  for (u32 q = 0; q < PA_PPT; q++) {
    if (!found[q]) continue;
    const u32 i = i0 + q * 256u;
    const u32 ri = PASS0 ? i : V[i];
    auto emit = [&](u32 j) {
      if (at >= er.cap_r) return;
      const u32 rj = PASS0 ? j : V[j];
      out[at++] = ri < rj ? (((u64)ri << 32) | rj) : (((u64)rj << 32) | ri);
      atomicOr(&bits[rj >> 5], 1u << (rj & 31));
    };
    atomicOr(&bits[ri >> 5], 1u << (ri & 31));
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

// both ends of a plain pair list marked in the bitmap (pairs that were not found by k_pairs_append: the
// large-bucket tiles, the edit-distance joins)
static __global__ void __launch_bounds__(256)
k_mark_pairs(const u64 *__restrict__ pairs, u32 n_pairs, u32 n_leaves, u32 *bits, u32 *bad) {
  HUMID_GUARD_LAST_VGPR();
  const u32 k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_pairs) return;
  const u64 e = pairs[k];
  const u32 a = (u32)(e >> 32), b = (u32)e;
  if (a >= n_leaves || b >= n_leaves || a == b) { *bad = 1; return; }
  atomicOr(&bits[a >> 5], 1u << (a & 31));
  atomicOr(&bits[b >> 5], 1u << (b & 31));
}

// ---- compact nodes -----------------------------------------------------------------------------------
// one thread per bitmap word: the marked leaves of the word become nodes[c], c ascending with the walk
// index; their counts are gathered and the per-node arrays of the graph stage initialised
static __global__ void __launch_bounds__(256)
k_nodes_init(BitRank br, u32 n_words, const u32 *__restrict__ s_cnt, u32 *__restrict__ nodes, u32 *__restrict__ ncnt,
             u32 *__restrict__ deg, u32 *__restrict__ parent, u32 *__restrict__ csize, u32 *__restrict__ cur, u32 n_blk) {
  HUMID_GUARD_LAST_VGPR();
  const u32 w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w == 0) deg[br.blk[n_blk]] = 0;                  // deg[M]: the sentinel of the offsets' scan
  if (w >= n_words) return;
  u32 x = br.bits[w];
  if (!x) return;
  u32 c = br_rank(br, w << 5);
  while (x) {
    const u32 bit = (u32)__ffs((int)x) - 1u;
    x &= x - 1u;
    const u32 u = (w << 5) | bit;
    nodes[c] = u;
    if (s_cnt) ncnt[c] = s_cnt[u];                     // (null: the counts come with pair records, k_records_relabel)
    deg[c] = 0; parent[c] = c; csize[c] = 0; cur[c] = 0;
    c++;
  }
}

// scan input: deg[i] for i < *M, 0 beyond (the launch is sized by a host-side bound on M)
struct DegIn {
  const u32 *deg;
  const u32 *m_dev;
  __device__ __forceinline__ u32 operator()(u64 i) const { return i < (u64)*m_dev ? deg[i] : 0u; }
};

// pairs in walk indices -> pairs in compact indices (in place), degrees, component forest
// grid: x over the positions of a region, y = region (ER_REGIONS + 1 of them)
static __global__ void __launch_bounds__(256)
k_pairs_relabel(EdgeRegs er, BitRank br, const u32 *__restrict__ ncnt, u32 *deg, u32 *parent, bool join_by_count) {
  HUMID_GUARD_LAST_VGPR();
  const u32 r = blockIdx.y;
  const u32 n_r = er_count(er, r);
  for (u32 k = blockIdx.x * blockDim.x + threadIdx.x; k < n_r; k += gridDim.x * blockDim.x) {
    u64 *slot = er_at(er, r, k);
    const u64 e = *slot;
    const u32 a = br_rank(br, (u32)(e >> 32)), b = br_rank(br, (u32)e);
    *slot = ((u64)a << 32) | b;
    atomicAdd(&deg[a], 1u);
    atomicAdd(&deg[b], 1u);
    if (joins_for_clustering(join_by_count ? ncnt : (const u32 *)nullptr, a, b)) uf_union(parent, a, b);
  }
}

// k_pairs_fill and k_comp_stats in ONE launch (the first gx * (ER_REGIONS + 1) workgroups fill the CSR rows, the rest
// flatten the forest and size the components): neither reads what the other writes, and each alone is one round of
// workgroups whose time is a chain of memory round trips -- side by side the two chains overlap
static __global__ void __launch_bounds__(256)
k_fill_and_stats(EdgeRegs er, const u32 *__restrict__ off, u32 *cur, u32 *__restrict__ idx, u32 *__restrict__ regions_max, u32 gx,
                 const u32 *__restrict__ deg, u32 *P, u32 n, u32 *csize, const u32 *__restrict__ n_dev, u32 r_first) {
  HUMID_GUARD_LAST_VGPR();
  // r_first: the first region this launch covers -- 0, or ER_REGIONS when the pairs are one dense list (records of the
  // exchange pass, given edges: only the `far` region holds anything; 64 x gx workgroups that find their region empty
  // cost 30 us of a launch there)
  const u32 nf = gx * (ER_REGIONS + 1 - r_first);
  if (blockIdx.x >= nf) {
    const u32 u = (blockIdx.x - nf) * blockDim.x + threadIdx.x;
    if (n_dev && *n_dev < n) n = *n_dev;
    if (u >= n || deg[u] == 0) return;
    const u32 root = uf_find(P, u);
    P[u] = root;
    atomicAdd(&csize[root], 1u);
    return;
  }
  const u32 bx = blockIdx.x % gx, r = r_first + blockIdx.x / gx;
  if (regions_max && blockIdx.x == 0 && threadIdx.x < 64) {
    // the fullest region's cursor (what the search wanted of ONE region: all regions have the same room)
    u32 c = er.cur[threadIdx.x * ER_STRIDE];
#pragma unroll
    for (u32 d = 32; d >= 1; d >>= 1) { const u32 y = __shfl_xor(c, d); c = y > c ? y : c; }
    if (threadIdx.x == 0) regions_max[0] = c;
  }
  const u32 n_r = er_count(er, r);
  for (u32 k = bx * blockDim.x + threadIdx.x; k < n_r; k += gx * blockDim.x) {
    const u64 e = *er_at(er, r, k);
    const u32 a = (u32)(e >> 32), b = (u32)e;
    idx[off[a] + atomicAdd(&cur[a], 1u)] = b;
    idx[off[b] + atomicAdd(&cur[b], 1u)] = a;
  }
}

// CSR rows through per-row cursors (put in ascending order afterwards by k_sort_lists)
static __global__ void __launch_bounds__(256)
k_pairs_fill(EdgeRegs er, const u32 *__restrict__ off, u32 *cur, u32 *__restrict__ idx, u32 *__restrict__ regions_max) {
  HUMID_GUARD_LAST_VGPR();
  if (regions_max && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 64) {
    // the fullest region's cursor (what the search wanted of ONE region: all regions have the same room)
    u32 c = er.cur[threadIdx.x * ER_STRIDE];
#pragma unroll
    for (u32 d = 32; d >= 1; d >>= 1) { const u32 y = __shfl_xor(c, d); c = y > c ? y : c; }
    if (threadIdx.x == 0) regions_max[0] = c;
  }
  const u32 r = blockIdx.y;
  const u32 n_r = er_count(er, r);
  for (u32 k = blockIdx.x * blockDim.x + threadIdx.x; k < n_r; k += gridDim.x * blockDim.x) {
    const u64 e = *er_at(er, r, k);
    const u32 a = (u32)(e >> 32), b = (u32)e;
    idx[off[a] + atomicAdd(&cur[a], 1u)] = b;
    idx[off[b] + atomicAdd(&cur[b], 1u)] = a;
  }
}

// ---- pair RECORDS as the source (multi-GPU: pairs in global unique indices with both ends' counts) ----
// record = {smaller id << 32 | larger id, count(smaller) | count(larger) << 32}
static __global__ void __launch_bounds__(256)
k_mark_records(const ulonglong2 *__restrict__ recs, u32 n_recs, u32 n_ids, u32 *bits, u32 *bad) {
  HUMID_GUARD_LAST_VGPR();
  const u32 k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_recs) return;
  const u64 e = recs[k].x;
  const u32 a = (u32)(e >> 32), b = (u32)e;
  if (a >= n_ids || b >= n_ids || a == b) { *bad = 1; return; }
  atomicOr(&bits[a >> 5], 1u << (a & 31));
  atomicOr(&bits[b >> 5], 1u << (b & 31));
}
// records -> compact pairs (a dense list: the `far` region of an EdgeRegs), the nodes' counts (all records
// of a node carry the same count: equal values race freely), degrees, component forest
static __global__ void __launch_bounds__(256)
k_records_relabel(const ulonglong2 *__restrict__ recs, u32 n_recs, u32 n_ids, BitRank br, u64 *__restrict__ cpairs,
                  u32 *__restrict__ ncnt, u32 *deg, u32 *parent, bool join_by_count) {
  HUMID_GUARD_LAST_VGPR();
  const u32 k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_recs) return;
  const ulonglong2 r = recs[k];
  const u32 ia = (u32)(r.x >> 32), ib = (u32)r.x;
  if (ia >= n_ids || ib >= n_ids || ia == ib) { cpairs[k] = 0; return; }      // (reported by k_mark_records)
  const u32 a = br_rank(br, ia), b = br_rank(br, ib);
  const u32 ca = (u32)r.y, cb = (u32)(r.y >> 32);
  cpairs[k] = ((u64)a << 32) | b;
  ncnt[a] = ca;
  ncnt[b] = cb;
  atomicAdd(&deg[a], 1u);
  atomicAdd(&deg[b], 1u);
  if (!join_by_count || at_least_double(ca, cb) || at_least_double(cb, ca)) uf_union(parent, a, b);
}

// ---- the trivial components, and the lists the other cluster kernels work from ----------------------
// k_cluster_trivial (kernels_cluster.hip.h) for a compact graph (every node has neighbours; the number of
// nodes is read on the device: the launch is sized by a bound) which ALSO does what k_comp_count did in a
// pass of its own: the roots of the components of 3 .. SMALL_COMP nodes as a dense list (collected per
// workgroup in LDS, one global atomic per flush) and the number of nodes in larger components.
template <bool MAXIMUM>
__global__ void __launch_bounds__(256)
k_cg_trivial(const u32 *__restrict__ P, const u32 *__restrict__ csize, const u32 *__restrict__ m_dev, const u32 *__restrict__ cnt,
             const u32 *__restrict__ off, const u32 *__restrict__ idx, u32 *cl_of, u32 *maxleaf, u64 *cl_size, ull *ctr,
             u32 *__restrict__ small_roots) {
  HUMID_GUARD_LAST_VGPR();
  __shared__ u32 lds[4];
  __shared__ u32 lroots[CC_ROOTS], lroots_n, lroots_base;
  if (threadIdx.x == 0) lroots_n = 0;
  __syncthreads();
  const u32 n = *m_dev;
  const u32 lane = threadIdx.x & 63;
  u32 mb = 0, round = 0;
  for (u32 a0 = blockIdx.x * blockDim.x; a0 < n; a0 += gridDim.x * blockDim.x) {     // (whole waves stay in the loop: ballot below)
    const u32 a = a0 + threadIdx.x;
    bool small_root = false;
    if (a < n) {
      const u32 root = P[a];                         // flattened by k_comp_stats
      const u32 members = csize[root];
      if (members > SMALL_COMP) { mb++; cl_of[a] = 0; }
      else if (members > 2) { small_root = root == a; cl_of[a] = 0; }
      else if (members == 1) { cl_of[a] = a + 1; maxleaf[a] = a; cl_size[a] = cnt[a]; }
      else if (root == a) {                          // two nodes a < b: closed form, done by a's lane
        u32 b = a;
        for (u32 k = off[a]; k < off[a + 1]; k++) {
          const u32 nb = idx[k];
          if (P[nb] == root) { b = nb; break; }
        }
        const u64 ca = cnt[a], cb = cnt[b];
        if (MAXIMUM) {
          cl_of[a] = a + 1; cl_of[b] = a + 1;
          maxleaf[a] = (cb > ca) ? b : a;
          cl_size[a] = ca + cb;
        } else if (at_least_double(cb, ca)) {        // a climbs to b, b floods back to a
          cl_of[a] = a + 1; cl_of[b] = a + 1;
          maxleaf[a] = b;
          cl_size[a] = ca + cb;
        } else if (at_least_double(ca, cb)) {        // a stays, absorbs b
          cl_of[a] = a + 1; cl_of[b] = a + 1;
          maxleaf[a] = a;
          cl_size[a] = ca + cb;
        } else {                                     // two clusters; b finds a already assigned
          cl_of[a] = a + 1; maxleaf[a] = a; cl_size[a] = ca;
          cl_of[b] = b + 1; maxleaf[b] = b; cl_size[b] = cb;
        }
      }
    }
    const u64 bm = __ballot(small_root);
    if (bm) {
      u32 base = 0;
      if (lane == 0) base = atomicAdd(&lroots_n, (u32)__popcll(bm));
      base = __shfl(base, 0);
      if (small_root) lroots[base + (u32)__popcll(bm & ((1ull << lane) - 1ull))] = a;
    }
    if ((++round & (CC_ROOTS / 256u - 1u)) == 0) {   // every eighth round: flush (the buffer cannot overfill in between)
      __syncthreads();
      const u32 cntl = lroots_n;
      if (cntl) {
        if (threadIdx.x == 0) lroots_base = (u32)atomicAdd(&ctr[CTR_SMALLROOTS], (ull)cntl);
        __syncthreads();
        for (u32 k = threadIdx.x; k < cntl; k += blockDim.x) small_roots[lroots_base + k] = lroots[k];
        __syncthreads();
        if (threadIdx.x == 0) lroots_n = 0;
      }
      __syncthreads();
    }
  }
  __syncthreads();
  const u32 cntl = lroots_n;
  if (cntl) {
    if (threadIdx.x == 0) lroots_base = (u32)atomicAdd(&ctr[CTR_SMALLROOTS], (ull)cntl);
    __syncthreads();
    for (u32 k = threadIdx.x; k < cntl; k += blockDim.x) small_roots[lroots_base + k] = lroots[k];
  }
  const u32 tb = block_sum(mb, lds);
  if (threadIdx.x == 0 && tb) atomicAdd(&ctr[CTR_MEMBERS], (ull)tb);
}

// ---- results -------------------------------------------------------------------------------------------
// nodes that did not create a cluster, as a bitmap over walk indices
static __global__ void __launch_bounds__(256)
k_noncreator_bits(const u32 *__restrict__ cl_of, const u32 *__restrict__ nodes, u32 m, u32 *bits) {
  HUMID_GUARD_LAST_VGPR();
  const u32 c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= m || cl_of[c] == c + 1) return;
  const u32 u = nodes[c];
  atomicOr(&bits[u >> 5], 1u << (u & 31));
}

// per unique word: (read to keep, cluster id) at the slot of its word.  Cluster ids follow
// src/humid.cc:177-180: 1 + the number of cluster-creating leaves before the creator in the walk =
// 1 + creator - (nodes before it that created nothing).
static __global__ void __launch_bounds__(256)
k_finalize_leaves(BitRank in_graph, BitRank noncreator, const u32 *__restrict__ nodes, const u32 *__restrict__ cl_of,
                  const u32 *__restrict__ maxleaf, u32 n, const u32 *__restrict__ s_first, const u32 *__restrict__ s_slot,
                  u64 *__restrict__ slot_out, u32 *__restrict__ cid_out, u8 *__restrict__ ismax_out) {
  HUMID_GUARD_LAST_VGPR();
  const u32 u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= n) return;
  u32 g = u;
  bool mx = true;
  if (br_test(in_graph, u)) {
    const u32 c = br_rank(in_graph, u);
    const u32 cc = cl_of[c] - 1u;                      // compact index of the creator
    g = nodes[cc];
    mx = maxleaf[cc] == c;
  }
  const u32 id = 1u + g - br_rank(noncreator, g);
  if (slot_out) slot_out[s_slot ? s_slot[u] : u] = ((u64)(mx ? s_first[u] : NONE32) << 32) | id;
  if (cid_out) { cid_out[u] = id; ismax_out[u] = mx ? 1 : 0; }
}

// ---- the legacy (per unique word) view, built on demand for the accessors -------------------------
static __global__ void __launch_bounds__(256)
k_expand_leaves(BitRank in_graph, const u32 *__restrict__ nodes, const u32 *__restrict__ cdeg, const u32 *__restrict__ ccl_of,
                const u32 *__restrict__ cmaxleaf, const u64 *__restrict__ ccl_size, const u32 *__restrict__ cnt, u32 n,
                u32 *__restrict__ deg, u32 *__restrict__ cl_of, u32 *__restrict__ maxleaf, u64 *__restrict__ cl_size) {
  HUMID_GUARD_LAST_VGPR();
  const u32 u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u > n) return;
  if (u == n) { deg[u] = 0; return; }
  if (!br_test(in_graph, u)) { deg[u] = 0; cl_of[u] = u + 1; maxleaf[u] = u; cl_size[u] = cnt[u]; return; }
  const u32 c = br_rank(in_graph, u);
  deg[u] = cdeg[c];
  const u32 cc = ccl_of[c] - 1u;
  cl_of[u] = nodes[cc] + 1u;
  if (cc == c) { maxleaf[u] = nodes[cmaxleaf[c]]; cl_size[u] = ccl_size[c]; }
}
static __global__ void __launch_bounds__(256)
k_expand_rows(const u32 *__restrict__ nodes, const u32 *__restrict__ coff, const u32 *__restrict__ cidx, u32 m,
              const u32 *__restrict__ off, u32 *__restrict__ idx) {
  HUMID_GUARD_LAST_VGPR();
  const u32 c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= m) return;
  const u32 b = coff[c], d = coff[c + 1] - b, o = off[nodes[c]];
  for (u32 k = 0; k < d; k++) idx[o + k] = nodes[cidx[b + k]];
}

#endif  // HUMID_KERNELS_CGRAPH_HIP_H
