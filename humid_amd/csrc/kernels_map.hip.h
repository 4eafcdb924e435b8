// kernels_map.hip.h -- per-read outputs, multi-GPU result return, histograms
// Part of libhumid_hip.so (see humid_hip.hip for the pipeline and the C ABI).  Device code for
// gfx950 only; included once, in this order, by humid_hip.hip.
#ifndef HUMID_KERNELS_MAP_HIP_H
#define HUMID_KERNELS_MAP_HIP_H

#include "common.hip.h"
#include "kernels_graph.hip.h"

// --------------------------------------------------------------------------------
// 6. per-read map: cluster id and the duplicate flag
// --------------------------------------------------------------------------------
// keep = this read is the first (input order) whose word is its cluster's maxLeaf
// (/root/reference/src/humid.cc:224-231); cluster 0 for filtered reads (:272).
static __global__ void __launch_bounds__(256)
k_read_map(const u32 *__restrict__ slot_of_read, const u64 *__restrict__ slot_out, u32 n_reads,
           u32 *__restrict__ cluster_id, u8 *__restrict__ keep) {
  HUMID_GUARD_LAST_VGPR();
  for (u32 r = blockIdx.x * blockDim.x + threadIdx.x; r < n_reads; r += gridDim.x * blockDim.x) {
    const u32 s = slot_of_read[r];
    u32 c = 0;
    u8 k = 0;
    if (s != NOSLOT) {
      const u64 o = slot_out[s];
      c = (u32)o;
      k = ((u32)(o >> 32) == r) ? 1 : 0;
    }
    cluster_id[r] = c;
    keep[r] = k;
  }
}

// The same in PARTITION order (LDS-partitioned counts): position i of the partitioned arrays
// holds read vals[i] and the slot of its word; the slot lookups are partition-local (cached).
// The un-permute is ONE scattered 4-byte store per read (cluster id | keep << 31; ids < 2^31
// because n_reads < 2^31); k_split_out then writes both output arrays coalesced.
static __global__ void __launch_bounds__(256)
k_read_map_part(const u32 *__restrict__ vals, const u32 *__restrict__ pslot, const u64 *__restrict__ slot_out,
                u32 n_reads, u32 *__restrict__ packed) {
  HUMID_GUARD_LAST_VGPR();
  for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n_reads; i += gridDim.x * blockDim.x) {
    const u32 r = vals[i] & 0x7fffffffu;
    if (r >= n_reads) continue;
    const u32 s = pslot[i];
    u32 c = 0;
    if (s != NOSLOT) {
      const u64 o = slot_out[s];
      c = (u32)o | (((u32)(o >> 32) == r) ? 0x80000000u : 0u);
    }
    packed[r] = c;
  }
}

// The same per BUCKET of the LDS-partitioned counts (one workgroup per bucket, like k_dedup_lds): a
// bucket's positions refer to its own unique words only -- padded slots [beg, beg + ucount) -- so
// their result words are first copied into LDS with one coalesced read and the per-position look-up
// never leaves the CU.  What remains is the scattered store itself (a bare random scatter of 10 M
// 4-byte values takes 0.13 ms on this GPU, tools/scatter_roofline.py).
static __global__ void __launch_bounds__(256)
k_read_map_bucket(const u32 *__restrict__ vals, const u32 *__restrict__ pslot, const u64 *__restrict__ slot_out,
                  const u32 *__restrict__ pbeg, const u32 *__restrict__ ucount, u32 n_reads,
                  u32 *__restrict__ packed) {
  HUMID_GUARD_LAST_VGPR();
  __shared__ u64 lres[LDS_SLOTS + 1];
  const u32 b = blockIdx.x;
  const u32 beg = pbeg[b], end = pbeg[b + 1];
  if (beg >= end || end > n_reads) return;
  u32 uc = ucount[b];
  if (uc > LDS_SLOTS + 1) uc = LDS_SLOTS + 1;               // never more than the table held
  for (u32 li = threadIdx.x; li < uc; li += 256) lres[li] = slot_out[beg + li];
  __syncthreads();
  for (u32 i = beg + threadIdx.x; i < end; i += 256) {
    const u32 r = vals[i] & 0x7fffffffu;
    if (r >= n_reads) continue;
    const u32 s = pslot[i];
    u32 c = 0;
    if (s != NOSLOT && s - beg < uc) {
      const u64 o = lres[s - beg];
      c = (u32)o | (((u32)(o >> 32) == r) ? 0x80000000u : 0u);
    }
    packed[r] = c;
  }
}

// global-table variant with the packed result word as output (every read is owned and usable)
static __global__ void __launch_bounds__(256)
k_read_map_packed(const u32 *__restrict__ slot_of_read, const u64 *__restrict__ slot_out, u32 n_reads,
                  u32 *__restrict__ packed) {
  HUMID_GUARD_LAST_VGPR();
  for (u32 r = blockIdx.x * blockDim.x + threadIdx.x; r < n_reads; r += gridDim.x * blockDim.x) {
    const u32 s = slot_of_read[r];
    u32 c = 0;
    if (s != NOSLOT) {
      const u64 o = slot_out[s];
      c = (u32)o | (((u32)(o >> 32) == r) ? 0x80000000u : 0u);
    }
    packed[r] = c;
  }
}

static __global__ void __launch_bounds__(256)
k_split_out(const u32 *__restrict__ packed, u32 n_reads, u32 *__restrict__ cluster_id, u8 *__restrict__ keep) {
  HUMID_GUARD_LAST_VGPR();
  for (u32 r = blockIdx.x * blockDim.x + threadIdx.x; r < n_reads; r += gridDim.x * blockDim.x) {
    const u32 t = packed[r];
    cluster_id[r] = t & 0x7fffffffu;
    keep[r] = (u8)(t >> 31);
  }
}

// --------------------------------------------------------------------------------
// 7. multi-GPU result return: dense per-shard streams instead of N-sized arrays
// --------------------------------------------------------------------------------
#define MAX_RANKS 16
struct OwnerRanges {            // value ranges of the ranks, by value, statically indexed
  u64 lo[MAX_RANKS];
  u64 hi[MAX_RANKS];
};

struct OwnedRangeFlagOp {       // 1 for the usable reads whose word lies in [lo, hi]
  const u64 *words;
  const u8 *filtered;
  u64 lo, hi;
  u32 n;
  __device__ u32 operator()(u32 i) const {
    if (i >= n || filtered[i]) return 0u;
    const u64 w = words[i];
    return (w >= lo && w <= hi) ? 1u : 0u;
  }
};

// dense copy of the owned reads' words, in read order
static __global__ void __launch_bounds__(256)
k_gather_owned(const u64 *__restrict__ words, const u8 *__restrict__ filtered, const u32 *__restrict__ opos,
               u64 lo, u64 hi, u32 n_reads, u64 *__restrict__ own_words) {
  HUMID_GUARD_LAST_VGPR();
  for (u32 r = blockIdx.x * blockDim.x + threadIdx.x; r < n_reads; r += gridDim.x * blockDim.x) {
    if (filtered[r]) continue;
    const u64 w = words[r];
    if (w >= lo && w <= hi) own_words[opos[r]] = w;
  }
}

// the same for two-word words: the range is one of HEADS (k_wide_head64), the words follow
static __global__ void __launch_bounds__(256)
k_gather_owned_w2(const W2 *__restrict__ words, const u64 *__restrict__ heads, const u8 *__restrict__ filtered,
                  const u32 *__restrict__ opos, u64 lo, u64 hi, u32 n_reads, W2 *__restrict__ own_words) {
  HUMID_GUARD_LAST_VGPR();
  for (u32 r = blockIdx.x * blockDim.x + threadIdx.x; r < n_reads; r += gridDim.x * blockDim.x) {
    if (filtered[r]) continue;
    const u64 h = heads[r];
    if (h >= lo && h <= hi) own_words[opos[r]] = words[r];
  }
}

struct OwnedFlagOp {            // 1 for the reads this rank counted (global-table variant)
  const u32 *slot_of_read;
  u32 n;
  __device__ u32 operator()(u32 i) const { return (i < n && slot_of_read[i] != NOSLOT) ? 1u : 0u; }
};

// packed result (cluster id | keep << 31) of every owned read, dense, in read order
static __global__ void __launch_bounds__(256)
k_owned_results(const u32 *__restrict__ slot_of_read, const u32 *__restrict__ opos,
                const u64 *__restrict__ slot_out, u32 n_reads, u32 *__restrict__ packed) {
  HUMID_GUARD_LAST_VGPR();
  for (u32 r = blockIdx.x * blockDim.x + threadIdx.x; r < n_reads; r += gridDim.x * blockDim.x) {
    const u32 s = slot_of_read[r];
    if (s == NOSLOT) continue;
    const u64 o = slot_out[s];
    packed[opos[r]] = (u32)o | (((u32)(o >> 32) == r) ? 0x80000000u : 0u);
  }
}

// owner rank of every local read (n_ranks = nobody: filtered reads)
static __global__ void __launch_bounds__(256)
k_owner_of(const u64 *__restrict__ words, const u8 *__restrict__ filtered, u32 n, OwnerRanges rg,
           u32 n_ranks, u8 *__restrict__ owner) {
  HUMID_GUARD_LAST_VGPR();
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u32 o = n_ranks;
  if (!filtered[i]) {
    const u64 w = words[i];
#pragma unroll
    for (u32 q = 0; q < MAX_RANKS; q++)
      if (q < n_ranks && rg.lo[q] <= rg.hi[q] && w >= rg.lo[q] && w <= rg.hi[q]) o = q;
  }
  owner[i] = (u8)o;
}

// first position of every owner in the owner-sorted order (n_ranks + 2 boundaries)
static __global__ void k_owner_bounds(const u8 *__restrict__ sorted_owner, u32 n, u32 n_ranks, u32 *__restrict__ bounds) {
  HUMID_GUARD_LAST_VGPR();
  u32 q = threadIdx.x;
  if (q > n_ranks + 1) return;
  u32 lo = 0, hi = n;
  while (lo < hi) {
    u32 mid = lo + ((hi - lo) >> 1);
    if (sorted_owner[mid] < q) lo = mid + 1; else hi = mid;
  }
  bounds[q] = lo;
}

// received dense stream (owner-major, read order inside) -> this shard's outputs
static __global__ void __launch_bounds__(256)
k_scatter_results(const u32 *__restrict__ perm, const u32 *__restrict__ packed, u32 n_recv,
                  u32 *__restrict__ cluster_id, u8 *__restrict__ keep) {
  HUMID_GUARD_LAST_VGPR();
  for (u32 k = blockIdx.x * blockDim.x + threadIdx.x; k < n_recv; k += gridDim.x * blockDim.x) {
    const u32 r = perm[k];
    const u32 t = packed[k];
    cluster_id[r] = t & 0x7fffffffu;
    keep[r] = (u8)(t >> 31);
  }
}

// --------------------------------------------------------------------------------
// 8. multi-GPU exchange mode: routing of unique words by combination key, compact node lists
// --------------------------------------------------------------------------------
// owner rank of a unique word for combination `cf`: a hash of its key, so that all words of one
// bucket meet on one rank whatever the key distribution is
template <class WT>
__global__ void __launch_bounds__(256)
k_combo_owner(const WT *__restrict__ words, u32 n, ComboFields cf, u32 n_ranks, u8 *__restrict__ owner) {
  HUMID_GUARD_LAST_VGPR();
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const WT w = words[i];
  u64 k = 0;
#pragma unroll
  for (u32 f = 0; f < MAX_FIELDS; f++) {
    if (f < cf.nf) {
      const u32 wd = cf.width[f];
      k = ((wd >= 64) ? 0ull : (k << wd)) | w_field(w, cf.shift[f], wd);
    }
  }
  const u64 h = mix64(k ^ 0x9e3779b97f4a7c15ull);
  owner[i] = (u8)(((h >> 32) * (u64)n_ranks) >> 32);
}

// (word, id | count << 32) items in routed order: id = id_base + index in the local unique array
static __global__ void __launch_bounds__(256)
k_route_items(const u64 *__restrict__ words, const u32 *__restrict__ counts, const u32 *__restrict__ perm, u32 n,
              u64 id_base, ulonglong2 *__restrict__ items) {
  HUMID_GUARD_LAST_VGPR();
  u32 k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const u32 i = perm ? perm[k] : k;                  // perm == null: one rank, the array goes as it stands
  items[k] = make_ulonglong2(words[i], (id_base + i) | ((u64)(counts ? counts[i] : 0u) << 32));
}

// ---- the same for two-word (wide) words: 24-byte items (hi, lo, id | count << 32) ----
struct Item3 { u64 hi, lo, idc; };
static __global__ void __launch_bounds__(256)
k_route_items_w2(const W2 *__restrict__ words, const u32 *__restrict__ counts, const u32 *__restrict__ perm, u32 n,
                 u64 id_base, Item3 *__restrict__ items) {
  HUMID_GUARD_LAST_VGPR();
  u32 k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const u32 i = perm ? perm[k] : k;
  const W2 w = words[i];
  items[k] = Item3{w.hi, w.lo, (id_base + i) | ((u64)(counts ? counts[i] : 0u) << 32)};
}
static __global__ void k_split_items_w2(const Item3 *__restrict__ items, u32 n, W2 *__restrict__ w, u32 *__restrict__ id,
                                 u32 *__restrict__ cnt) {
  HUMID_GUARD_LAST_VGPR();
  u32 k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const Item3 it = items[k];
  w[k] = W2{it.hi, it.lo};
  id[k] = (u32)it.idc;
  cnt[k] = (u32)(it.idc >> 32);
}
// the top 64 bits of a wide word's 2n-bit value (hbits = 2 (n - 32) bits live in .hi): value ranges cut
// at the bins of a <= 12-bit prefix histogram are decided by these bits alone
// (drop: that many low bits of the head are shifted out -- the count stage's partition keys, WideReadsSrc)
static __global__ void k_wide_head64(const W2 *__restrict__ w, u32 n, u32 hbits, u64 *__restrict__ out, u32 drop = 0) {
  HUMID_GUARD_LAST_VGPR();
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const W2 x = w[i];
  out[i] = (hbits >= 64 ? x.hi : ((x.hi << (64 - hbits)) | (x.lo >> hbits))) >> drop;
}
static __global__ void k_gather_w2(const W2 *__restrict__ w, const u32 *__restrict__ perm, u32 n, W2 *__restrict__ out) {
  HUMID_GUARD_LAST_VGPR();
  u32 k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) out[k] = w[perm[k]];
}

static __global__ void k_split_items(const ulonglong2 *__restrict__ items, u32 n, u64 *__restrict__ w,
                              u32 *__restrict__ id, u32 *__restrict__ cnt) {
  HUMID_GUARD_LAST_VGPR();
  u32 k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const ulonglong2 it = items[k];
  w[k] = it.x;
  id[k] = (u32)it.y;
  cnt[k] = (u32)(it.y >> 32);
}

// pairs over item POSITIONS -> records {smaller id << 32 | larger id, count(smaller) | count(larger) << 32}
// id_of == null: id = id_base + position (the plain ascending array of the prefix combination)
static __global__ void __launch_bounds__(256)
k_edge_records(const u64 *__restrict__ pos_edges, u32 n_edges, const u32 *__restrict__ id_of, u32 id_base,
               const u32 *__restrict__ cnt_of, ulonglong2 *__restrict__ rec) {
  HUMID_GUARD_LAST_VGPR();
  u32 k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_edges) return;
  const u64 e = pos_edges[k];
  const u32 pa = (u32)(e >> 32), pb = (u32)e;
  u32 a = id_of ? id_of[pa] : id_base + pa, b = id_of ? id_of[pb] : id_base + pb;
  u32 ca = cnt_of ? cnt_of[pa] : 0u, cb = cnt_of ? cnt_of[pb] : 0u;
  if (a > b) { u32 t = a; a = b; b = t; t = ca; ca = cb; cb = t; }
  rec[k] = make_ulonglong2(((u64)a << 32) | b, (u64)ca | ((u64)cb << 32));
}

static __global__ void k_iota_base(u32 *p, u32 n, u32 base) {
  HUMID_GUARD_LAST_VGPR();
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = base + i;
}

static __global__ void k_gather_u32(const u32 *__restrict__ src, const u32 *__restrict__ idx, u32 n, u32 *__restrict__ dst) {
  HUMID_GUARD_LAST_VGPR();
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[idx[i]];
}

// both endpoints of every edge (smaller << 32 | larger) with their slot 2k / 2k+1; stride = uint64
// per edge record (1 or 2)
static __global__ void k_edge_ends(const u64 *__restrict__ edges, u32 n_edges, u32 stride, u32 *__restrict__ ends,
                            u32 *__restrict__ slot) {
  HUMID_GUARD_LAST_VGPR();
  u32 k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_edges) return;
  const u64 e = edges[(size_t)k * stride];
  ends[2 * k] = (u32)(e >> 32);
  ends[2 * k + 1] = (u32)e;
  slot[2 * k] = 2 * k;
  slot[2 * k + 1] = 2 * k + 1;
}

// head[i] = 1 where a new value starts in the sorted array; head[n] = 0 (scan sentinel)
static __global__ void k_heads_u32(const u32 *__restrict__ sorted, u32 n, u32 *__restrict__ head) {
  HUMID_GUARD_LAST_VGPR();
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i > n) return;
  head[i] = (i < n && (i == 0 || sorted[i] != sorted[i - 1])) ? 1u : 0u;
}

static __global__ void k_compact_heads_u32(const u32 *__restrict__ sorted, const u32 *__restrict__ head,
                                    const u32 *__restrict__ hpos, u32 n, u32 *__restrict__ out) {
  HUMID_GUARD_LAST_VGPR();
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && head[i]) out[hpos[i]] = sorted[i];
}

// sorted endpoint i (slot = which end of which edge) -> its position in the node list, written back
// to the edge's slot: cends[slot] = hpos[i] + head[i] - 1; with records (stride 2) also the node's
// count (all records of a node carry the same count).  One scattered 4-byte store per endpoint
// instead of a binary search over the node list.
static __global__ void k_relabel_ends(const u32 *__restrict__ slot_s, const u32 *__restrict__ head,
                               const u32 *__restrict__ hpos, u32 n_ends, const u64 *__restrict__ records,
                               u32 stride, u32 *__restrict__ cends, u32 *__restrict__ node_cnt) {
  HUMID_GUARD_LAST_VGPR();
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_ends) return;
  const u32 node = hpos[i] + head[i] - 1u;
  const u32 sl = slot_s[i];
  cends[sl] = node;
  if (stride == 2 && head[i]) {
    const u64 cc = records[(size_t)(sl >> 1) * 2 + 1];
    node_cnt[node] = (sl & 1) ? (u32)(cc >> 32) : (u32)cc;
  }
}

// (position of the smaller end, position of the larger end) -> one 64-bit compact edge
static __global__ void k_pack_cedges(const u32 *__restrict__ cends, u32 n_edges, u64 *__restrict__ out) {
  HUMID_GUARD_LAST_VGPR();
  u32 k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n_edges) out[k] = ((u64)cends[2 * k] << 32) | cends[2 * k + 1];
}

// ---- compact node list through a mark array (ids below a known bound) instead of a sort ----
// mark[id] = 1 and cnt_of[id] = count for both ends of every pair record (equal values race freely)
static __global__ void k_mark_ends(const u64 *__restrict__ records, u32 n_edges, u32 stride, u32 id_bound,
                            u8 *__restrict__ mark, u32 *__restrict__ cnt_of, ull *ctr) {
  HUMID_GUARD_LAST_VGPR();
  u32 k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_edges) return;
  const u64 e = records[(size_t)k * stride];
  const u32 a = (u32)(e >> 32), b = (u32)e;
  if (a >= id_bound || b >= id_bound) { ctr[CTR_OVERFULL] = 1; return; }
  mark[a] = 1;
  mark[b] = 1;
  if (stride == 2) {
    const u64 cc = records[(size_t)k * 2 + 1];
    cnt_of[a] = (u32)cc;
    cnt_of[b] = (u32)(cc >> 32);
  }
}
// pos = exclusive scan of mark: nodes[pos[id]] = id, node_cnt[pos[id]] = cnt_of[id] for the marked ids
static __global__ void k_marked_nodes(const u8 *__restrict__ mark, const u32 *__restrict__ pos, const u32 *__restrict__ cnt_of,
                               u32 id_bound, bool counts, u32 *__restrict__ nodes, u32 *__restrict__ node_cnt) {
  HUMID_GUARD_LAST_VGPR();
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= id_bound || !mark[i]) return;
  const u32 q = pos[i];
  nodes[q] = i;
  if (counts) node_cnt[q] = cnt_of[i];
}
static __global__ void k_relabel_pairs(const u64 *__restrict__ records, u32 n_edges, u32 stride, u32 id_bound,
                                const u32 *__restrict__ pos, u64 *__restrict__ out) {
  HUMID_GUARD_LAST_VGPR();
  u32 k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_edges) return;
  const u64 e = records[(size_t)k * stride];
  const u32 a = (u32)(e >> 32), b = (u32)e;
  out[k] = (a < id_bound && b < id_bound) ? (((u64)pos[a] << 32) | pos[b]) : 0ull;
}

// routed copy of the usable reads' words (owner-major order of humid_stage_owner_perm)
static __global__ void __launch_bounds__(256)
k_route_words(const u64 *__restrict__ words, const u32 *__restrict__ perm, u32 n, u64 *__restrict__ out) {
  HUMID_GUARD_LAST_VGPR();
  for (u32 k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) out[k] = words[perm[k]];
}

// ---- routing of a rank's usable reads to the owners of their value ranges, STABLE ----
// (humid_stage_route.)  The routed array must keep the input order inside every owner's block: the
// owner takes "position in what it received" for "input order" when it decides which read of a word
// came first (the keep rule, src/humid.cc:224-231).  Three kernels over tiles of ROUTE_TILE reads:
// per-tile owner counts, one block that turns them into per-tile offsets inside every owner's block
// (the owners' block bases come from the host: it knows the totals from the all-gathered histograms),
// and the scatter, whose in-tile ranks come from wave ballots (at most 16 owners).
#define ROUTE_TILE 8192u
__device__ __forceinline__ u32 owner_of_word(const OwnerRanges &rg, u32 n_ranks, u64 w) {
  u32 o = n_ranks;
#pragma unroll
  for (u32 q = 0; q < MAX_RANKS; q++)
    if (q < n_ranks && rg.lo[q] <= rg.hi[q] && w >= rg.lo[q] && w <= rg.hi[q]) o = q;
  return o;
}

// Owner of a word through a table in LDS when the value ranges are cut at the bins of a <= 12-bit
// prefix histogram (they are: humid_amd/sharded.py, csrc/host/sharded.cpp): own[w >> shift], filled by
// the block itself from the ranges (4 bins per thread).  One LDS read per read instead of 16 pairs of
// 64-bit compares.  TABLE = false: the compare loop (ranges of any other shape).
#define ROUTE_BINS 4096u
// the table is computed once (k_route_table, one block) and copied into LDS by every workgroup: filling it
// from the ranges in each of the ~1200 workgroups cost as many instructions as the routing itself
static __global__ void __launch_bounds__(1024)
k_route_table(OwnerRanges rg, u32 n_ranks, u32 shift, u8 *__restrict__ table) {
  HUMID_GUARD_LAST_VGPR();
  for (u32 b = threadIdx.x; b < ROUTE_BINS; b += blockDim.x) table[b] = (u8)owner_of_word(rg, n_ranks, (u64)b << shift);
}
template <bool TABLE>
__device__ __forceinline__ void route_fill_table(u8 *own, const u8 *__restrict__ table) {
  if (!TABLE) return;
  for (u32 b = threadIdx.x; b < ROUTE_BINS / 4; b += blockDim.x) ((u32 *)own)[b] = ((const u32 *)table)[b];
  __syncthreads();
}
template <bool TABLE>
__device__ __forceinline__ u32 route_owner(const u8 *own, const OwnerRanges &rg, u32 n_ranks, u32 shift, u64 w) {
  if (!TABLE) return owner_of_word(rg, n_ranks, w);
  const u64 b = w >> shift;
  return own[b < ROUTE_BINS ? (u32)b : ROUTE_BINS - 1];
}

// per tile of ROUTE_TILE reads: usable reads of every owner.  All loads of a thread are issued before
// the first is used; a wave counts an owner with one ballot, lane q keeps owner q's count.
template <bool TABLE>
__global__ void __launch_bounds__(1024)
k_route_tile_hist(const u64 *__restrict__ words, const u8 *__restrict__ filtered, u32 n, OwnerRanges rg,
                  u32 n_ranks, u32 shift, const u8 *__restrict__ table, u32 *__restrict__ tile_cnt, u32 *__restrict__ bad) {
  HUMID_GUARD_LAST_VGPR();
  __shared__ u32 cnt[MAX_RANKS];
  __shared__ __attribute__((aligned(16))) u8 own[TABLE ? ROUTE_BINS : 4];
  if (blockIdx.x == 0 && threadIdx.x == 0) bad[0] = 0;       // k_route_scan (next launch) may raise it
  if (threadIdx.x < MAX_RANKS) cnt[threadIdx.x] = 0;
  const u32 beg = blockIdx.x * ROUTE_TILE;
  u64 w[ROUTE_TILE / 1024];
  u8 f[ROUTE_TILE / 1024];
#pragma unroll
  for (u32 k = 0; k < ROUTE_TILE / 1024; k++) {
    const u32 j = beg + k * 1024 + threadIdx.x;
    f[k] = j < n ? filtered[j] : (u8)1;
    w[k] = j < n ? words[j] : 0;
  }
  route_fill_table<TABLE>(own, table);
  if (!TABLE) __syncthreads();
  const u32 lane = threadIdx.x & 63;
  u32 acc = 0;
#pragma unroll
  for (u32 k = 0; k < ROUTE_TILE / 1024; k++) {
    const u32 o = f[k] ? MAX_RANKS : route_owner<TABLE>(own, rg, n_ranks, shift, w[k]);
    for (u32 q = 0; q < n_ranks; q++) {
      const u64 m = __ballot(o == q);
      if (lane == q) acc += (u32)__popcll(m);
    }
  }
  if (lane < n_ranks && acc) atomicAdd(&cnt[lane], acc);
  __syncthreads();
  if (threadIdx.x < MAX_RANKS) tile_cnt[blockIdx.x * MAX_RANKS + threadIdx.x] = cnt[threadIdx.x];
}

struct OwnerBases { u32 b[MAX_RANKS + 1]; };       // first routed position of every owner's block

// one block, one wave per owner: tile_cnt[t][q] -> exclusive offset of tile t inside owner q's block
// (in place), + the block's base; bad[0] = 1 if an owner's total differs from the host's count
static __global__ void __launch_bounds__(1024)
k_route_scan(u32 *tile_cnt, u32 n_tiles, OwnerBases ob, u32 *bad) {
  HUMID_GUARD_LAST_VGPR();
  const u32 q = threadIdx.x >> 6, lane = threadIdx.x & 63;       // 16 waves = MAX_RANKS owners
  u32 run = ob.b[q];
  // chunks of 16 rows of 64 tiles: the chunk's counts are requested together (one memory round trip per chunk, not
  // per row: the kernel is ONE workgroup, its time is its chain of round trips), then scanned row by row
  constexpr u32 ROWS = 16;
  for (u32 c0 = 0; c0 < n_tiles; c0 += 64 * ROWS) {
    u32 x[ROWS];
#pragma unroll
    for (u32 k = 0; k < ROWS; k++) {
      const u32 t = c0 + 64 * k + lane;
      x[k] = t < n_tiles ? tile_cnt[t * MAX_RANKS + q] : 0u;
    }
#pragma unroll
    for (u32 k = 0; k < ROWS; k++) {
      const u32 t = c0 + 64 * k + lane;
      const u32 incl = wave_incl_scan(x[k]);
      if (t < n_tiles) tile_cnt[t * MAX_RANKS + q] = run + incl - x[k];
      run += (u32)__builtin_amdgcn_readlane((int)incl, 63);
    }
  }
  if (lane == 0 && run != ob.b[q + 1]) bad[0] = 1;
}

// stable scatter of a tile: position = tile offset of the owner + usable reads of that owner before
// this one inside the tile.  Loads first (all in flight), then per round and wave one ballot per
// owner (rank inside the wave + the wave's count), ONE pass of 16 threads that turns the
// [round][wave][owner] counts into offsets, then the stores: two barriers per tile.
template <bool TABLE>
__global__ void __launch_bounds__(1024)
k_route_scatter(const u64 *__restrict__ words, const u8 *__restrict__ filtered, u32 n, OwnerRanges rg,
                u32 n_ranks, u32 shift, const u8 *__restrict__ table, const u32 *__restrict__ tile_off,
                u64 *__restrict__ routed, u32 *__restrict__ perm, u32 *__restrict__ inv = nullptr) {
  HUMID_GUARD_LAST_VGPR();
  constexpr u32 R = ROUTE_TILE / 1024;
  __shared__ u32 wcnt[R][16][MAX_RANKS];   // reads of every owner per round and wave -> their offsets
  __shared__ __attribute__((aligned(16))) u8 own[TABLE ? ROUTE_BINS : 4];
  const u32 wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const u64 lt = (1ull << lane) - 1ull;
  const u32 beg = blockIdx.x * ROUTE_TILE;
  u64 w[R];
  u8 f[R];
#pragma unroll
  for (u32 k = 0; k < R; k++) {
    const u32 j = beg + k * 1024 + threadIdx.x;
    f[k] = j < n ? filtered[j] : (u8)1;
    w[k] = j < n ? words[j] : 0;
  }
  route_fill_table<TABLE>(own, table);
  u32 o[R], rk[R];
#pragma unroll
  for (u32 k = 0; k < R; k++) {
    o[k] = f[k] ? MAX_RANKS : route_owner<TABLE>(own, rg, n_ranks, shift, w[k]);
    rk[k] = 0;
    for (u32 q = 0; q < n_ranks; q++) {
      const u64 m = __ballot(o[k] == q);
      if (o[k] == q) rk[k] = (u32)__popcll(m & lt);
      if (lane == q) wcnt[k][wave][q] = (u32)__popcll(m);
    }
  }
  __syncthreads();
  if (threadIdx.x < n_ranks) {
    u32 run = tile_off[blockIdx.x * MAX_RANKS + threadIdx.x];
    for (u32 k = 0; k < R; k++)
      for (u32 w2 = 0; w2 < 16; w2++) {
        const u32 c = wcnt[k][w2][threadIdx.x];
        wcnt[k][w2][threadIdx.x] = run;
        run += c;
      }
  }
  __syncthreads();
  // inv (may be null): routed position of every read (~0: filtered / nobody's) -- the result return then
  // GATHERS per read with coalesced stores (k_gather_results) instead of two memsets and a scatter
#pragma unroll
  for (u32 k = 0; k < R; k++) {
    const u32 j = beg + k * 1024 + threadIdx.x;
    if (o[k] < n_ranks) {
      const u32 pos = wcnt[k][wave][o[k]] + rk[k];
      routed[pos] = w[k];
      perm[pos] = j;
      if (inv) inv[j] = pos;
    } else if (inv && j < n) inv[j] = NONE32;
  }
}

// this shard's outputs from the received dense stream through the routed position of every read
static __global__ void __launch_bounds__(256)
k_gather_results(const u32 *__restrict__ inv, const u32 *__restrict__ packed, u32 n_recv, u32 n_reads,
                 u32 *__restrict__ cluster_id, u8 *__restrict__ keep) {
  HUMID_GUARD_LAST_VGPR();
  for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n_reads; i += gridDim.x * blockDim.x) {
    const u32 p = inv[i];
    const u32 t = p < n_recv ? packed[p] : 0u;
    cluster_id[i] = t & 0x7fffffffu;
    keep[i] = (u8)(t >> 31);
  }
}

// ---- cluster ids of one rank's unique words from the replicated compact graph ----
// creator (smallest member = the leaf whose walk step created the cluster) of every compact cluster
static __global__ void k_xid_creators(const u32 *__restrict__ ccid, u32 n_nodes, u32 n_clusters, u32 *creator) {
  HUMID_GUARD_LAST_VGPR();
  u32 k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_nodes) return;
  const u32 c = ccid[k];
  if (c >= 1 && c <= n_clusters) atomicMin(&creator[c - 1], k);
}

// base_id[c] = global cluster id - 1 of compact cluster c: creators before it in the whole walk =
// singletons before its creator (global index - compact position) + compact creators before it (c)
static __global__ void k_xid_base(const u32 *__restrict__ nodes, const u32 *__restrict__ creator, u32 n_nodes,
                           u32 n_clusters, u32 goff, u32 u_local, u32 *__restrict__ base_id,
                           u32 *__restrict__ mark_cr) {
  HUMID_GUARD_LAST_VGPR();
  u32 c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n_clusters) return;
  const u32 k = creator[c];
  if (k >= n_nodes) { base_id[c] = 0; return; }            // malformed ids: never index with them
  const u32 g = nodes[k];
  base_id[c] = g - k + c;
  if (g >= goff && g - goff < u_local) mark_cr[g - goff] = 1u;
}

// mark[i] = compact position + 1 of local unique word i (0 = singleton)
static __global__ void k_xid_mark(const u32 *__restrict__ nodes, u32 n_nodes, u32 goff, u32 u_local, u32 *__restrict__ mark) {
  HUMID_GUARD_LAST_VGPR();
  u32 k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_nodes) return;
  const u32 g = nodes[k];
  if (g >= goff && g - goff < u_local) mark[g - goff] = k + 1u;
}

// first[0] = compact nodes below goff, first[1] = compact creators below goff (binary searches)
static __global__ void k_xid_first(const u32 *__restrict__ nodes, const u32 *__restrict__ creator, u32 n_nodes,
                            u32 n_clusters, u32 goff, u32 *__restrict__ first) {
  HUMID_GUARD_LAST_VGPR();
  if (threadIdx.x == 0) {
    u32 lo = 0, hi = n_nodes;
    while (lo < hi) { const u32 mid = lo + ((hi - lo) >> 1); if (nodes[mid] < goff) lo = mid + 1; else hi = mid; }
    first[0] = lo;
  } else if (threadIdx.x == 1) {
    u32 lo = 0, hi = n_clusters;               // creators ascend with the cluster id
    while (lo < hi) {
      const u32 mid = lo + ((hi - lo) >> 1);
      const u32 k = creator[mid];
      if (k < n_nodes && nodes[k] < goff) lo = mid + 1; else hi = mid;
    }
    first[1] = lo;
  }
}

struct XidFlagOp {               // (is compact) | (is compact creator) << 32, scanned in one pass
  const u32 *mark, *mark_cr;
  u32 n;
  __device__ u64 operator()(u32 i) const {
    if (i >= n) return 0ull;
    return (u64)(mark[i] != 0u) | ((u64)(mark_cr[i] != 0u) << 32);
  }
};

static __global__ void __launch_bounds__(256)
k_xid_assign(const u32 *__restrict__ mark, const u64 *__restrict__ scan, const u32 *__restrict__ first,
             const u32 *__restrict__ ccid, const u8 *__restrict__ cismax, const u32 *__restrict__ base_id,
             u32 n_clusters, u32 goff, u32 u_local, u32 *__restrict__ l_cid, u8 *__restrict__ l_ismax) {
  HUMID_GUARD_LAST_VGPR();
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= u_local) return;
  const u32 m = mark[i];
  if (m) {
    const u32 c = ccid[m - 1];
    l_cid[i] = (c >= 1 && c <= n_clusters) ? 1u + base_id[c - 1] : 0u;
    l_ismax[i] = cismax[m - 1];
  } else {
    const u64 sc = scan[i];
    const u32 nb = first[0] + (u32)sc, cb = first[1] + (u32)(sc >> 32);
    l_cid[i] = 1u + (goff + i) - nb + cb;     // creators before it: singletons + compact creators
    l_ismax[i] = 1;
  }
}

static __global__ void k_widen32(const u32 *__restrict__ in, u32 n, u64 *__restrict__ out) {
  HUMID_GUARD_LAST_VGPR();
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i];
}

static __global__ void k_creator_sizes(const u32 *__restrict__ flag, const u32 *__restrict__ pos,
                                const u64 *__restrict__ cl_size, u32 n, u64 *__restrict__ out) {
  HUMID_GUARD_LAST_VGPR();
  u32 u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u < n && flag[u]) out[pos[u]] = cl_size[u];
}

// reads per top-`bits` bin of (word - lo) * scale (usable reads only): balanced range splitters for
// the multi-GPU path (lo = 0, scale = 2^(64-2n): the top bits of the word itself) and the
// uniformity check of the word-ordered buckets.  LDS-privatised, fixed grid.
static __global__ void __launch_bounds__(1024)
k_top_hist(const u64 *__restrict__ words, const u8 *__restrict__ filtered, u32 n_reads, u64 lo, u64 scale,
           u32 bits, u32 *hist) {
  HUMID_GUARD_LAST_VGPR();
  extern __shared__ u32 lh[];
  const u32 n_bins = 1u << bits;
  for (u32 b = threadIdx.x; b < n_bins; b += blockDim.x) lh[b] = 0;
  __syncthreads();
  const u32 stride = gridDim.x * blockDim.x;
  u32 r = blockIdx.x * blockDim.x + threadIdx.x;
  for (; r + 3 * stride < n_reads; r += 4 * stride) {         // four independent loads in flight
    u64 w[4];
    bool ok[4];
#pragma unroll
    for (u32 q = 0; q < 4; q++) {
      ok[q] = !(filtered && filtered[r + q * stride]);
      w[q] = words[r + q * stride];
    }
#pragma unroll
    for (u32 q = 0; q < 4; q++)
      if (ok[q]) atomicAdd(&lh[(u32)(((w[q] - lo) * scale) >> (64 - bits))], 1u);   // < n_bins by construction
  }
  for (; r < n_reads; r += stride)
    if (!(filtered && filtered[r])) atomicAdd(&lh[(u32)(((words[r] - lo) * scale) >> (64 - bits))], 1u);
  __syncthreads();
  for (u32 b = threadIdx.x; b < n_bins; b += blockDim.x)
    if (lh[b]) atomicAdd(&hist[b], lh[b]);
}

static __global__ void k_at_least_double(u64 a, u64 b, int *out) {
  HUMID_GUARD_LAST_VGPR(); *out = at_least_double(a, b) ? 1 : 0; }


#endif  // HUMID_KERNELS_MAP_HIP_H
