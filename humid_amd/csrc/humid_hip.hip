// humid_hip.hip -- HUMID's neighbour-search-and-cluster hot path for MI355X (gfx950).
//
// Pipeline (all device-side; integer/bit work, HBM/latency bound, no MFMA):
//   1. k_hash_insert      exact counts: open-address table of 2-bit-packed words in HBM
//                         (replaces Trie::add, call site /root/reference/src/humid.cc:95)
//   2. unique sort        radix sort of the U unique words -> Trie::walk() order
//   3. k_pairs            pigeonhole radix buckets (d+1 segments), nucleotide Hamming by
//                         popcount, both directions appended -> CSR with ascending lists
//                         (replaces walk x asymmetricHamming, src/humid.cc:113-130)
//   4. union-find CC      components of the neighbour graph (independent clustering units)
//   5. k_cluster          per component: the findClusters loop + src/cluster.cc, order-exact
//   6. ids + k_read_map   cluster ids in creator order; per read (cluster_id, keep)
//                         (replaces trie.find()->leaf->cluster, src/humid.cc:223-231,276-277)
//
// No CPU fallback lives here: every entry point either runs on the GPU or fails.
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_run_length_encode.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

#include "../../include/humid_hip.h"

typedef uint64_t u64;
typedef uint32_t u32;
typedef uint8_t u8;
typedef unsigned long long ull;

#define EMPTY_KEY 0xffffffffffffffffull
#define NOSLOT 0xffffffffu
#define NONE32 0xffffffffu

enum { CTR_UNIQUE = 0, CTR_USABLE, CTR_EDGES, CTR_NONSINGLE, CTR_MEMBERS, CTR_SPECIAL,
       CTR_CLUSTERS, CTR_OVERFULL, CTR_N = 16 };

// --------------------------------------------------------------------------------
// device helpers
// --------------------------------------------------------------------------------
__device__ __forceinline__ u64 mix64(u64 x) {
  x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull;
  x ^= x >> 27; x *= 0x94d049bb133111ebull;
  x ^= x >> 31;
  return x;
}

// inverse of mix64 (mix64 is a bijection on 64-bit words)
__host__ __device__ __forceinline__ u64 unmix64(u64 x) {
  x = (x ^ (x >> 31) ^ (x >> 62)) * 0x319642b2d24d8ec3ull;
  x = (x ^ (x >> 27) ^ (x >> 54)) * 0x96de1b173f119089ull;
  x = x ^ (x >> 30) ^ (x >> 60);
  return x;
}

// nucleotide (not bit) mismatches between two packed words
__device__ __forceinline__ u32 nt_mismatch(u64 x) {
  return (u32)__popcll((x | (x >> 1)) & 0x5555555555555555ull);
}

__device__ __forceinline__ u32 ld_agent(const u32 *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// /root/reference/src/cluster.cc:31-33 atLeastDouble_
__device__ __forceinline__ bool at_least_double(u64 a, u64 b) { return a >= 2 * b; }

// --------------------------------------------------------------------------------
// 1. exact counts: open-address hash of packed words
// --------------------------------------------------------------------------------
// One 16-byte slot per word so that the key probe and both atomics touch ONE line.
// The table is initialised by a plain 0xff memset: key = EMPTY, cnt = 0xffffffff (count-1,
// wraps to 0 on the first add), first = 0xffffffff (atomicMin identity).
// tab[cap+1]: slot `cap` is reserved for the word that equals EMPTY_KEY (n = 32, all T).
struct __attribute__((aligned(16))) Slot {
  u64 key;
  u32 cntm1;   // occurrences - 1; 0xffffffff = never touched
  u32 first;   // smallest read index with this word
};

__global__ void __launch_bounds__(256)
k_hash_insert(const u64 *__restrict__ words, const u8 *__restrict__ filtered, u32 n_reads,
              Slot *tab, u32 cap_log2, u32 *__restrict__ slot_of_read, u64 range_lo, u64 range_hi,
              u32 max_probe, ull *ctr) {
  const u32 mask = (1u << cap_log2) - 1u;
  const u32 cap = 1u << cap_log2;
  for (u32 r = blockIdx.x * blockDim.x + threadIdx.x; r < n_reads; r += gridDim.x * blockDim.x) {
    if (filtered[r]) { slot_of_read[r] = NOSLOT; continue; }
    const u64 w = words[r];
    if (w < range_lo || w > range_hi) { slot_of_read[r] = NOSLOT; continue; }   // another rank's word
    u32 s;
    if (w == EMPTY_KEY) {
      s = cap;
    } else {
      s = (u32)(mix64(w) >> (64 - cap_log2)) & mask;
      u32 probes = 0;
      while (true) {
        u64 k = tab[s].key;
        if (k == EMPTY_KEY) k = atomicCAS((ull *)&tab[s].key, EMPTY_KEY, (ull)w);
        if (k == EMPTY_KEY || k == w) break;
        s = (s + 1) & mask;
        if (++probes > max_probe) { s = NOSLOT; break; }   // table (nearly) full: never spin forever
      }
      if (s == NOSLOT) { ctr[CTR_OVERFULL] = 1; slot_of_read[r] = NOSLOT; continue; }
    }
    atomicAdd(&tab[s].cntm1, 1u);
    atomicMin(&tab[s].first, r);
    slot_of_read[r] = s;
  }
}

// A single-address global atomic costs ~12 ns and serialises (rocprof: 80 k of them = 1 ms), so
// compaction kernels run a FIXED small grid; each block owns a contiguous chunk, counts its
// items, reserves output space with ONE atomic, then writes in a second pass over the chunk
// (L2-resident by then).
#define COMPACT_BLOCKS 1024u

// block-wide sum of a per-thread value (256 threads); result valid in all threads
__device__ __forceinline__ u32 block_sum(u32 x, u32 *lds /* >= 4 u32 */) {
#pragma unroll
  for (u32 d = 32; d >= 1; d >>= 1) x += __shfl_xor(x, d);
  if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = x;
  __syncthreads();
  u32 t = lds[0] + lds[1] + lds[2] + lds[3];
  __syncthreads();
  return t;
}

// exclusive position of this thread's flag among the block's 256 flags; *total = block count
__device__ __forceinline__ u32 block_rank(bool flag, u32 *lds /* >= 4 u32 */, u32 *total) {
  const u32 lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const u64 m = __ballot(flag);
  if (lane == 0) lds[wv] = (u32)__popcll(m);
  __syncthreads();
  u32 before = 0;
  for (u32 k = 0; k < wv; k++) before += lds[k];
  *total = lds[0] + lds[1] + lds[2] + lds[3];
  __syncthreads();
  return before + (u32)__popcll(m & ((1ull << lane) - 1ull));
}

// occupied slots -> (word, slot) list in arbitrary order; also sums the usable reads
__global__ void __launch_bounds__(256)
k_compact_table(const Slot *__restrict__ tab, u32 n_slots, u64 *__restrict__ uniq_word,
                u32 *__restrict__ uniq_slot, u32 uniq_cap, ull *ctr) {
  __shared__ u32 lds[8];
  const u32 chunk = (n_slots + gridDim.x - 1) / gridDim.x;
  const u32 lo = blockIdx.x * chunk;
  const u32 hi = (lo + chunk < n_slots) ? lo + chunk : n_slots;
  u32 mine = 0, reads = 0;
  for (u32 sidx = lo + threadIdx.x; sidx < hi; sidx += 256) {
    const u32 c = tab[sidx].cntm1;
    if (c != NONE32) { mine++; reads += c + 1u; }
  }
  const u32 total = block_sum(mine, lds);
  const u32 total_reads = block_sum(reads, lds);
  if (threadIdx.x == 0) {
    lds[4] = total ? (u32)atomicAdd(&ctr[CTR_UNIQUE], (ull)total) : 0u;
    if (total_reads) atomicAdd(&ctr[CTR_USABLE], (ull)total_reads);
  }
  __syncthreads();
  u32 base = lds[4];
  for (u32 s0 = lo; s0 < hi; s0 += 256) {
    const u32 sidx = s0 + threadIdx.x;
    Slot sl;
    sl.cntm1 = NONE32;
    if (sidx < hi) sl = tab[sidx];
    u32 tot;
    const u32 r = block_rank(sl.cntm1 != NONE32, lds, &tot);
    if (sl.cntm1 != NONE32) {
      if (base + r < uniq_cap) {
        uniq_word[base + r] = sl.key;
        uniq_slot[base + r] = sidx;
      } else {
        ctr[CTR_OVERFULL] = 1;
      }
    }
    base += tot;
  }
}

// --------------------------------------------------------------------------------
// 1b. exact counts, partitioned: the reads are first bucketed by the top PB bits of mix64(word)
// (radix partition; mix64 is a bijection, so equal keys <=> equal words), then every bucket is
// counted by one workgroup in an LDS-resident open-address table.  No random HBM line traffic:
// the only scattered access left is the 4-byte slot_of_read[r] store.
// --------------------------------------------------------------------------------
#define LDS_SLOTS 2048u          // 16-byte entries: 32 KiB of LDS per workgroup, 5 workgroups per CU
#define LDS_FILL_LIMIT 1536u     // unique words a bucket may hold (75 % load)
#define PART_TARGET 700u         // mean reads per bucket

struct MixKeyOp {                // keys_input transform: word -> partition-ordered key
  __host__ __device__ u64 operator()(u64 w) const { return mix64(w); }
};
struct ReadTagOp {               // values_input transform: read index | excluded << 31
  const u64 *words;
  const u8 *filtered;
  u64 lo, hi;
  __device__ u32 operator()(u32 r) const {
    const u64 w = words[r];
    const bool excl = filtered[r] != 0 || w < lo || w > hi;
    return r | (excl ? 0x80000000u : 0u);
  }
};

// first position of every bucket in the partitioned key array (binary search)
__global__ void k_part_bounds(const u64 *__restrict__ keys, u32 n, u32 pb, u32 n_parts, u32 *__restrict__ pbeg,
                              u32 *__restrict__ ucount) {
  u32 p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p > n_parts) return;
  if (p == n_parts) { pbeg[p] = n; ucount[p] = 0; return; }   // ucount tail: scan sentinel
  const u64 target = (u64)p << (64 - pb);
  u32 lo = 0, hi = n;
  while (lo < hi) {
    u32 mid = lo + ((hi - lo) >> 1);
    if (keys[mid] < target) lo = mid + 1; else hi = mid;
  }
  pbeg[p] = lo;
}

// One workgroup per bucket.  Entry s of the LDS table: lkey (mixed word), lcnt (occurrences, 0 =
// empty), lfirst (smallest read index).  Entry LDS_SLOTS is reserved for the key that equals the
// EMPTY sentinel.  Outputs, in a PADDED layout (bucket b owns positions [pbeg[b], pbeg[b+1]) of
// N-sized arrays, its u unique words take the first u of them):
//   pad_word/pad_cnt/pad_first, ucount[b], pusable[b]; slot_of_read[r] = padded position.
__global__ void __launch_bounds__(256)
k_dedup_lds(const u64 *__restrict__ keys, const u32 *__restrict__ vals, const u32 *__restrict__ pbeg,
            u32 n_reads, u32 pb, u64 *__restrict__ pad_word, uint2 *__restrict__ pad_cf,
            u32 *__restrict__ ucount, u32 *__restrict__ pusable, u32 *__restrict__ pslot, ull *ctr) {
  __shared__ u64 lkey[LDS_SLOTS + 1];
  __shared__ u32 lcnt[LDS_SLOTS + 1];
  __shared__ u32 lfirst[LDS_SLOTS + 1];
  __shared__ u32 lds[8];
  const u32 b = blockIdx.x;
  const u32 beg = pbeg[b], end = pbeg[b + 1];
  if (beg >= end || end > n_reads) {
    if (beg > end || end > n_reads) ctr[CTR_OVERFULL] = 1;   // malformed partition: never index with it
    if (threadIdx.x == 0) { ucount[b] = 0; pusable[b] = 0; }
    return;
  }
  for (u32 s = threadIdx.x; s <= LDS_SLOTS; s += 256) { lkey[s] = EMPTY_KEY; lcnt[s] = 0; lfirst[s] = NONE32; }
  __syncthreads();
  const u32 hshift = 64 - pb - 11;      // table index = the 11 key bits below the bucket bits
  u32 usable = 0;
  bool overflow = false;
  for (u32 i = beg + threadIdx.x; i < end; i += 256) {
    const u32 v = vals[i];
    if ((v & 0x7fffffffu) >= n_reads) { overflow = true; break; }   // a malformed index is never used
    if (v & 0x80000000u) { pslot[i] = NOSLOT; continue; }
    usable++;
    const u64 k = keys[i];
    u32 s;
    if (k == EMPTY_KEY) {
      s = LDS_SLOTS;
    } else {
      s = (u32)(k >> hshift) & (LDS_SLOTS - 1);
      u32 probes = 0;
      while (true) {
        u64 cur = lkey[s];
        if (cur == EMPTY_KEY) cur = atomicCAS((ull *)&lkey[s], EMPTY_KEY, (ull)k);
        if (cur == EMPTY_KEY || cur == k) break;
        s = (s + 1) & (LDS_SLOTS - 1);
        if (++probes >= LDS_SLOTS) { overflow = true; break; }
      }
      if (overflow) break;
    }
    atomicAdd(&lcnt[s], 1u);
    atomicMin(&lfirst[s], v);
  }
  if (overflow) ctr[CTR_OVERFULL] = 1;
  __syncthreads();
  // compaction of the occupied entries -> padded arrays; lfirst[s] is then reused as slot -> index
  u32 base = 0;
  for (u32 s0 = 0; s0 <= LDS_SLOTS; s0 += 256) {
    const u32 s = s0 + threadIdx.x;
    const bool occ = (s <= LDS_SLOTS) && lcnt[s] != 0;
    u32 tot;
    const u32 r = block_rank(occ, lds, &tot);
    if (occ) {
      const u32 li = base + r;           // li < unique words <= reads of the bucket = padded room
      pad_word[beg + li] = unmix64(lkey[s]);
      pad_cf[beg + li] = make_uint2(lcnt[s], lfirst[s]);
      lfirst[s] = li;
    }
    base += tot;
  }
  if (threadIdx.x == 0) ucount[b] = base;
  const u32 tu = block_sum(usable, lds);
  if (threadIdx.x == 0) pusable[b] = tu;
  __syncthreads();
  // second pass: every position learns the padded slot of its word (coalesced store; the
  // per-read outputs are produced later in this same partition order, see k_read_map_part)
  for (u32 i = beg + threadIdx.x; i < end; i += 256) {
    const u32 v = vals[i];
    if (v >= n_reads) continue;          // excluded read (bit 31) or malformed index
    const u64 k = keys[i];
    u32 s;
    if (k == EMPTY_KEY) {
      s = LDS_SLOTS;
    } else {
      s = (u32)(k >> hshift) & (LDS_SLOTS - 1);
      u32 probes = 0;
      while (lkey[s] != k && probes++ < LDS_SLOTS) s = (s + 1) & (LDS_SLOTS - 1);
    }
    const u32 li = lfirst[s];
    pslot[i] = (li < end - beg) ? beg + li : NOSLOT;
  }
}

// totals over the buckets: U = sum ucount, usable = sum pusable (one block)
__global__ void __launch_bounds__(256)
k_part_totals(const u32 *__restrict__ ucount, const u32 *__restrict__ pusable, u32 n_parts, ull *ctr) {
  __shared__ u32 lds[4];
  ull u = 0, us = 0;
  for (u32 p = threadIdx.x; p < n_parts; p += 256) { u += ucount[p]; us += pusable[p]; }
  // 64-bit block sums via two 32-bit halves are unnecessary: both totals are < 2^32
  const u32 tu = block_sum((u32)u, lds);
  const u32 ts = block_sum((u32)us, lds);
  if (threadIdx.x == 0) { ctr[CTR_UNIQUE] = tu; ctr[CTR_USABLE] = ts; }
}

// padded -> dense unique list (word, padded position); order = bucket order (sorted afterwards)
__global__ void __launch_bounds__(256)
k_compact_padded(const u64 *__restrict__ pad_word, const u32 *__restrict__ pbeg, const u32 *__restrict__ ucount,
                 const u32 *__restrict__ ubase, u32 n_parts, u64 *__restrict__ uniq_word,
                 u32 *__restrict__ uniq_slot) {
  // one wave per bucket
  const u32 wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const u32 lane = threadIdx.x & 63;
  if (wave >= n_parts) return;
  const u32 beg = pbeg[wave], uc = ucount[wave], ub = ubase[wave];
  for (u32 j = lane; j < uc; j += 64) {
    uniq_word[ub + j] = pad_word[beg + j];
    uniq_slot[ub + j] = beg + j;
  }
}

// after the sort, padded variant: gather count / first read of rank i (one 8-byte gather)
__global__ void k_post_sort_padded(const u32 *__restrict__ s_slot, const uint2 *__restrict__ pad_cf, u32 n,
                                   u32 *__restrict__ s_cnt, u32 *__restrict__ s_first) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const uint2 cf = pad_cf[s_slot[i]];
    s_cnt[i] = cf.x;
    s_first[i] = cf.y;
  }
}

// after the sort: per rank i gather count / first read from the table
__global__ void k_post_sort(const u32 *__restrict__ s_slot, const Slot *__restrict__ tab, u32 n,
                            u32 *__restrict__ s_cnt, u32 *__restrict__ s_first) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const Slot sl = tab[s_slot[i]];
    s_cnt[i] = sl.cntm1 + 1u;
    s_first[i] = sl.first;
  }
}

// --------------------------------------------------------------------------------
// 3. neighbour search: pigeonhole segments
// --------------------------------------------------------------------------------
// Generalised pigeonhole: the n nucleotides are cut into s segments; two words within
// Hamming distance d agree exactly on at least s-d of them, so every pair is found in the bucket
// of some COMBINATION of s-d segments.  d=1: s=2, 2 combos of 12 nt (n=24).  d=2: s=4, 6 combos
// of 12 nt -- not 3 segments of 8 nt, whose 65 536 buckets hold hundreds of words each.
// Combo 0 is always the top s-d segments, i.e. a prefix: its buckets are runs of the sorted
// unique array and need no sort.  mask[c] = bits of combo c; a pair is emitted from the FIRST
// combo it agrees on.
#define MAX_COMBOS 20
#define MAX_FIELDS 8
struct ComboPlan {
  u32 ncombo;
  u32 key_bits;                       // bits of a combo key (sum of its field widths)
  u64 mask[MAX_COMBOS];
  u8 nfield[MAX_COMBOS];
  u8 shift[MAX_COMBOS][MAX_FIELDS];   // fields from most to least significant
  u8 width[MAX_COMBOS][MAX_FIELDS];
};

// bucket key of combo `cb` for every unique word (fields concatenated, most significant first)
// Kernel arguments derived from the plan are passed BY VALUE in small structs and indexed
// STATICALLY (unrolled loops with a predicate).  A 1 KB plan struct indexed dynamically in the
// kernarg segment -- and equally a plan freshly uploaded to device memory and read through
// wave-uniform (scalar) loads -- returned stale fields for single waves on gfx950 / ROCm 7.2
// (5-25 of 219 k edges lost at 10 M reads, tools/det_check.py), so neither form is used.
struct EarlierMasks {
  u64 m[MAX_COMBOS];
};

// fields of ONE combo
struct ComboFields {
  u32 nf;
  u8 shift[MAX_FIELDS];
  u8 width[MAX_FIELDS];
};

template <class KeyT>
__global__ void k_combo_keys(const u64 *__restrict__ s_word, u32 n, ComboFields cf,
                             KeyT *__restrict__ key, u32 *__restrict__ val) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const u64 w = s_word[i];
  u64 k = 0;
#pragma unroll
  for (u32 f = 0; f < MAX_FIELDS; f++) {
    if (f < cf.nf) {
      const u32 wd = cf.width[f];
      k = (k << wd) | ((w >> cf.shift[f]) & ((wd >= 64) ? ~0ull : ((1ull << wd) - 1ull)));
    }
  }
  key[i] = (KeyT)k;
  val[i] = i;
}

// --------------------------------------------------------------------------------
// 4. connected components (lock-free union-find, smaller index wins => root = min rank)
// --------------------------------------------------------------------------------
__device__ __forceinline__ u32 uf_find(const u32 *P, u32 x) {
  u32 p = ld_agent(&P[x]);
  while (p != x) { x = p; p = ld_agent(&P[x]); }
  return x;
}

__device__ __forceinline__ void uf_union(u32 *P, u32 a, u32 b) {
  while (true) {
    a = uf_find(P, a);
    b = uf_find(P, b);
    if (a == b) return;
    if (a > b) { u32 t = a; a = b; b = t; }
    if (atomicCAS(&P[b], b, a) == b) return;
  }
}

// One thread per position i of the bucket-sorted order (V = ranks in that order; combo 0 uses
// the sorted unique array itself); compares with the following elements of its bucket: the
// bucket ends at the first j whose word differs inside the combo mask.  Ranks ascend inside a
// bucket, so (ri < rj) always.  A pair is emitted only from the FIRST combo it agrees on.
// Two phases with identical control flow and no shared append counter:
//   FILL = false: deg[] += 1 per endpoint, union(ri, rj) in the component forest
//   FILL = true : writes rj into ri's CSR row and ri into rj's (per-row cursors; the rows are
//                 put in ascending order afterwards by k_sort_lists)
// MODE: what happens to a found pair
//   PM_COUNT      deg[] += 1 per endpoint, union(ri, rj)            (single-GPU phase A)
//   PM_FILL       both directions into the CSR rows via cursors      (single-GPU phase B)
//   PM_EMIT_COUNT pc[t] = pairs found by this thread                 (multi-GPU share, phase A)
//   PM_EMIT_FILL  edge (min << 32 | max) at poff[t] + k              (multi-GPU share, phase B)
// i0/n_i: the thread block covers positions [i0, i0 + n_i) as the first element of a pair; the
// second runs on to the end of the bucket anywhere in [0, n).
enum { PM_COUNT = 0, PM_FILL = 1, PM_EMIT_COUNT = 2, PM_EMIT_FILL = 3 };

template <bool PASS0, int MODE>
__global__ void __launch_bounds__(256)
k_pairs(const u64 *__restrict__ s_word, const u32 *__restrict__ V, u32 n, u32 i0, u32 n_i, u64 mask,
        EarlierMasks em, u32 cb, u32 distance, u32 *deg, u32 *parent,
        const u32 *__restrict__ nbr_off, u32 *cur, u32 *nbr_idx, u32 *__restrict__ pc,
        const u32 *__restrict__ poff, u64 *__restrict__ edges) {
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_i) return;
  const u32 i = i0 + t;
  const u32 ri = PASS0 ? i : V[i];
  const u64 wi = s_word[ri];
  u32 found = 0;
  u64 e = (MODE == PM_EMIT_FILL) ? (u64)poff[t] : 0;
  for (u32 j = i + 1; j < n; j++) {
    const u32 rj = PASS0 ? j : V[j];
    const u64 x = wi ^ s_word[rj];
    if (x & mask) break;                           // left the bucket
    if (nt_mismatch(x) > distance) continue;
    bool first = true;
#pragma unroll
    for (u32 q = 0; q < MAX_COMBOS; q++)
      first = first && !(q < cb && (x & em.m[q]) == 0);
    if (!first) continue;
    if (MODE == PM_FILL) {
      nbr_idx[nbr_off[ri] + atomicAdd(&cur[ri], 1u)] = rj;
      nbr_idx[nbr_off[rj] + atomicAdd(&cur[rj], 1u)] = ri;
    } else if (MODE == PM_COUNT) {
      found++;
      atomicAdd(&deg[rj], 1u);
      uf_union(parent, ri, rj);
    } else if (MODE == PM_EMIT_COUNT) {
      found++;
    } else {
      edges[e++] = ri < rj ? (((u64)ri << 32) | rj) : (((u64)rj << 32) | ri);
    }
  }
  if (MODE == PM_COUNT && found) atomicAdd(&deg[ri], found);
  if (MODE == PM_EMIT_COUNT) pc[t] = found;
}

// the same two phases driven by an explicit edge list (multi-GPU: the ranks' shares, all-gathered)
template <bool FILL>
__global__ void __launch_bounds__(256)
k_edges_apply(const u64 *__restrict__ edges, u64 n_edges, u32 n_nodes, u32 *deg, u32 *parent,
              const u32 *__restrict__ nbr_off, u32 *cur, u32 *nbr_idx, ull *ctr) {
  for (u64 k = (u64)blockIdx.x * blockDim.x + threadIdx.x; k < n_edges; k += (u64)gridDim.x * blockDim.x) {
    const u64 ed = edges[k];
    const u32 a = (u32)(ed >> 32), b = (u32)ed;
    if (a >= n_nodes || b >= n_nodes || a == b) { ctr[CTR_OVERFULL] = 1; continue; }   // malformed edge
    if (FILL) {
      nbr_idx[nbr_off[a] + atomicAdd(&cur[a], 1u)] = b;
      nbr_idx[nbr_off[b] + atomicAdd(&cur[b], 1u)] = a;
    } else {
      atomicAdd(&deg[a], 1u);
      atomicAdd(&deg[b], 1u);
      uf_union(parent, a, b);
    }
  }
}

// multi-GPU share of a sorted combo: the unique words whose combo key lies in [klo, khi]
// (key, rank) appended in arbitrary order; fixed grid, one global atomic per block
template <class KeyT>
__global__ void __launch_bounds__(256)
k_select_keyrange(const u64 *__restrict__ s_word, u32 n, ComboFields cf, u64 klo, u64 khi,
                  KeyT *__restrict__ key_out, u32 *__restrict__ val_out, ull *ctr) {
  __shared__ u32 lds[8];
  const u32 chunk = (n + gridDim.x - 1) / gridDim.x;
  const u32 lo = blockIdx.x * chunk;
  const u32 hi = (lo + chunk < n) ? lo + chunk : n;
  auto key_of = [&](u32 i) {
    const u64 w = s_word[i];
    u64 k = 0;
#pragma unroll
    for (u32 f = 0; f < MAX_FIELDS; f++) {
      if (f < cf.nf) {
        const u32 wd = cf.width[f];
        k = (k << wd) | ((w >> cf.shift[f]) & ((wd >= 64) ? ~0ull : ((1ull << wd) - 1ull)));
      }
    }
    return k;
  };
  u32 mine = 0;
  for (u32 i = lo + threadIdx.x; i < hi; i += 256) {
    const u64 k = key_of(i);
    mine += (k >= klo && k <= khi) ? 1u : 0u;
  }
  const u32 total = block_sum(mine, lds);
  if (threadIdx.x == 0) lds[4] = total ? (u32)atomicAdd(&ctr[CTR_SPECIAL], (ull)total) : 0u;
  __syncthreads();
  u32 base = lds[4];
  if (total == 0) return;
  for (u32 i0 = lo; i0 < hi; i0 += 256) {
    const u32 i = i0 + threadIdx.x;
    u64 k = 0;
    bool sel = false;
    if (i < hi) { k = key_of(i); sel = (k >= klo && k <= khi); }
    u32 tot;
    const u32 r = block_rank(sel, lds, &tot);
    if (sel) { key_out[base + r] = (KeyT)k; val_out[base + r] = i; }
    base += tot;
  }
}

// every CSR row ascending (the order NLeaf::neighbours has under the trie hypotheses H1+H2)
__global__ void __launch_bounds__(256)
k_sort_lists(const u32 *__restrict__ off, u32 n, u32 *idx) {
  u32 u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= n) return;
  const u32 b = off[u], d = off[u + 1] - b;
  if (d < 2) return;
  u32 *a = idx + b;
  if (d <= 32) {
    u32 v[32];
    for (u32 k = 0; k < d; k++) v[k] = a[k];
    for (u32 k = 1; k < d; k++) {          // insertion sort
      u32 x = v[k];
      u32 m = k;
      while (m > 0 && v[m - 1] > x) { v[m] = v[m - 1]; m--; }
      v[m] = x;
    }
    for (u32 k = 0; k < d; k++) a[k] = v[k];
  } else {                                 // heap sort in place
    for (u32 start = d / 2; start-- > 0;) {
      u32 r = start;
      while (true) {
        u32 ch = 2 * r + 1;
        if (ch >= d) break;
        if (ch + 1 < d && a[ch + 1] > a[ch]) ch++;
        if (a[r] >= a[ch]) break;
        u32 t = a[r]; a[r] = a[ch]; a[ch] = t;
        r = ch;
      }
    }
    for (u32 end = d - 1; end > 0; end--) {
      u32 t = a[0]; a[0] = a[end]; a[end] = t;
      u32 r = 0;
      while (true) {
        u32 ch = 2 * r + 1;
        if (ch >= end) break;
        if (ch + 1 < end && a[ch + 1] > a[ch]) ch++;
        if (a[r] >= a[ch]) break;
        u32 t2 = a[r]; a[r] = a[ch]; a[ch] = t2;
        r = ch;
      }
    }
  }
}

// --------------------------------------------------------------------------------
// 4. connected components: sizes, and the split small / big
// --------------------------------------------------------------------------------
#define SMALL_COMP 32u      // components up to this many leaves are clustered by one lane, in registers

// flatten the forest and count the leaves of every component at its root
__global__ void k_comp_stats(const u32 *__restrict__ deg, u32 *P, u32 n, u32 *csize) {
  u32 u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= n || deg[u] == 0) return;
  const u32 root = uf_find(P, u);
  P[u] = root;
  atomicAdd(&csize[root], 1u);
}

// M = leaves with >= 1 neighbour, Mbig = those in components larger than SMALL_COMP
__global__ void __launch_bounds__(256)
k_comp_count(const u32 *__restrict__ deg, const u32 *__restrict__ P, const u32 *__restrict__ csize, u32 n,
             ull *ctr) {
  __shared__ u32 lds[4];
  u32 m = 0, mb = 0;
  for (u32 u = blockIdx.x * blockDim.x + threadIdx.x; u < n; u += gridDim.x * blockDim.x) {
    if (deg[u]) {
      m++;
      if (csize[P[u]] > SMALL_COMP) mb++;              // P was flattened by k_comp_stats
    }
  }
  const u32 tm = block_sum(m, lds);
  const u32 tb = block_sum(mb, lds);
  if (threadIdx.x == 0) {
    if (tm) atomicAdd(&ctr[CTR_NONSINGLE], (ull)tm);
    if (tb) atomicAdd(&ctr[CTR_MEMBERS], (ull)tb);
  }
}

__global__ void k_iota(u32 *p, u32 n) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = i;
}

// explicit-graph entry point: union every CSR entry (u, nbr)
__global__ void k_union_csr(const u32 *__restrict__ off, const u32 *__restrict__ idx, u32 n, u32 *P) {
  u32 u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= n) return;
  for (u32 k = off[u]; k < off[u + 1]; k++)
    if (idx[k] != u) uf_union(P, u, idx[k]);
}

// members of the BIG components, keyed (root << 32 | rank); unordered, sorted afterwards
__global__ void __launch_bounds__(256)
k_member_keys(const u32 *__restrict__ deg, u32 *P, const u32 *__restrict__ csize, u32 n, u64 *mkeys,
              ull *ctr) {
  __shared__ u32 lds[8];
  const u32 chunk = (n + gridDim.x - 1) / gridDim.x;
  const u32 lo = blockIdx.x * chunk;
  const u32 hi = (lo + chunk < n) ? lo + chunk : n;
  u32 mine = 0;
  for (u32 u = lo + threadIdx.x; u < hi; u += 256)
    mine += (deg[u] && csize[uf_find(P, u)] > SMALL_COMP) ? 1u : 0u;
  const u32 total = block_sum(mine, lds);
  if (threadIdx.x == 0) lds[4] = total ? (u32)atomicAdd(&ctr[CTR_SPECIAL], (ull)total) : 0u;
  __syncthreads();
  u32 base = lds[4];
  if (total == 0) return;
  for (u32 u0 = lo; u0 < hi; u0 += 256) {
    const u32 u = u0 + threadIdx.x;
    u32 root = 0;
    bool mem = (u < hi) && deg[u] != 0;
    if (mem) { root = uf_find(P, u); mem = csize[root] > SMALL_COMP; }
    u32 tot;
    const u32 r = block_rank(mem, lds, &tot);
    if (mem) mkeys[base + r] = ((u64)root << 32) | u;
    base += tot;
  }
}

// --------------------------------------------------------------------------------
// 5. clustering
// --------------------------------------------------------------------------------
// singletons (no neighbours): the leaf creates its own cluster (src/humid.cc:179-187 with an
// empty neighbour list: maxNeighbour_ returns the leaf, cluster.cc:39-51)
__global__ void k_cluster_singletons(const u32 *__restrict__ deg, const u32 *__restrict__ cnt, u32 n,
                                     u32 *cl_of, u32 *maxleaf, u64 *cl_size) {
  u32 u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= n) return;
  if (deg[u] == 0) {
    cl_of[u] = u + 1;
    maxleaf[u] = u;
    cl_size[u] = cnt[u];
  } else {
    cl_of[u] = 0;
  }
}

// The findClusters loop over the leaves of ONE connected component, ascending.  Literal
// restatement of
//   findClusters loop            /root/reference/src/humid.cc:176-189  (members ascending)
//   maxNeighbour_                src/cluster.cc:39-51  (first qualifying neighbour, restart)
//   assignDirectionalCluster_    src/cluster.cc:58-69  (pre-order flood, explicit stack)
//   assignMaxCluster             src/cluster.cc:72-80
// A cluster is named by its creating leaf (cl_of = creator rank + 1); ids come later from a
// prefix sum over creators, which reproduces `id++` in walk order.  `st` holds 2 words per member.
template <bool MAXIMUM, class MemberAt>
__device__ __forceinline__ void cluster_one_component(MemberAt member_at, u32 n_members,
                                                      const u32 *__restrict__ cnt,
                                                      const u32 *__restrict__ off,
                                                      const u32 *__restrict__ idx, u32 *cl_of,
                                                      u32 *maxleaf, u64 *cl_size, u32 *st) {
  for (u32 m = 0; m < n_members; m++) {
    const u32 u = member_at(m);
    if (cl_of[u] != 0) continue;                  // src/humid.cc:179
    const u32 label = u + 1;                      // new Cluster, creator u
    u32 start = u;
    u32 best = u;
    u32 bestc = 0;
    if (!MAXIMUM) {
      // maxNeighbour_
      u32 leaf = u;
      u32 k = off[leaf], kend = off[leaf + 1];
      u64 lc = cnt[leaf];
      while (k < kend) {
        u32 nb = idx[k++];
        if (cl_of[nb] == 0 && at_least_double(cnt[nb], lc)) {
          leaf = nb; lc = cnt[leaf];
          k = off[leaf]; kend = off[leaf + 1];
        }
      }
      start = leaf;
      best = leaf;                                // updateMaxCount_ once, cluster.cc:85
    }
    u64 size = 0;
    u32 depth = 0;
    // assignLeaf_(start)
    cl_of[start] = label;
    size += cnt[start];
    if (MAXIMUM) { bestc = cnt[start]; best = start; }
    st[0] = start; st[1] = off[start]; depth = 1;
    while (depth) {
      const u32 cur = st[2 * (depth - 1)];
      u32 k = st[2 * (depth - 1) + 1];
      const u32 kend = off[cur + 1];
      const u64 cc = cnt[cur];
      bool descended = false;
      while (k < kend) {
        const u32 nb = idx[k++];
        if (cl_of[nb] != 0) continue;
        if (!MAXIMUM && !at_least_double(cc, cnt[nb])) continue;
        cl_of[nb] = label;
        const u32 nc = cnt[nb];
        size += nc;
        if (MAXIMUM && nc > bestc) { bestc = nc; best = nb; }   // updateMaxCount_ strict >
        st[2 * (depth - 1) + 1] = k;
        st[2 * depth] = nb; st[2 * depth + 1] = off[nb];
        depth++;
        descended = true;
        break;
      }
      if (!descended) depth--;
    }
    maxleaf[u] = best;
    cl_size[u] = size;
  }
}

// BIG components: one lane per component = the head of its run in the sorted member keys;
// stack in HBM scratch (2 words per member of the run).
template <bool MAXIMUM>
__global__ void __launch_bounds__(64)
k_cluster_components(const u64 *__restrict__ mkeys, u32 n_members, const u32 *__restrict__ cnt,
                     const u32 *__restrict__ off, const u32 *__restrict__ idx, u32 *cl_of,
                     u32 *maxleaf, u64 *cl_size, u32 *stk) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_members) return;
  const u32 root = (u32)(mkeys[i] >> 32);
  if (i > 0 && (u32)(mkeys[i - 1] >> 32) == root) return;   // not a component head
  u32 len = 1;
  while (i + len < n_members && (u32)(mkeys[i + len] >> 32) == root) len++;
  cluster_one_component<MAXIMUM>([&](u32 m) { return (u32)mkeys[i + m]; }, len, cnt, off, idx, cl_of,
                                 maxleaf, cl_size, stk + 2 * (u64)i);
}

// Components of exactly two leaves a < b (one centre + one satellite: the bulk of the non-trivial
// components on UMI data) have a closed form of the same loop; no private arrays, no scratch.
template <bool MAXIMUM>
__global__ void __launch_bounds__(256)
k_cluster_pairs(const u32 *__restrict__ deg, const u32 *__restrict__ P, const u32 *__restrict__ csize, u32 n,
                const u32 *__restrict__ cnt, const u32 *__restrict__ off, const u32 *__restrict__ idx,
                u32 *cl_of, u32 *maxleaf, u64 *cl_size) {
  u32 a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= n || deg[a] == 0 || P[a] != a || csize[a] != 2) return;
  const u32 b = idx[off[a]];
  const u64 ca = cnt[a], cb = cnt[b];
  if (MAXIMUM) {                                   // whole component, maxLeaf = first strict maximum
    cl_of[a] = a + 1; cl_of[b] = a + 1;
    maxleaf[a] = (cb > ca) ? b : a;
    cl_size[a] = ca + cb;
    return;
  }
  if (at_least_double(cb, ca)) {                   // a climbs to b, b floods back to a
    cl_of[a] = a + 1; cl_of[b] = a + 1;
    maxleaf[a] = b;
    cl_size[a] = ca + cb;
  } else if (at_least_double(ca, cb)) {            // a stays, absorbs b
    cl_of[a] = a + 1; cl_of[b] = a + 1;
    maxleaf[a] = a;
    cl_size[a] = ca + cb;
  } else {                                         // two clusters; b finds a already assigned
    cl_of[a] = a + 1; maxleaf[a] = a; cl_size[a] = ca;
    cl_of[b] = b + 1; maxleaf[b] = b; cl_size[b] = cb;
  }
}

// SMALL components (<= SMALL_COMP leaves): one lane per component root collects the members by
// a breadth-first walk, orders them, and runs the same loop with member list and stack in
// private memory.  No sort, no scratch.
template <bool MAXIMUM>
__global__ void __launch_bounds__(128)
k_cluster_small(const u32 *__restrict__ deg, const u32 *__restrict__ P, const u32 *__restrict__ csize,
                u32 n, const u32 *__restrict__ cnt, const u32 *__restrict__ off,
                const u32 *__restrict__ idx, u32 *cl_of, u32 *maxleaf, u64 *cl_size) {
  u32 u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= n || deg[u] == 0 || P[u] != u) return;
  const u32 target = csize[u];
  if (target > SMALL_COMP || target == 2) return;   // pairs: k_cluster_pairs; big: k_cluster_components
  u32 mem[SMALL_COMP];
  u32 st[2 * SMALL_COMP];
  u32 nm = 1;
  mem[0] = u;
  for (u32 q = 0; q < nm && nm < target; q++) {
    const u32 v = mem[q];
    for (u32 k = off[v]; k < off[v + 1] && nm < target; k++) {
      const u32 nb = idx[k];
      bool seen = false;
      for (u32 t = 0; t < nm; t++) seen |= (mem[t] == nb);
      if (!seen) mem[nm++] = nb;
    }
  }
  for (u32 k = 1; k < nm; k++) {              // ascending = walk order inside the component
    u32 x = mem[k];
    u32 m = k;
    while (m > 0 && mem[m - 1] > x) { mem[m] = mem[m - 1]; m--; }
    mem[m] = x;
  }
  cluster_one_component<MAXIMUM>([&](u32 m) { return mem[m]; }, nm, cnt, off, idx, cl_of, maxleaf, cl_size, st);
}

__global__ void k_creator_flags(const u32 *__restrict__ cl_of, u32 n, u32 *flag) {
  u32 u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u < n) flag[u] = (cl_of[u] == u + 1) ? 1u : 0u;
}

// per node: final cluster id (creators numbered in walk order) and maxLeaf flag
__global__ void k_finalize_nodes(const u32 *__restrict__ cl_of, const u32 *__restrict__ pos,
                                 const u32 *__restrict__ maxleaf, u32 n, u32 *__restrict__ cid,
                                 u8 *__restrict__ ismax) {
  u32 u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= n) return;
  const u32 creator = cl_of[u] - 1;
  cid[u] = pos[creator] + 1;
  ismax[u] = (maxleaf[creator] == u) ? 1 : 0;
}

// per hash slot: (cluster id, read to keep) of the word it holds
__global__ void k_slot_results(const u32 *__restrict__ l_cid, const u8 *__restrict__ l_ismax,
                               const u32 *__restrict__ s_first, const u32 *__restrict__ s_slot, u32 n,
                               u64 *__restrict__ slot_out) {
  u32 u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= n) return;
  slot_out[s_slot[u]] = ((u64)(l_ismax[u] ? s_first[u] : NONE32) << 32) | l_cid[u];
}

__global__ void k_export_clusters(const u32 *__restrict__ flag, const u32 *__restrict__ pos,
                                  const u32 *__restrict__ maxleaf, const u64 *__restrict__ cl_size,
                                  const u32 *__restrict__ cnt, u32 n, u64 *o_size, u32 *o_maxcount,
                                  u32 *o_maxleaf) {
  u32 u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= n || !flag[u]) return;
  const u32 c = pos[u];
  if (o_size) o_size[c] = cl_size[u];
  if (o_maxleaf) o_maxleaf[c] = maxleaf[u];
  if (o_maxcount) o_maxcount[c] = cnt[maxleaf[u]];
}

// --------------------------------------------------------------------------------
// 6. per-read map: cluster id and the duplicate flag
// --------------------------------------------------------------------------------
// keep = this read is the first (input order) whose word is its cluster's maxLeaf
// (/root/reference/src/humid.cc:224-231); cluster 0 for filtered reads (:272).
__global__ void __launch_bounds__(256)
k_read_map(const u32 *__restrict__ slot_of_read, const u64 *__restrict__ slot_out, u32 n_reads,
           u32 *__restrict__ cluster_id, u8 *__restrict__ keep) {
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
__global__ void __launch_bounds__(256)
k_read_map_part(const u32 *__restrict__ vals, const u32 *__restrict__ pslot, const u64 *__restrict__ slot_out,
                u32 n_reads, u32 *__restrict__ packed) {
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

__global__ void __launch_bounds__(256)
k_split_out(const u32 *__restrict__ packed, u32 n_reads, u32 *__restrict__ cluster_id, u8 *__restrict__ keep) {
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

struct OwnedFlagOp {            // 1 for the reads this rank counted (global-table variant)
  const u32 *slot_of_read;
  u32 n;
  __device__ u32 operator()(u32 i) const { return (i < n && slot_of_read[i] != NOSLOT) ? 1u : 0u; }
};

// packed result (cluster id | keep << 31) of every owned read, dense, in read order
__global__ void __launch_bounds__(256)
k_owned_results(const u32 *__restrict__ slot_of_read, const u32 *__restrict__ opos,
                const u64 *__restrict__ slot_out, u32 n_reads, u32 *__restrict__ packed) {
  for (u32 r = blockIdx.x * blockDim.x + threadIdx.x; r < n_reads; r += gridDim.x * blockDim.x) {
    const u32 s = slot_of_read[r];
    if (s == NOSLOT) continue;
    const u64 o = slot_out[s];
    packed[opos[r]] = (u32)o | (((u32)(o >> 32) == r) ? 0x80000000u : 0u);
  }
}

// owner rank of every local read (n_ranks = nobody: filtered reads)
__global__ void __launch_bounds__(256)
k_owner_of(const u64 *__restrict__ words, const u8 *__restrict__ filtered, u32 n, OwnerRanges rg,
           u32 n_ranks, u8 *__restrict__ owner) {
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
__global__ void k_owner_bounds(const u8 *__restrict__ sorted_owner, u32 n, u32 n_ranks, u32 *__restrict__ bounds) {
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
__global__ void __launch_bounds__(256)
k_scatter_results(const u32 *__restrict__ perm, const u32 *__restrict__ packed, u32 n_recv,
                  u32 *__restrict__ cluster_id, u8 *__restrict__ keep) {
  for (u32 k = blockIdx.x * blockDim.x + threadIdx.x; k < n_recv; k += gridDim.x * blockDim.x) {
    const u32 r = perm[k];
    const u32 t = packed[k];
    cluster_id[r] = t & 0x7fffffffu;
    keep[r] = (u8)(t >> 31);
  }
}

__global__ void k_widen32(const u32 *__restrict__ in, u32 n, u64 *__restrict__ out) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i];
}

__global__ void k_creator_sizes(const u32 *__restrict__ flag, const u32 *__restrict__ pos,
                                const u64 *__restrict__ cl_size, u32 n, u64 *__restrict__ out) {
  u32 u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u < n && flag[u]) out[pos[u]] = cl_size[u];
}

// reads per top-`bits` bin of the word (usable reads only): balanced range splitters for the
// multi-GPU path.  LDS-privatised, fixed grid.
__global__ void __launch_bounds__(256)
k_top_hist(const u64 *__restrict__ words, const u8 *__restrict__ filtered, u32 n_reads, u32 shift,
           u32 n_bins, u32 *hist) {
  extern __shared__ u32 lh[];
  for (u32 b = threadIdx.x; b < n_bins; b += blockDim.x) lh[b] = 0;
  __syncthreads();
  for (u32 r = blockIdx.x * blockDim.x + threadIdx.x; r < n_reads; r += gridDim.x * blockDim.x)
    if (!filtered[r]) {
      u32 b = (u32)(words[r] >> shift);
      atomicAdd(&lh[b < n_bins ? b : n_bins - 1], 1u);   // malformed words cannot index out of LDS
    }
  __syncthreads();
  for (u32 b = threadIdx.x; b < n_bins; b += blockDim.x)
    if (lh[b]) atomicAdd(&hist[b], lh[b]);
}

__global__ void k_at_least_double(u64 a, u64 b, int *out) { *out = at_least_double(a, b) ? 1 : 0; }

// --------------------------------------------------------------------------------
// host side
// --------------------------------------------------------------------------------
struct DBuf {
  void *p = nullptr;
  size_t cap = 0;
  hipError_t ensure(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
    size_t want = bytes + bytes / 8 + 256;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) { p = nullptr; return e; }
    cap = want;
    return hipSuccess;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
  template <class T> T *as() const { return (T *)p; }
};

struct humid_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  std::string err;
  ull *d_ctr = nullptr;
  ull *h_ctr = nullptr;   // pinned mirror
  DBuf in_words, in_filt, out_cid, out_keep;                 // host entry point staging
  DBuf table, slot_out, slot_of_read, uniq_slot;             // table (cap+1) and per-read
  DBuf pk_keys, pk_vals, pbeg, ucount, pusable, ubase, pad_word, pad_cf, pslot;   // partitioned counts
  DBuf opos, own_packed, owner, owner_sorted, perm, small;                        // multi-GPU result return
  DBuf pc, poff, share_edges;                                                     // multi-GPU pair-search share
  int count_mode = 0;        // 0: hash-partitioned LDS tables (default), 1: one global HBM table
  u32 force_segments = 0;    // 0: automatic pigeonhole plan; else the number of segments s
  bool last_count_lds = false;
  DBuf uniq_word, s_word, s_slot, s_cnt, s_first;            // unique words (walk order)
  DBuf deg, nbr_off, nbr_idx, seg_k0, seg_v0, seg_ks, seg_vs, csize, cur, plan_dev;
  ComboPlan h_plan;          // host copy of the plan in flight (source of the async upload)
  DBuf parent, mk0, mk1, cl_of, maxleaf, cl_size, flag, pos, cid, ismax, stk, tmp, scratch;
  hipEvent_t ev[6] = {};
  hipEvent_t kev[40] = {};   // per-kernel timing: [0,1] insert, [2,3] cluster, [4..19] pairs fill, [20..35] pairs count
  bool have_run = false;     // a full dedup run completed (all accessors valid)
  bool have_graph = false;   // stage B completed (leaf/adjacency/cluster accessors valid)
  bool graph_mode = false;   // last call was humid_cluster_graph
  const u64 *g_word = nullptr;   // arrays stage B ran on
  const u32 *g_cnt = nullptr;
  u32 gU = 0;
  u32 cap_log2 = 0;
  u64 N = 0, U = 0, E = 0, M = 0, C = 0, usable = 0;
  u32 word_nt = 0, distance = 0, method = 0;
};

static std::string g_err;

static int fail(humid_ctx *c, int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (c) c->err = buf; else g_err = buf;
  return code;
}

#define HIPCHK(expr)                                                                        \
  do {                                                                                      \
    hipError_t _e = (expr);                                                                 \
    if (_e != hipSuccess)                                                                   \
      return fail(c, _e == hipErrorOutOfMemory ? HUMID_E_NOMEM : HUMID_E_HIP, "%s: %s (%s:%d)", \
                  #expr, hipGetErrorString(_e), __FILE__, __LINE__);                        \
  } while (0)

#define ENSURE(buf, bytes) HIPCHK((buf).ensure(bytes))

static inline u32 blocks_for(u64 n, u32 bs = 256) { return (u32)((n + bs - 1) / bs); }
static inline u32 grid_stride_blocks(u64 n, u32 bs = 256) {
  u64 b = (n + bs - 1) / bs;
  if (b > 256 * 8) b = 256 * 8;
  if (b == 0) b = 1;
  return (u32)b;
}
static inline u32 bits_for(u64 n) {  // bits needed to represent values < n
  u32 b = 0;
  while (b < 64 && ((u64)1 << b) < n) b++;
  return b ? b : 1;
}

// ---- rocPRIM wrappers (temp storage grown on demand) ---------------------------------
template <class K, class V>
static int sort_pairs(humid_ctx *c, const K *kin, K *kout, const V *vin, V *vout, u64 n, u32 b0, u32 b1) {
  size_t bytes = 0;
  HIPCHK(rocprim::radix_sort_pairs(nullptr, bytes, kin, kout, vin, vout, (size_t)n, b0, b1, c->stream));
  ENSURE(c->tmp, bytes);
  HIPCHK(rocprim::radix_sort_pairs(c->tmp.p, bytes, kin, kout, vin, vout, (size_t)n, b0, b1, c->stream));
  return HUMID_OK;
}
template <class K>
static int sort_keys(humid_ctx *c, const K *kin, K *kout, u64 n, u32 b0, u32 b1) {
  size_t bytes = 0;
  HIPCHK(rocprim::radix_sort_keys(nullptr, bytes, kin, kout, (size_t)n, b0, b1, c->stream));
  ENSURE(c->tmp, bytes);
  HIPCHK(rocprim::radix_sort_keys(c->tmp.p, bytes, kin, kout, (size_t)n, b0, b1, c->stream));
  return HUMID_OK;
}
static int exscan_u32(humid_ctx *c, const u32 *in, u32 *out, u64 n) {
  size_t bytes = 0;
  HIPCHK(rocprim::exclusive_scan(nullptr, bytes, in, out, 0u, (size_t)n, rocprim::plus<u32>(), c->stream));
  ENSURE(c->tmp, bytes);
  HIPCHK(rocprim::exclusive_scan(c->tmp.p, bytes, in, out, 0u, (size_t)n, rocprim::plus<u32>(), c->stream));
  return HUMID_OK;
}

// device counters -> pinned mirror, one stream sync.  extra32 (device u32, may be null) lands
// in h_ctr[CTR_N - 1].
static int read_counters(humid_ctx *c, const u32 *extra32 = nullptr) {
  HIPCHK(hipMemcpyAsync(c->h_ctr, c->d_ctr, CTR_N * sizeof(ull), hipMemcpyDeviceToHost, c->stream));
  if (extra32)
    HIPCHK(hipMemcpyAsync(&c->h_ctr[CTR_N - 1], extra32, 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return HUMID_OK;
}

#define TRY(...) do { int _rc = (__VA_ARGS__); if (_rc != HUMID_OK) return _rc; } while (0)

// Plan of the generalised pigeonhole search (see ComboPlan).  s is chosen so that combo keys are
// long enough for buckets to be small at this U (>= ~log4(U) nucleotides) without exceeding
// MAX_COMBOS combinations; d >= n degenerates to one empty-mask combo (every pair compared).
static u64 n_choose_k(u32 n, u32 k) {
  if (k > n) return 0;
  u64 r = 1;
  for (u32 i = 1; i <= k; i++) r = r * (n - k + i) / i;
  return r;
}

static ComboPlan make_plan(u32 n, u32 d, u64 U, u32 force_segments) {
  ComboPlan p;
  memset(&p, 0, sizeof p);
  if (d >= n) { p.ncombo = 1; p.key_bits = 0; p.mask[0] = 0; p.nfield[0] = 0; return p; }
  u32 want = 1;                                  // nucleotides of key wanted: 4^want >= U
  while (want < n && ((u64)1 << (2 * want)) < U) want++;
  u32 best_s = d + 1, best_len = 0;
  u64 best_c = ~0ull;
  for (u32 sgm = d + 1; sgm <= n && sgm <= d + MAX_FIELDS; sgm++) {
    const u64 combos = n_choose_k(sgm, sgm - d);
    if (combos > MAX_COMBOS) break;
    if (force_segments) {                         // test hook: take exactly this s if it is legal
      if (sgm == force_segments) { best_s = sgm; best_len = (sgm - d) * (n / sgm); best_c = combos; break; }
      continue;
    }
    const u32 len = (sgm - d) * (n / sgm);       // guaranteed key length (short segments)
    const bool better = (best_len < want) ? (len > best_len) : (len >= want && combos < best_c);
    if (best_len == 0 || better) { best_s = sgm; best_len = len; best_c = combos; }
    if (best_len >= want) break;                 // smallest s that reaches the wanted length
  }
  const u32 sgm = best_s, k = sgm - d;
  u32 seg_shift[64], seg_width[64];
  {
    u32 base = n / sgm, rem = n % sgm, pos = 0;
    for (u32 t = 0; t < sgm; t++) {
      u32 len = base + (t < rem ? 1 : 0);
      seg_shift[t] = 2 * (n - pos - len);
      seg_width[t] = 2 * len;
      pos += len;
    }
  }
  // combinations of k segments in lexicographic order: the first is {0..k-1}, a prefix
  u32 idx[64];
  for (u32 t = 0; t < k; t++) idx[t] = t;
  u32 c = 0, maxbits = 0;
  while (true) {
    u64 m = 0;
    u32 bits = 0;
    for (u32 t = 0; t < k; t++) {
      const u32 sg = idx[t];
      p.shift[c][t] = (u8)seg_shift[sg];
      p.width[c][t] = (u8)seg_width[sg];
      m |= ((seg_width[sg] >= 64) ? ~0ull : (((u64)1 << seg_width[sg]) - 1)) << seg_shift[sg];
      bits += seg_width[sg];
    }
    p.mask[c] = m;
    p.nfield[c] = (u8)k;
    if (bits > maxbits) maxbits = bits;
    c++;
    int t = (int)k - 1;
    while (t >= 0 && idx[t] == sgm - k + (u32)t) t--;
    if (t < 0) break;
    idx[t]++;
    for (u32 q = (u32)t + 1; q < k; q++) idx[q] = idx[q - 1] + 1;
  }
  p.ncombo = c;
  p.key_bits = maxbits;
  return p;
}

// ---- cluster stage shared by the full pipeline and the explicit-graph entry point ------
// needs: g_cnt[U], deg[U], nbr_off[U+1], nbr_idx, parent[U] + csize[U] (k_comp_stats done),
// M = leaves with deg > 0, Mbig = those in components larger than SMALL_COMP.
static int cluster_stage(humid_ctx *c, const u32 *g_cnt, u32 U, u64 M, u64 Mbig, u32 method) {
  hipStream_t st = c->stream;
  ENSURE(c->cl_of, (size_t)U * 4);
  ENSURE(c->maxleaf, (size_t)U * 4);
  ENSURE(c->cl_size, (size_t)U * 8);
  ENSURE(c->flag, (size_t)U * 4);
  ENSURE(c->pos, (size_t)(U + 1) * 4);
  ENSURE(c->cid, (size_t)U * 4);
  ENSURE(c->ismax, (size_t)U);
  hipLaunchKernelGGL(k_cluster_singletons, dim3(blocks_for(U)), dim3(256), 0, st, c->deg.as<u32>(),
                     g_cnt, U, c->cl_of.as<u32>(), c->maxleaf.as<u32>(), c->cl_size.as<u64>());
  if (M > 0) {
    HIPCHK(hipEventRecord(c->kev[2], st));
    if (method == HUMID_METHOD_MAXIMUM)
      hipLaunchKernelGGL(k_cluster_pairs<true>, dim3(blocks_for(U)), dim3(256), 0, st, c->deg.as<u32>(),
                         c->parent.as<u32>(), c->csize.as<u32>(), U, g_cnt, c->nbr_off.as<u32>(),
                         c->nbr_idx.as<u32>(), c->cl_of.as<u32>(), c->maxleaf.as<u32>(), c->cl_size.as<u64>());
    else
      hipLaunchKernelGGL(k_cluster_pairs<false>, dim3(blocks_for(U)), dim3(256), 0, st, c->deg.as<u32>(),
                         c->parent.as<u32>(), c->csize.as<u32>(), U, g_cnt, c->nbr_off.as<u32>(),
                         c->nbr_idx.as<u32>(), c->cl_of.as<u32>(), c->maxleaf.as<u32>(), c->cl_size.as<u64>());
    if (method == HUMID_METHOD_MAXIMUM)
      hipLaunchKernelGGL(k_cluster_small<true>, dim3(blocks_for(U, 128)), dim3(128), 0, st, c->deg.as<u32>(),
                         c->parent.as<u32>(), c->csize.as<u32>(), U, g_cnt, c->nbr_off.as<u32>(),
                         c->nbr_idx.as<u32>(), c->cl_of.as<u32>(), c->maxleaf.as<u32>(), c->cl_size.as<u64>());
    else
      hipLaunchKernelGGL(k_cluster_small<false>, dim3(blocks_for(U, 128)), dim3(128), 0, st, c->deg.as<u32>(),
                         c->parent.as<u32>(), c->csize.as<u32>(), U, g_cnt, c->nbr_off.as<u32>(),
                         c->nbr_idx.as<u32>(), c->cl_of.as<u32>(), c->maxleaf.as<u32>(), c->cl_size.as<u64>());
    if (Mbig > 0) {
      ENSURE(c->mk0, (size_t)Mbig * 8);
      ENSURE(c->mk1, (size_t)Mbig * 8);
      ENSURE(c->stk, (size_t)Mbig * 8);
      HIPCHK(hipMemsetAsync(&c->d_ctr[CTR_SPECIAL], 0, sizeof(ull), st));
      hipLaunchKernelGGL(k_member_keys, dim3(COMPACT_BLOCKS), dim3(256), 0, st, c->deg.as<u32>(),
                         c->parent.as<u32>(), c->csize.as<u32>(), U, c->mk0.as<u64>(), c->d_ctr);
      TRY(sort_keys<u64>(c, c->mk0.as<u64>(), c->mk1.as<u64>(), Mbig, 0, 32 + bits_for(U)));
      if (method == HUMID_METHOD_MAXIMUM)
        hipLaunchKernelGGL(k_cluster_components<true>, dim3(blocks_for(Mbig, 64)), dim3(64), 0, st,
                           c->mk1.as<u64>(), (u32)Mbig, g_cnt, c->nbr_off.as<u32>(),
                           c->nbr_idx.as<u32>(), c->cl_of.as<u32>(), c->maxleaf.as<u32>(),
                           c->cl_size.as<u64>(), c->stk.as<u32>());
      else
        hipLaunchKernelGGL(k_cluster_components<false>, dim3(blocks_for(Mbig, 64)), dim3(64), 0, st,
                           c->mk1.as<u64>(), (u32)Mbig, g_cnt, c->nbr_off.as<u32>(),
                           c->nbr_idx.as<u32>(), c->cl_of.as<u32>(), c->maxleaf.as<u32>(),
                           c->cl_size.as<u64>(), c->stk.as<u32>());
    }
    HIPCHK(hipEventRecord(c->kev[3], st));
  }
  hipLaunchKernelGGL(k_creator_flags, dim3(blocks_for(U)), dim3(256), 0, st, c->cl_of.as<u32>(), U,
                     c->flag.as<u32>());
  TRY(exscan_u32(c, c->flag.as<u32>(), c->pos.as<u32>(), U));
  HIPCHK(hipGetLastError());
  return HUMID_OK;
}

static int n_clusters_from_scan(humid_ctx *c, u32 U, u64 *out) {
  u32 last_pos = 0, last_flag = 0;
  HIPCHK(hipMemcpyAsync(&last_pos, c->pos.as<u32>() + (U - 1), 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipMemcpyAsync(&last_flag, c->flag.as<u32>() + (U - 1), 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  *out = (u64)last_pos + last_flag;
  return HUMID_OK;
}

// ---- stage A: exact counts + walk order ------------------------------------------------
// Inserts the reads whose word lies in [range_lo, range_hi] (inclusive; the multi-GPU path
// gives every rank one range, a single GPU takes everything), compacts the table and sorts
// the unique words.  Leaves table/slot_of_read/s_word/s_slot/s_cnt/s_first in the context.
static int stage_count_global(humid_ctx *c, const u64 *d_words, const u8 *d_filt, u32 N, u32 word_nt,
                              u64 range_lo, u64 range_hi, u64 expected_reads, humid_summary &s) {
  hipStream_t st = c->stream;
  c->last_count_lds = false;
  if (expected_reads == 0 || expected_reads > N) expected_reads = N;
  u32 cap_log2 = 10;
  while (((u64)1 << cap_log2) < expected_reads + expected_reads / 2) cap_log2++;
  const u64 cap = (u64)1 << cap_log2;
  c->cap_log2 = cap_log2;
  ENSURE(c->table, (cap + 1) * sizeof(Slot));
  ENSURE(c->slot_out, (cap + 1) * 8);
  ENSURE(c->slot_of_read, (size_t)N * 4);
  ENSURE(c->uniq_slot, (size_t)expected_reads * 4 + 4);
  ENSURE(c->uniq_word, (size_t)expected_reads * 8 + 8);
  HIPCHK(hipEventRecord(c->ev[0], st));
  HIPCHK(hipMemsetAsync(c->d_ctr, 0, CTR_N * sizeof(ull), st));
  HIPCHK(hipMemsetAsync(c->table.p, 0xff, (cap + 1) * sizeof(Slot), st));
  HIPCHK(hipEventRecord(c->kev[0], st));
  hipLaunchKernelGGL(k_hash_insert, dim3(grid_stride_blocks(N)), dim3(256), 0, st, d_words, d_filt, N,
                     c->table.as<Slot>(), cap_log2, c->slot_of_read.as<u32>(), range_lo, range_hi,
                     (u32)(cap - cap / 8), c->d_ctr);
  HIPCHK(hipEventRecord(c->kev[1], st));
  hipLaunchKernelGGL(k_compact_table, dim3(COMPACT_BLOCKS), dim3(256), 0, st,
                     c->table.as<Slot>(), (u32)(cap + 1), c->uniq_word.as<u64>(), c->uniq_slot.as<u32>(),
                     (u32)expected_reads, c->d_ctr);
  HIPCHK(hipGetLastError());
  TRY(read_counters(c));
  if (c->h_ctr[CTR_OVERFULL])
    return fail(c, HUMID_E_INVALID, "hash table over-full: more reads fell into this range than expected_reads");
  const u32 U = (u32)c->h_ctr[CTR_UNIQUE];
  s.usable = c->usable = c->h_ctr[CTR_USABLE];
  s.unique = c->U = U;
  if (U == 0) { HIPCHK(hipEventRecord(c->ev[1], st)); return HUMID_OK; }
  ENSURE(c->s_word, (size_t)U * 8);
  ENSURE(c->s_slot, (size_t)U * 4);
  ENSURE(c->s_cnt, (size_t)U * 4);
  ENSURE(c->s_first, (size_t)U * 4);
  TRY(sort_pairs<u64, u32>(c, c->uniq_word.as<u64>(), c->s_word.as<u64>(), c->uniq_slot.as<u32>(),
                           c->s_slot.as<u32>(), U, 0, 2 * word_nt));
  hipLaunchKernelGGL(k_post_sort, dim3(blocks_for(U)), dim3(256), 0, st, c->s_slot.as<u32>(),
                     c->table.as<Slot>(), U, c->s_cnt.as<u32>(), c->s_first.as<u32>());
  HIPCHK(hipEventRecord(c->ev[1], st));
  HIPCHK(hipGetLastError());
  return HUMID_OK;
}

// Partitioned variant of stage A (see section 1b of the kernels).  Returns HUMID_OK with
// *overflowed = true when a bucket held more unique words than its LDS table (the caller then
// runs the global-table variant; results are never taken from an overflowed run).
static int stage_count_lds(humid_ctx *c, const u64 *d_words, const u8 *d_filt, u32 N, u32 word_nt,
                           u64 range_lo, u64 range_hi, humid_summary &s, bool *overflowed) {
  hipStream_t st = c->stream;
  *overflowed = false;
  c->last_count_lds = true;
  u32 pb = 1;
  while (pb < 26 && ((u64)PART_TARGET << pb) < (u64)N) pb++;
  const u32 n_parts = 1u << pb;
  ENSURE(c->pk_keys, (size_t)N * 8);
  ENSURE(c->pk_vals, (size_t)N * 4);
  ENSURE(c->pbeg, (size_t)(n_parts + 1) * 4);
  ENSURE(c->ucount, (size_t)(n_parts + 1) * 4);
  ENSURE(c->pusable, (size_t)(n_parts + 1) * 4);
  ENSURE(c->ubase, (size_t)(n_parts + 1) * 4);
  ENSURE(c->pad_word, (size_t)N * 8);
  ENSURE(c->pad_cf, (size_t)N * 8);
  ENSURE(c->pslot, (size_t)N * 4);
  ENSURE(c->slot_out, ((size_t)N + 1) * 8);
  ENSURE(c->uniq_slot, (size_t)N * 4 + 4);
  ENSURE(c->uniq_word, (size_t)N * 8 + 8);
  HIPCHK(hipEventRecord(c->ev[0], st));
  HIPCHK(hipMemsetAsync(c->d_ctr, 0, CTR_N * sizeof(ull), st));
  {
    auto kin = rocprim::make_transform_iterator(d_words, MixKeyOp{});
    auto vin = rocprim::make_transform_iterator(rocprim::counting_iterator<u32>(0),
                                                ReadTagOp{d_words, d_filt, range_lo, range_hi});
    // MergeSortLimit = 0: block sort up to 1024 items, Onesweep above (never the merge path)
    using part_cfg = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                                rocprim::default_config, 0>;
    size_t bytes = 0;
    HIPCHK(rocprim::radix_sort_pairs<part_cfg>(nullptr, bytes, kin, c->pk_keys.as<u64>(), vin,
                                               c->pk_vals.as<u32>(), (size_t)N, 64 - pb, 64, st));
    ENSURE(c->tmp, bytes);
    HIPCHK(rocprim::radix_sort_pairs<part_cfg>(c->tmp.p, bytes, kin, c->pk_keys.as<u64>(), vin,
                                               c->pk_vals.as<u32>(), (size_t)N, 64 - pb, 64, st));
  }
  hipLaunchKernelGGL(k_part_bounds, dim3(blocks_for(n_parts + 1)), dim3(256), 0, st, c->pk_keys.as<u64>(), N,
                     pb, n_parts, c->pbeg.as<u32>(), c->ucount.as<u32>());
  HIPCHK(hipEventRecord(c->kev[0], st));
  hipLaunchKernelGGL(k_dedup_lds, dim3(n_parts), dim3(256), 0, st, c->pk_keys.as<u64>(), c->pk_vals.as<u32>(),
                     c->pbeg.as<u32>(), N, pb, c->pad_word.as<u64>(), c->pad_cf.as<uint2>(),
                     c->ucount.as<u32>(), c->pusable.as<u32>(), c->pslot.as<u32>(), c->d_ctr);
  HIPCHK(hipEventRecord(c->kev[1], st));
  hipLaunchKernelGGL(k_part_totals, dim3(1), dim3(256), 0, st, c->ucount.as<u32>(), c->pusable.as<u32>(),
                     n_parts, c->d_ctr);
  TRY(exscan_u32(c, c->ucount.as<u32>(), c->ubase.as<u32>(), (u64)n_parts + 1));
  HIPCHK(hipGetLastError());
  TRY(read_counters(c));
  if (c->h_ctr[CTR_OVERFULL]) { *overflowed = true; return HUMID_OK; }
  const u32 U = (u32)c->h_ctr[CTR_UNIQUE];
  s.usable = c->usable = c->h_ctr[CTR_USABLE];
  s.unique = c->U = U;
  if (U == 0) { HIPCHK(hipEventRecord(c->ev[1], st)); return HUMID_OK; }
  ENSURE(c->s_word, (size_t)U * 8);
  ENSURE(c->s_slot, (size_t)U * 4);
  ENSURE(c->s_cnt, (size_t)U * 4);
  ENSURE(c->s_first, (size_t)U * 4);
  hipLaunchKernelGGL(k_compact_padded, dim3(blocks_for((u64)n_parts * 64)), dim3(256), 0, st,
                     c->pad_word.as<u64>(), c->pbeg.as<u32>(), c->ucount.as<u32>(), c->ubase.as<u32>(), n_parts,
                     c->uniq_word.as<u64>(), c->uniq_slot.as<u32>());
  TRY(sort_pairs<u64, u32>(c, c->uniq_word.as<u64>(), c->s_word.as<u64>(), c->uniq_slot.as<u32>(),
                           c->s_slot.as<u32>(), U, 0, 2 * word_nt));
  hipLaunchKernelGGL(k_post_sort_padded, dim3(blocks_for(U)), dim3(256), 0, st, c->s_slot.as<u32>(),
                     c->pad_cf.as<uint2>(), U, c->s_cnt.as<u32>(), c->s_first.as<u32>());
  HIPCHK(hipEventRecord(c->ev[1], st));
  HIPCHK(hipGetLastError());
  return HUMID_OK;
}

// stage A dispatcher: partitioned LDS tables when the whole value range is counted here, the
// global table for a partial range (multi-GPU ranks) or after a bucket overflow.
static int stage_count(humid_ctx *c, const u64 *d_words, const u8 *d_filt, u32 N, u32 word_nt,
                       u64 range_lo, u64 range_hi, u64 expected_reads, humid_summary &s) {
  const bool full_range = (range_lo == 0 && range_hi == ~0ull);
  if (c->count_mode == 0 && full_range) {
    bool overflowed = false;
    TRY(stage_count_lds(c, d_words, d_filt, N, word_nt, range_lo, range_hi, s, &overflowed));
    if (!overflowed) return HUMID_OK;
  }
  return stage_count_global(c, d_words, d_filt, N, word_nt, range_lo, range_hi, expected_reads, s);
}

// ---- stage B: neighbours + clusters over a sorted unique array ---------------------------
// g_word[U] ascending, g_cnt[U] (device; the context's own arrays on one GPU, the gathered
// arrays of all ranks on several).  Leaves deg/nbr_off/nbr_idx/cl_of/maxleaf/cl_size/flag/
// pos/cid/ismax in the context.
// ext_edges != nullptr: the neighbour pairs are GIVEN (multi-GPU: every rank searched its share,
// humid_stage_pairs, and the shares were all-gathered); otherwise they are searched here.
static int stage_graph(humid_ctx *c, const u64 *g_word, const u32 *g_cnt, u32 U, u32 word_nt,
                       u32 distance, u32 method, humid_summary &s, u32 &n_pair_segs_out,
                       const u64 *ext_edges = nullptr, u64 n_ext_edges = 0) {
  hipStream_t st = c->stream;
  c->g_word = g_word;
  c->g_cnt = g_cnt;
  c->gU = U;
  // ---------------- 3. neighbours -----------------
  // deg has U+1 entries (last stays 0) so that one exclusive scan yields nbr_off[U] = 2E
  ENSURE(c->deg, (size_t)(U + 1) * 4);
  ENSURE(c->nbr_off, (size_t)(U + 1) * 4);
  ENSURE(c->parent, (size_t)U * 4);
  ENSURE(c->csize, (size_t)U * 4);
  HIPCHK(hipMemsetAsync(c->deg.p, 0, (size_t)(U + 1) * 4, st));
  HIPCHK(hipMemsetAsync(c->csize.p, 0, (size_t)U * 4, st));
  HIPCHK(hipMemsetAsync(&c->d_ctr[CTR_NONSINGLE], 0, 2 * sizeof(ull), st));   // NONSINGLE, MEMBERS
  HIPCHK(hipMemsetAsync(&c->d_ctr[CTR_OVERFULL], 0, sizeof(ull), st));
  hipLaunchKernelGGL(k_iota, dim3(blocks_for(U)), dim3(256), 0, st, c->parent.as<u32>(), U);
  u64 E = 0, M = 0, Mbig = 0;
  u32 n_pair_segs = 0;
  c->h_plan = make_plan(word_nt, distance, U, c->force_segments);
  const ComboPlan &plan = c->h_plan;
  EarlierMasks d_masks;                              // masks of all combos, for the first-combo rule
  for (u32 t = 0; t < MAX_COMBOS; t++) d_masks.m[t] = plan.mask[t];
  auto fields_of = [&](u32 cb) {
    ComboFields cf;
    cf.nf = plan.nfield[cb];
    for (u32 f = 0; f < MAX_FIELDS; f++) { cf.shift[f] = plan.shift[cb][f]; cf.width[f] = plan.width[cb][f]; }
    return cf;
  };
  const bool given = ext_edges != nullptr;
  const bool search = !given && distance > 0 && U > 1;
  if (given && n_ext_edges) {
    hipLaunchKernelGGL(k_edges_apply<false>, dim3(grid_stride_blocks(n_ext_edges)), dim3(256), 0, st, ext_edges,
                       n_ext_edges, U, c->deg.as<u32>(), c->parent.as<u32>(), (const u32 *)nullptr,
                       (u32 *)nullptr, (u32 *)nullptr, c->d_ctr);
    hipLaunchKernelGGL(k_comp_stats, dim3(blocks_for(U)), dim3(256), 0, st, c->deg.as<u32>(),
                       c->parent.as<u32>(), U, c->csize.as<u32>());
    hipLaunchKernelGGL(k_comp_count, dim3(512), dim3(256), 0, st, c->deg.as<u32>(), c->parent.as<u32>(),
                       c->csize.as<u32>(), U, c->d_ctr);
  }
  if (search) {
    const u32 nseg = plan.ncombo;
    n_pair_segs = nseg < 8 ? nseg : 8;
    if (nseg > 1) {
      ENSURE(c->seg_k0, (size_t)U * 8);
      ENSURE(c->seg_v0, (size_t)U * 4);
      ENSURE(c->seg_ks, (size_t)U * 8);                     // sorted keys: scratch, not kept
      ENSURE(c->seg_vs, (size_t)(nseg - 1) * U * 4);        // ranks in bucket order, per combo
    }
    // phase A: bucket order per combo; degrees and component forest
    for (u32 seg = 0; seg < nseg; seg++) {
      if (seg == 0) {
        HIPCHK(hipEventRecord(c->kev[20], st));
        hipLaunchKernelGGL((k_pairs<true, PM_COUNT>), dim3(blocks_for(U)), dim3(256), 0, st, g_word,
                           (const u32 *)nullptr, U, 0u, U, plan.mask[seg], d_masks, seg, distance, c->deg.as<u32>(),
                           c->parent.as<u32>(), (const u32 *)nullptr, (u32 *)nullptr, (u32 *)nullptr,
                           (u32 *)nullptr, (const u32 *)nullptr, (u64 *)nullptr);
      } else {
        u32 *vs = c->seg_vs.as<u32>() + (size_t)(seg - 1) * U;
        const u32 kb = plan.key_bits ? plan.key_bits : 1;
        if (kb <= 32) {
          hipLaunchKernelGGL(k_combo_keys<u32>, dim3(blocks_for(U)), dim3(256), 0, st, g_word, U, fields_of(seg),
                             c->seg_k0.as<u32>(), c->seg_v0.as<u32>());
          TRY(sort_pairs<u32, u32>(c, c->seg_k0.as<u32>(), c->seg_ks.as<u32>(), c->seg_v0.as<u32>(), vs, U, 0, kb));
        } else {
          hipLaunchKernelGGL(k_combo_keys<u64>, dim3(blocks_for(U)), dim3(256), 0, st, g_word, U, fields_of(seg),
                             c->seg_k0.as<u64>(), c->seg_v0.as<u32>());
          TRY(sort_pairs<u64, u32>(c, c->seg_k0.as<u64>(), c->seg_ks.as<u64>(), c->seg_v0.as<u32>(), vs, U, 0, kb));
        }
        if (seg < 8) HIPCHK(hipEventRecord(c->kev[20 + 2 * seg], st));
        hipLaunchKernelGGL((k_pairs<false, PM_COUNT>), dim3(blocks_for(U)), dim3(256), 0, st, g_word,
                           vs, U, 0u, U, plan.mask[seg], d_masks, seg, distance, c->deg.as<u32>(), c->parent.as<u32>(),
                           (const u32 *)nullptr, (u32 *)nullptr, (u32 *)nullptr, (u32 *)nullptr,
                           (const u32 *)nullptr, (u64 *)nullptr);
      }
      if (seg < 8) HIPCHK(hipEventRecord(c->kev[21 + 2 * seg], st));
    }
    hipLaunchKernelGGL(k_comp_stats, dim3(blocks_for(U)), dim3(256), 0, st, c->deg.as<u32>(),
                       c->parent.as<u32>(), U, c->csize.as<u32>());
    hipLaunchKernelGGL(k_comp_count, dim3(512), dim3(256), 0, st, c->deg.as<u32>(), c->parent.as<u32>(),
                       c->csize.as<u32>(), U, c->d_ctr);
  }
  TRY(exscan_u32(c, c->deg.as<u32>(), c->nbr_off.as<u32>(), (u64)U + 1));
  if (search || (given && n_ext_edges)) {
    HIPCHK(hipGetLastError());
    TRY(read_counters(c, c->nbr_off.as<u32>() + U));   // h_ctr[CTR_N-1] = 2E
    if (c->h_ctr[CTR_OVERFULL]) return fail(c, HUMID_E_INVALID, "malformed edge list (node index out of range)");
    const u64 twoE = c->h_ctr[CTR_N - 1] & 0xffffffffull;
    E = twoE / 2;
    M = c->h_ctr[CTR_NONSINGLE];
    Mbig = c->h_ctr[CTR_MEMBERS];
  }
  s.edges = c->E = E;
  s.nonsingle = c->M = M;
  ENSURE(c->nbr_idx, (size_t)(2 * E + 1) * 4);
  if (E > 0) {
    ENSURE(c->cur, (size_t)U * 4);
    HIPCHK(hipMemsetAsync(c->cur.p, 0, (size_t)U * 4, st));
    if (given)
      hipLaunchKernelGGL(k_edges_apply<true>, dim3(grid_stride_blocks(n_ext_edges)), dim3(256), 0, st, ext_edges,
                         n_ext_edges, U, (u32 *)nullptr, (u32 *)nullptr, c->nbr_off.as<u32>(),
                         c->cur.as<u32>(), c->nbr_idx.as<u32>(), c->d_ctr);
    // phase B: same loops, now writing the CSR rows
    for (u32 seg = 0; !given && seg < plan.ncombo; seg++) {
      if (seg < 8) HIPCHK(hipEventRecord(c->kev[4 + 2 * seg], st));
      if (seg == 0) {
        hipLaunchKernelGGL((k_pairs<true, PM_FILL>), dim3(blocks_for(U)), dim3(256), 0, st, g_word,
                           (const u32 *)nullptr, U, 0u, U, plan.mask[seg], d_masks, seg, distance, (u32 *)nullptr,
                           (u32 *)nullptr, c->nbr_off.as<u32>(), c->cur.as<u32>(), c->nbr_idx.as<u32>(),
                           (u32 *)nullptr, (const u32 *)nullptr, (u64 *)nullptr);
      } else {
        const u32 *vs = c->seg_vs.as<u32>() + (size_t)(seg - 1) * U;
        hipLaunchKernelGGL((k_pairs<false, PM_FILL>), dim3(blocks_for(U)), dim3(256), 0, st, g_word,
                           vs, U, 0u, U, plan.mask[seg], d_masks, seg, distance, (u32 *)nullptr, (u32 *)nullptr,
                           c->nbr_off.as<u32>(), c->cur.as<u32>(), c->nbr_idx.as<u32>(),
                           (u32 *)nullptr, (const u32 *)nullptr, (u64 *)nullptr);
      }
      if (seg < 8) HIPCHK(hipEventRecord(c->kev[5 + 2 * seg], st));
    }
    hipLaunchKernelGGL(k_sort_lists, dim3(blocks_for(U)), dim3(256), 0, st, c->nbr_off.as<u32>(), U,
                       c->nbr_idx.as<u32>());
  }
  HIPCHK(hipEventRecord(c->ev[2], st));

  // clusters
  TRY(cluster_stage(c, g_cnt, U, M, Mbig, method));
  hipLaunchKernelGGL(k_finalize_nodes, dim3(blocks_for(U)), dim3(256), 0, st, c->cl_of.as<u32>(),
                     c->pos.as<u32>(), c->maxleaf.as<u32>(), U, c->cid.as<u32>(), c->ismax.as<u8>());
  HIPCHK(hipGetLastError());
  n_pair_segs_out = n_pair_segs;
  return HUMID_OK;
}

// ---- multi-GPU: this rank's share of the neighbour search ----------------------------------
// Every rank holds the whole ascending unique array.  Rank r of P looks for the pairs whose
// first element lies in its slice: for the prefix combo the r-th P-th of the positions, for a
// sorted combo the words whose combo key falls into the r-th P-th of the key space (a bucket is
// never split).  The union over ranks is every pair exactly once; pairs come out as
// (smaller rank << 32 | larger rank), unordered.
static int stage_pairs_share(humid_ctx *c, const u64 *g_word, u32 U, u32 word_nt, u32 distance,
                             u32 part_rank, u32 part_world, u64 *n_edges_out) {
  hipStream_t st = c->stream;
  *n_edges_out = 0;
  if (distance == 0 || U < 2) return HUMID_OK;
  c->h_plan = make_plan(word_nt, distance, U, c->force_segments);
  const ComboPlan &plan = c->h_plan;
  EarlierMasks d_masks;
  for (u32 t = 0; t < MAX_COMBOS; t++) d_masks.m[t] = plan.mask[t];
  auto fields_of = [&](u32 cb) {
    ComboFields cf;
    cf.nf = plan.nfield[cb];
    for (u32 f = 0; f < MAX_FIELDS; f++) { cf.shift[f] = plan.shift[cb][f]; cf.width[f] = plan.width[cb][f]; }
    return cf;
  };
  const u32 nseg = plan.ncombo;
  const u32 kb = plan.key_bits ? plan.key_bits : 1;
  // share of the prefix combo: an equal slice of the positions
  const u32 p_lo = (u32)((u64)U * part_rank / part_world), p_hi = (u32)((u64)U * (part_rank + 1) / part_world);
  std::vector<u32> n_sel(nseg, 0);
  n_sel[0] = p_hi - p_lo;
  if (nseg > 1) {
    ENSURE(c->seg_k0, (size_t)U * 8);
    ENSURE(c->seg_v0, (size_t)U * 4);
    ENSURE(c->seg_ks, (size_t)U * 8);
    ENSURE(c->seg_vs, (size_t)(nseg - 1) * U * 4);
  }
  // key range of this rank: [floor(r 2^kb / P), floor((r+1) 2^kb / P) - 1]
  const unsigned __int128 span = (unsigned __int128)1 << kb;
  const u64 klo = (u64)(span * part_rank / part_world);
  const u64 khi = (u64)(span * (part_rank + 1) / part_world - 1);
  for (u32 seg = 1; seg < nseg; seg++) {
    u32 *vs = c->seg_vs.as<u32>() + (size_t)(seg - 1) * U;
    HIPCHK(hipMemsetAsync(&c->d_ctr[CTR_SPECIAL], 0, sizeof(ull), st));
    if (kb <= 32)
      hipLaunchKernelGGL(k_select_keyrange<u32>, dim3(COMPACT_BLOCKS), dim3(256), 0, st, g_word, U, fields_of(seg),
                         klo, khi, c->seg_k0.as<u32>(), c->seg_v0.as<u32>(), c->d_ctr);
    else
      hipLaunchKernelGGL(k_select_keyrange<u64>, dim3(COMPACT_BLOCKS), dim3(256), 0, st, g_word, U, fields_of(seg),
                         klo, khi, c->seg_k0.as<u64>(), c->seg_v0.as<u32>(), c->d_ctr);
    HIPCHK(hipGetLastError());
    TRY(read_counters(c));
    n_sel[seg] = (u32)c->h_ctr[CTR_SPECIAL];
    if (n_sel[seg] > 1) {
      if (kb <= 32) TRY(sort_pairs<u32, u32>(c, c->seg_k0.as<u32>(), c->seg_ks.as<u32>(), c->seg_v0.as<u32>(), vs, n_sel[seg], 0, kb));
      else TRY(sort_pairs<u64, u32>(c, c->seg_k0.as<u64>(), c->seg_ks.as<u64>(), c->seg_v0.as<u32>(), vs, n_sel[seg], 0, kb));
    }
  }
  u64 T = 0;
  std::vector<u64> base(nseg, 0);
  for (u32 seg = 0; seg < nseg; seg++) { base[seg] = T; T += n_sel[seg]; }
  if (T == 0) return HUMID_OK;
  if (T + 1 >= 0xffffffffull) return fail(c, HUMID_E_OVERFLOW, "too many positions in one share");
  ENSURE(c->pc, (size_t)(T + 1) * 4);
  ENSURE(c->poff, (size_t)(T + 1) * 4);
  HIPCHK(hipMemsetAsync(c->pc.as<u32>() + T, 0, 4, st));
  for (int phase = 0; phase < 2; phase++) {
    for (u32 seg = 0; seg < nseg; seg++) {
      if (n_sel[seg] == 0) continue;
      u32 *pcs = c->pc.as<u32>() + base[seg];
      const u32 *pos = c->poff.as<u32>() + base[seg];
      const u32 *vs = seg ? c->seg_vs.as<u32>() + (size_t)(seg - 1) * U : nullptr;
      u64 *ed = c->share_edges.as<u64>();
      const dim3 grid(blocks_for(n_sel[seg])), blk(256);
      if (seg == 0 && phase == 0)
        hipLaunchKernelGGL((k_pairs<true, PM_EMIT_COUNT>), grid, blk, 0, st, g_word, vs, U, p_lo, n_sel[0], plan.mask[0],
                           d_masks, 0u, distance, (u32 *)nullptr, (u32 *)nullptr, (const u32 *)nullptr,
                           (u32 *)nullptr, (u32 *)nullptr, pcs, pos, ed);
      else if (seg == 0)
        hipLaunchKernelGGL((k_pairs<true, PM_EMIT_FILL>), grid, blk, 0, st, g_word, vs, U, p_lo, n_sel[0], plan.mask[0],
                           d_masks, 0u, distance, (u32 *)nullptr, (u32 *)nullptr, (const u32 *)nullptr,
                           (u32 *)nullptr, (u32 *)nullptr, pcs, pos, ed);
      else if (phase == 0)
        hipLaunchKernelGGL((k_pairs<false, PM_EMIT_COUNT>), grid, blk, 0, st, g_word, vs, n_sel[seg], 0u, n_sel[seg],
                           plan.mask[seg], d_masks, seg, distance, (u32 *)nullptr, (u32 *)nullptr,
                           (const u32 *)nullptr, (u32 *)nullptr, (u32 *)nullptr, pcs, pos, ed);
      else
        hipLaunchKernelGGL((k_pairs<false, PM_EMIT_FILL>), grid, blk, 0, st, g_word, vs, n_sel[seg], 0u, n_sel[seg],
                           plan.mask[seg], d_masks, seg, distance, (u32 *)nullptr, (u32 *)nullptr,
                           (const u32 *)nullptr, (u32 *)nullptr, (u32 *)nullptr, pcs, pos, ed);
    }
    if (phase == 0) {
      TRY(exscan_u32(c, c->pc.as<u32>(), c->poff.as<u32>(), T + 1));
      HIPCHK(hipGetLastError());
      TRY(read_counters(c, c->poff.as<u32>() + T));
      const u64 E = c->h_ctr[CTR_N - 1] & 0xffffffffull;
      *n_edges_out = E;
      if (E == 0) return HUMID_OK;
      ENSURE(c->share_edges, (size_t)E * 8);
    }
  }
  HIPCHK(hipGetLastError());
  return HUMID_OK;
}

// ---- stage C: per-read outputs -------------------------------------------------------------
// l_cid/l_ismax: cluster id and maxLeaf flag of THIS context's unique words in local walk order
// (on one GPU the arrays stage B left behind; on several, this rank's slice of them).
static int stage_map(humid_ctx *c, const u32 *l_cid, const u8 *l_ismax, u32 N, u32 *d_cid, u8 *d_keep) {
  hipStream_t st = c->stream;
  const u32 U = (u32)c->U;
  if (U > 0)
    hipLaunchKernelGGL(k_slot_results, dim3(blocks_for(U)), dim3(256), 0, st, l_cid, l_ismax,
                       c->s_first.as<u32>(), c->s_slot.as<u32>(), U, c->slot_out.as<u64>());
  HIPCHK(hipEventRecord(c->ev[3], st));
  if (c->last_count_lds) {
    // pk_keys (the partitioned keys) is dead by now: reuse it for the packed per-read results
    u32 *packed = c->pk_keys.as<u32>();
    hipLaunchKernelGGL(k_read_map_part, dim3(grid_stride_blocks(N)), dim3(256), 0, st, c->pk_vals.as<u32>(),
                       c->pslot.as<u32>(), c->slot_out.as<u64>(), N, packed);
    HIPCHK(hipEventRecord(c->kev[36], st));
    hipLaunchKernelGGL(k_split_out, dim3(grid_stride_blocks(N)), dim3(256), 0, st, packed, N, d_cid, d_keep);
  } else
    hipLaunchKernelGGL(k_read_map, dim3(grid_stride_blocks(N)), dim3(256), 0, st, c->slot_of_read.as<u32>(),
                       c->slot_out.as<u64>(), N, d_cid, d_keep);
  HIPCHK(hipEventRecord(c->ev[4], st));
  HIPCHK(hipGetLastError());
  return HUMID_OK;
}

static int check_run_args(humid_ctx *c, u64 n_reads, u32 word_nt, u32 method) {
  if (word_nt == 0) return fail(c, HUMID_E_INVALID, "word_nt must be >= 1");
  if (word_nt > 32) return fail(c, HUMID_E_UNSUPPORTED, "word_nt %u > 32 is not supported by the HIP path", word_nt);
  if (method > 1) return fail(c, HUMID_E_INVALID, "method must be 0 (directional) or 1 (maximum)");
  if (n_reads > 0x7fffffffull) return fail(c, HUMID_E_OVERFLOW, "n_reads %llu exceeds 2^31-1", (ull)n_reads);
  return HUMID_OK;
}

// ---- the full pipeline on device buffers (one GPU) -------------------------------------------
static int run_device(humid_ctx *c, const u64 *d_words, const u8 *d_filt, u64 n_reads, u32 word_nt,
                      u32 distance, u32 method, u32 *d_cid, u8 *d_keep, humid_summary *sum) {
  if (!c) return HUMID_E_INVALID;
  c->have_run = false;
  c->graph_mode = false;
  c->have_graph = false;
  TRY(check_run_args(c, n_reads, word_nt, method));
  if (n_reads && (!d_words || !d_filt || !d_cid || !d_keep)) return fail(c, HUMID_E_INVALID, "null buffer");
  HIPCHK(hipSetDevice(c->device));
  hipStream_t st = c->stream;
  const u32 N = (u32)n_reads;
  humid_summary s;
  memset(&s, 0, sizeof s);
  s.total = n_reads;
  c->N = n_reads; c->U = c->E = c->M = c->C = c->usable = 0;
  c->word_nt = word_nt; c->distance = distance; c->method = method;
  c->gU = 0;
  if (N == 0) { if (sum) *sum = s; c->have_run = c->have_graph = true; return HUMID_OK; }
  TRY(stage_count(c, d_words, d_filt, N, word_nt, 0ull, ~0ull, 0, s));
  const u32 U = (u32)c->U;
  if (U == 0) {   // everything filtered
    HIPCHK(hipMemsetAsync(d_cid, 0, (size_t)N * 4, st));
    HIPCHK(hipMemsetAsync(d_keep, 0, (size_t)N, st));
    HIPCHK(hipStreamSynchronize(st));
    if (sum) *sum = s;
    c->have_run = c->have_graph = true;
    return HUMID_OK;
  }
  u32 n_pair_segs = 0;
  TRY(stage_graph(c, c->s_word.as<u64>(), c->s_cnt.as<u32>(), U, word_nt, distance, method, s, n_pair_segs));
  TRY(stage_map(c, c->cid.as<u32>(), c->ismax.as<u8>(), N, d_cid, d_keep));
  TRY(n_clusters_from_scan(c, U, &c->C));
  s.clusters = c->C;
  const u64 E = c->E, M = c->M;
  HIPCHK(hipEventElapsedTime(&s.ms_count, c->ev[0], c->ev[1]));
  HIPCHK(hipEventElapsedTime(&s.ms_neighbours, c->ev[1], c->ev[2]));
  HIPCHK(hipEventElapsedTime(&s.ms_cluster, c->ev[2], c->ev[3]));
  HIPCHK(hipEventElapsedTime(&s.ms_map, c->ev[3], c->ev[4]));
  HIPCHK(hipEventElapsedTime(&s.ms_total, c->ev[0], c->ev[4]));
  HIPCHK(hipEventElapsedTime(&s.ms_k_insert, c->kev[0], c->kev[1]));
  if (c->last_count_lds) HIPCHK(hipEventElapsedTime(&s.ms_k_map, c->ev[3], c->kev[36]));   // k_read_map_part alone
  else s.ms_k_map = s.ms_map;   // ev[3]..ev[4] bracket exactly the k_read_map launch
  s.count_mode_used = c->last_count_lds ? 0u : 1u;
  if (M > 0) HIPCHK(hipEventElapsedTime(&s.ms_k_cluster, c->kev[2], c->kev[3]));
  for (u32 g = 0; g < n_pair_segs; g++) {
    float t = 0;
    HIPCHK(hipEventElapsedTime(&t, c->kev[20 + 2 * g], c->kev[21 + 2 * g]));   // count phase
    s.ms_k_pairs += t;
    if (E > 0) {
      HIPCHK(hipEventElapsedTime(&t, c->kev[4 + 2 * g], c->kev[5 + 2 * g]));   // fill phase
      s.ms_k_pairs += t;
    }
  }
  if (sum) *sum = s;
  c->have_run = true;
  c->have_graph = true;
  return HUMID_OK;
}

// --------------------------------------------------------------------------------
// C ABI
// --------------------------------------------------------------------------------
extern "C" {

uint32_t humid_abi_version(void) { return HUMID_ABI_VERSION; }

int humid_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

const char *humid_last_error(const humid_ctx *ctx) { return ctx ? ctx->err.c_str() : g_err.c_str(); }

int humid_ctx_create(humid_ctx **out, int device, void *stream) {
  humid_ctx *c = nullptr;
  if (!out) return fail(nullptr, HUMID_E_INVALID, "out is null");
  *out = nullptr;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0)
    return fail(nullptr, HUMID_E_HIP, "no HIP device available (%s); this library has no CPU fallback",
                hipGetErrorString(e));
  if (device < 0) { if (hipGetDevice(&device) != hipSuccess) device = 0; }
  if (device >= ndev) return fail(nullptr, HUMID_E_INVALID, "device %d out of range (%d devices)", device, ndev);
  c = new (std::nothrow) humid_ctx();
  if (!c) return fail(nullptr, HUMID_E_NOMEM, "host allocation failed");
  c->device = device;
  auto bail = [&](hipError_t err, const char *what) {
    int rc = fail(nullptr, HUMID_E_HIP, "%s: %s", what, hipGetErrorString(err));
    humid_ctx_destroy(c);
    return rc;
  };
  if ((e = hipSetDevice(device)) != hipSuccess) return bail(e, "hipSetDevice");
  if (stream) { c->stream = (hipStream_t)stream; c->own_stream = false; }
  else {
    if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) return bail(e, "hipStreamCreate");
    c->own_stream = true;
  }
  if ((e = hipMalloc((void **)&c->d_ctr, CTR_N * sizeof(ull))) != hipSuccess) return bail(e, "hipMalloc");
  if ((e = hipHostMalloc((void **)&c->h_ctr, CTR_N * sizeof(ull), hipHostMallocDefault)) != hipSuccess) return bail(e, "hipHostMalloc");
  for (auto &ev : c->ev)
    if ((e = hipEventCreate(&ev)) != hipSuccess) return bail(e, "hipEventCreate");
  for (auto &ev : c->kev)
    if ((e = hipEventCreate(&ev)) != hipSuccess) return bail(e, "hipEventCreate");
  if (const char *m = getenv("HUMID_COUNT_MODE")) c->count_mode = (atoi(m) == 1) ? 1 : 0;
  *out = c;
  return HUMID_OK;
}

void humid_ctx_destroy(humid_ctx *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  DBuf *bufs[] = {&c->in_words, &c->in_filt, &c->out_cid, &c->out_keep, &c->table, &c->pk_keys, &c->pk_vals,
                  &c->pbeg, &c->ucount, &c->pusable, &c->ubase, &c->pad_word, &c->pad_cf, &c->pslot,
                  &c->opos, &c->own_packed, &c->owner, &c->owner_sorted, &c->perm, &c->small, &c->pc, &c->poff, &c->share_edges,
                  &c->slot_out, &c->slot_of_read, &c->uniq_slot, &c->uniq_word, &c->s_word, &c->s_slot,
                  &c->s_cnt, &c->s_first, &c->deg, &c->nbr_off, &c->nbr_idx, &c->seg_k0, &c->seg_ks,
                  &c->seg_v0, &c->seg_vs, &c->csize, &c->cur, &c->plan_dev, &c->parent, &c->mk0, &c->mk1, &c->cl_of,
                  &c->maxleaf, &c->cl_size, &c->flag, &c->pos, &c->cid, &c->ismax, &c->stk, &c->tmp,
                  &c->scratch};
  for (DBuf *b : bufs) b->release();
  if (c->d_ctr) (void)hipFree(c->d_ctr);
  if (c->h_ctr) (void)hipHostFree(c->h_ctr);
  for (auto &ev : c->ev) if (ev) (void)hipEventDestroy(ev);
  for (auto &ev : c->kev) if (ev) (void)hipEventDestroy(ev);
  if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

int humid_ctx_set_option(humid_ctx *c, const char *key, int64_t value) {
  if (!c || !key) return fail(c, HUMID_E_INVALID, "null argument");
  if (strcmp(key, "count_mode") == 0) {
    if (value != 0 && value != 1) return fail(c, HUMID_E_INVALID, "count_mode must be 0 (LDS-partitioned) or 1 (global table)");
    c->count_mode = (int)value;
    return HUMID_OK;
  }
  if (strcmp(key, "plan_segments") == 0) {
    if (value < 0 || value > 32) return fail(c, HUMID_E_INVALID, "plan_segments must be 0 (auto) .. 32");
    c->force_segments = (u32)value;
    return HUMID_OK;
  }
  return fail(c, HUMID_E_INVALID, "unknown option '%s'", key);
}

int humid_dedup_run_device(humid_ctx *c, const uint64_t *d_words, const uint8_t *d_filtered,
                           uint64_t n_reads, uint32_t word_nt, uint32_t distance, uint32_t method,
                           uint32_t *d_cluster_id, uint8_t *d_keep, humid_summary *summary) {
  if (!c) return fail(nullptr, HUMID_E_INVALID, "ctx is null");
  return run_device(c, d_words, d_filtered, n_reads, word_nt, distance, method, d_cluster_id, d_keep, summary);
}

int humid_dedup_run(humid_ctx *c, const uint64_t *words, const uint8_t *filtered, uint64_t n_reads,
                    uint32_t word_nt, uint32_t distance, uint32_t method, uint32_t *cluster_id,
                    uint8_t *keep, humid_summary *summary) {
  if (!c) return fail(nullptr, HUMID_E_INVALID, "ctx is null");
  if (n_reads && (!words || !filtered || !cluster_id || !keep)) return fail(c, HUMID_E_INVALID, "null buffer");
  if (n_reads > 0x7fffffffull) return fail(c, HUMID_E_OVERFLOW, "n_reads %llu exceeds 2^31-1", (ull)n_reads);
  HIPCHK(hipSetDevice(c->device));
  hipStream_t st = c->stream;
  humid_summary s;
  memset(&s, 0, sizeof s);
  hipEvent_t e0 = c->ev[5];
  hipEvent_t e1, e2, e3;
  HIPCHK(hipEventCreate(&e1)); HIPCHK(hipEventCreate(&e2)); HIPCHK(hipEventCreate(&e3));
  size_t n = (size_t)n_reads;
  ENSURE(c->in_words, n * 8 + 8);
  ENSURE(c->in_filt, n + 8);
  ENSURE(c->out_cid, n * 4 + 8);
  ENSURE(c->out_keep, n + 8);
  HIPCHK(hipEventRecord(e0, st));
  if (n) {
    HIPCHK(hipMemcpyAsync(c->in_words.p, words, n * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(c->in_filt.p, filtered, n, hipMemcpyHostToDevice, st));
  }
  HIPCHK(hipEventRecord(e1, st));
  int rc = run_device(c, c->in_words.as<u64>(), c->in_filt.as<u8>(), n_reads, word_nt, distance, method,
                      c->out_cid.as<u32>(), c->out_keep.as<u8>(), &s);
  if (rc == HUMID_OK) {
    hipError_t he = hipEventRecord(e2, st);
    if (he == hipSuccess && n) he = hipMemcpyAsync(cluster_id, c->out_cid.p, n * 4, hipMemcpyDeviceToHost, st);
    if (he == hipSuccess && n) he = hipMemcpyAsync(keep, c->out_keep.p, n, hipMemcpyDeviceToHost, st);
    if (he == hipSuccess) he = hipEventRecord(e3, st);
    if (he == hipSuccess) he = hipStreamSynchronize(st);
    if (he == hipSuccess) he = hipEventElapsedTime(&s.ms_h2d, e0, e1);
    if (he == hipSuccess) he = hipEventElapsedTime(&s.ms_d2h, e2, e3);
    if (he != hipSuccess) rc = fail(c, HUMID_E_HIP, "copy back: %s", hipGetErrorString(he));
  }
  (void)hipEventDestroy(e1); (void)hipEventDestroy(e2); (void)hipEventDestroy(e3);
  if (rc == HUMID_OK && summary) *summary = s;
  return rc;
}

#define NEED_RUN()                                                                            \
  do {                                                                                        \
    if (!c) return fail(nullptr, HUMID_E_INVALID, "ctx is null");                             \
    if (!c->have_graph || c->graph_mode) return fail(c, HUMID_E_STATE, "no completed dedup run / graph stage in this context"); \
    HIPCHK(hipSetDevice(c->device));                                                          \
  } while (0)

#define D2H(dst, src, bytes)                                                                  \
  do { if ((dst) && (bytes)) HIPCHK(hipMemcpyAsync((dst), (src), (bytes), hipMemcpyDeviceToHost, c->stream)); } while (0)

int humid_get_leaves(humid_ctx *c, uint64_t *word, uint32_t *count, uint32_t *first_read,
                     uint32_t *degree, uint32_t *cluster_id, uint8_t *is_max_leaf) {
  NEED_RUN();
  size_t U = (size_t)c->gU;
  if (U == 0) return HUMID_OK;
  if (first_read && !c->have_run)
    return fail(c, HUMID_E_STATE, "first_read is only available after a single-GPU humid_dedup_run*");
  D2H(word, c->g_word, U * 8);
  D2H(count, c->g_cnt, U * 4);
  D2H(first_read, c->s_first.p, U * 4);
  D2H(degree, c->deg.p, U * 4);
  D2H(cluster_id, c->cid.p, U * 4);
  D2H(is_max_leaf, c->ismax.p, U);
  HIPCHK(hipStreamSynchronize(c->stream));
  return HUMID_OK;
}

int humid_get_adjacency(humid_ctx *c, uint32_t *nbr_off, uint32_t *nbr_idx) {
  NEED_RUN();
  size_t U = (size_t)c->gU;
  if (U == 0) { if (nbr_off) nbr_off[0] = 0; return HUMID_OK; }
  D2H(nbr_off, c->nbr_off.p, (U + 1) * 4);
  D2H(nbr_idx, c->nbr_idx.p, (size_t)(2 * c->E) * 4);
  HIPCHK(hipStreamSynchronize(c->stream));
  return HUMID_OK;
}

static int export_clusters(humid_ctx *c, u32 U, u64 C, uint64_t *size, uint32_t *max_count, uint32_t *max_leaf) {
  if (C == 0) return HUMID_OK;
  ENSURE(c->scratch, (size_t)C * 16);
  u64 *d_size = c->scratch.as<u64>();
  u32 *d_mc = (u32 *)(d_size + C);
  u32 *d_ml = d_mc + C;
  hipLaunchKernelGGL(k_export_clusters, dim3(blocks_for(U)), dim3(256), 0, c->stream, c->flag.as<u32>(),
                     c->pos.as<u32>(), c->maxleaf.as<u32>(), c->cl_size.as<u64>(), c->g_cnt, U,
                     d_size, d_mc, d_ml);
  HIPCHK(hipGetLastError());
  D2H(size, d_size, (size_t)C * 8);
  D2H(max_count, d_mc, (size_t)C * 4);
  D2H(max_leaf, d_ml, (size_t)C * 4);
  HIPCHK(hipStreamSynchronize(c->stream));
  return HUMID_OK;
}

int humid_get_clusters(humid_ctx *c, uint64_t *size, uint32_t *max_count, uint32_t *max_leaf) {
  NEED_RUN();
  return export_clusters(c, c->gU, c->C, size, max_count, max_leaf);
}

int humid_get_histogram(humid_ctx *c, uint32_t which, uint64_t *keys, uint64_t *values, uint64_t cap,
                        uint64_t *n_out) {
  NEED_RUN();
  if (which > 2 || !n_out) return fail(c, HUMID_E_INVALID, "bad histogram selector");
  *n_out = 0;
  const u32 U = c->gU;
  u64 n = (which == 2) ? c->C : U;
  if (n == 0) return HUMID_OK;
  hipStream_t st = c->stream;
  // layout of scratch: vals[n] | sorted[n] | uniq[n] | counts u32[n] | runs u32
  ENSURE(c->scratch, (size_t)n * 28 + 64);
  u64 *vals = c->scratch.as<u64>();
  u64 *sorted = vals + n;
  u64 *uniq = sorted + n;
  u32 *counts = (u32 *)(uniq + n);
  u32 *runs = counts + n;
  if (which == 0) hipLaunchKernelGGL(k_widen32, dim3(blocks_for(U)), dim3(256), 0, st, c->g_cnt, U, vals);
  else if (which == 1) hipLaunchKernelGGL(k_widen32, dim3(blocks_for(U)), dim3(256), 0, st, c->deg.as<u32>(), U, vals);
  else hipLaunchKernelGGL(k_creator_sizes, dim3(blocks_for(U)), dim3(256), 0, st, c->flag.as<u32>(),
                          c->pos.as<u32>(), c->cl_size.as<u64>(), U, vals);
  TRY(sort_keys<u64>(c, vals, sorted, n, 0, 64));
  {
    size_t bytes = 0;
    HIPCHK(rocprim::run_length_encode(nullptr, bytes, sorted, (unsigned int)n, uniq, counts, runs, st));
    ENSURE(c->tmp, bytes);
    HIPCHK(rocprim::run_length_encode(c->tmp.p, bytes, sorted, (unsigned int)n, uniq, counts, runs, st));
  }
  u32 h_runs = 0;
  HIPCHK(hipMemcpyAsync(&h_runs, runs, 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  *n_out = h_runs;
  u64 take = h_runs < cap ? h_runs : cap;
  if (take && keys && values) {
    std::vector<u32> hc(take);
    HIPCHK(hipMemcpyAsync(keys, uniq, (size_t)take * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(hc.data(), counts, (size_t)take * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    for (u64 i = 0; i < take; i++) values[i] = hc[i];
  }
  return HUMID_OK;
}

int humid_cluster_graph(humid_ctx *c, const uint32_t *count, const uint32_t *nbr_off,
                        const uint32_t *nbr_idx, uint32_t n_leaves, uint32_t method,
                        uint32_t *leaf_cluster, uint64_t *cl_size, uint32_t *cl_max_count,
                        uint32_t *cl_max_leaf, uint32_t *n_clusters) {
  if (!c) return fail(nullptr, HUMID_E_INVALID, "ctx is null");
  if (method > 1) return fail(c, HUMID_E_INVALID, "method must be 0 or 1");
  if (n_clusters) *n_clusters = 0;
  const u32 U = n_leaves;
  if (U == 0) return HUMID_OK;
  if (!count || !nbr_off) return fail(c, HUMID_E_INVALID, "null buffer");
  const u32 twoE = nbr_off[U];
  if (twoE && !nbr_idx) return fail(c, HUMID_E_INVALID, "null nbr_idx");
  for (u32 u = 0; u < U; u++)
    if (nbr_off[u] > nbr_off[u + 1]) return fail(c, HUMID_E_INVALID, "nbr_off not monotone at %u", u);
  for (u32 k = 0; k < twoE; k++)
    if (nbr_idx[k] >= U) return fail(c, HUMID_E_INVALID, "nbr_idx[%u] = %u out of range", k, nbr_idx[k]);
  HIPCHK(hipSetDevice(c->device));
  hipStream_t st = c->stream;
  c->have_run = false;
  c->have_graph = false;
  c->graph_mode = true;
  ENSURE(c->s_cnt, (size_t)U * 4);
  c->g_cnt = c->s_cnt.as<u32>();
  c->gU = U;
  ENSURE(c->deg, (size_t)U * 4);
  ENSURE(c->nbr_off, (size_t)(U + 1) * 4);
  ENSURE(c->nbr_idx, (size_t)(twoE + 1) * 4);
  ENSURE(c->parent, (size_t)U * 4);
  std::vector<u32> hdeg(U);
  for (u32 u = 0; u < U; u++) hdeg[u] = nbr_off[u + 1] - nbr_off[u];
  {
    // NLeaf::neighbours is always symmetric (src/humid.cc:121-122, tests' link()); the component
    // walk relies on it.  Two linked leaves of count 0 make maxNeighbour_ (cluster.cc:39-51)
    // ping-pong forever in the reference: refuse instead of hanging the GPU.
    std::vector<u64> fwd, rev;
    fwd.reserve(twoE); rev.reserve(twoE);
    for (u32 u = 0; u < U; u++)
      for (u32 k = nbr_off[u]; k < nbr_off[u + 1]; k++) {
        const u32 v = nbr_idx[k];
        if (count[u] == 0 && count[v] == 0)
          return fail(c, HUMID_E_INVALID, "leaves %u and %u are linked and both have count 0", u, v);
        fwd.push_back(((u64)u << 32) | v);
        rev.push_back(((u64)v << 32) | u);
      }
    std::sort(fwd.begin(), fwd.end());
    std::sort(rev.begin(), rev.end());
    if (fwd != rev) return fail(c, HUMID_E_INVALID, "neighbour lists are not symmetric");
  }
  HIPCHK(hipMemcpyAsync(c->s_cnt.p, count, (size_t)U * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(c->deg.p, hdeg.data(), (size_t)U * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(c->nbr_off.p, nbr_off, (size_t)(U + 1) * 4, hipMemcpyHostToDevice, st));
  if (twoE) HIPCHK(hipMemcpyAsync(c->nbr_idx.p, nbr_idx, (size_t)twoE * 4, hipMemcpyHostToDevice, st));
  ENSURE(c->csize, (size_t)U * 4);
  HIPCHK(hipMemsetAsync(c->csize.p, 0, (size_t)U * 4, st));
  HIPCHK(hipMemsetAsync(&c->d_ctr[CTR_NONSINGLE], 0, 2 * sizeof(ull), st));
  hipLaunchKernelGGL(k_iota, dim3(blocks_for(U)), dim3(256), 0, st, c->parent.as<u32>(), U);
  hipLaunchKernelGGL(k_union_csr, dim3(blocks_for(U)), dim3(256), 0, st, c->nbr_off.as<u32>(),
                     c->nbr_idx.as<u32>(), U, c->parent.as<u32>());
  hipLaunchKernelGGL(k_comp_stats, dim3(blocks_for(U)), dim3(256), 0, st, c->deg.as<u32>(),
                     c->parent.as<u32>(), U, c->csize.as<u32>());
  hipLaunchKernelGGL(k_comp_count, dim3(512), dim3(256), 0, st, c->deg.as<u32>(), c->parent.as<u32>(),
                     c->csize.as<u32>(), U, c->d_ctr);
  HIPCHK(hipGetLastError());
  TRY(read_counters(c));   // also drains the stream: hdeg is a host temporary
  TRY(cluster_stage(c, c->s_cnt.as<u32>(), U, c->h_ctr[CTR_NONSINGLE], c->h_ctr[CTR_MEMBERS], method));
  hipLaunchKernelGGL(k_finalize_nodes, dim3(blocks_for(U)), dim3(256), 0, st, c->cl_of.as<u32>(),
                     c->pos.as<u32>(), c->maxleaf.as<u32>(), U, c->cid.as<u32>(), c->ismax.as<u8>());
  HIPCHK(hipGetLastError());
  u64 C = 0;
  TRY(n_clusters_from_scan(c, U, &C));
  D2H(leaf_cluster, c->cid.p, (size_t)U * 4);
  HIPCHK(hipStreamSynchronize(st));
  if (n_clusters) *n_clusters = (u32)C;
  return export_clusters(c, U, C, cl_size, cl_max_count, cl_max_leaf);
}

int humid_at_least_double(humid_ctx *c, uint64_t a, uint64_t b, int *result) {
  if (!c || !result) return fail(c, HUMID_E_INVALID, "null argument");
  HIPCHK(hipSetDevice(c->device));
  ENSURE(c->scratch, 64);
  hipLaunchKernelGGL(k_at_least_double, dim3(1), dim3(1), 0, c->stream, a, b, c->scratch.as<int>());
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(result, c->scratch.p, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return HUMID_OK;
}

// ---- multi-GPU stages (device pointers; see humid_amd/sharded.py) ----------------------------
int humid_stage_histogram(humid_ctx *c, const uint64_t *d_words, const uint8_t *d_filtered,
                          uint64_t n_reads, uint32_t word_nt, uint32_t bits, uint32_t *d_hist) {
  if (!c) return fail(nullptr, HUMID_E_INVALID, "ctx is null");
  TRY(check_run_args(c, n_reads, word_nt, 0));
  if (bits == 0 || bits > 12 || bits > 2 * word_nt || !d_hist) return fail(c, HUMID_E_INVALID, "bits must be 1..min(12, 2*word_nt)");
  HIPCHK(hipSetDevice(c->device));
  const u32 n_bins = 1u << bits;
  HIPCHK(hipMemsetAsync(d_hist, 0, n_bins * 4, c->stream));
  if (n_reads)
    hipLaunchKernelGGL(k_top_hist, dim3(512), dim3(256), n_bins * 4, c->stream, d_words, d_filtered,
                       (u32)n_reads, 2 * word_nt - bits, n_bins, d_hist);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(c->stream));
  return HUMID_OK;
}

int humid_stage_count(humid_ctx *c, const uint64_t *d_words, const uint8_t *d_filtered, uint64_t n_reads,
                      uint32_t word_nt, uint64_t range_lo, uint64_t range_hi, uint64_t expected_reads,
                      uint64_t *n_unique, uint64_t *n_usable) {
  if (!c) return fail(nullptr, HUMID_E_INVALID, "ctx is null");
  c->have_run = c->have_graph = false;
  c->graph_mode = false;
  TRY(check_run_args(c, n_reads, word_nt, 0));
  if (n_reads && (!d_words || !d_filtered)) return fail(c, HUMID_E_INVALID, "null buffer");
  HIPCHK(hipSetDevice(c->device));
  humid_summary s;
  memset(&s, 0, sizeof s);
  c->N = n_reads; c->U = c->E = c->M = c->C = c->usable = 0;
  c->word_nt = word_nt;
  if (n_reads) TRY(stage_count(c, d_words, d_filtered, (u32)n_reads, word_nt, range_lo, range_hi, expected_reads, s));
  HIPCHK(hipStreamSynchronize(c->stream));
  if (n_unique) *n_unique = c->U;
  if (n_usable) *n_usable = c->usable;
  return HUMID_OK;
}

int humid_stage_unique(humid_ctx *c, const uint64_t **d_word, const uint32_t **d_count,
                       const uint32_t **d_first) {
  if (!c) return fail(nullptr, HUMID_E_INVALID, "ctx is null");
  if (d_word) *d_word = c->U ? c->s_word.as<u64>() : nullptr;
  if (d_count) *d_count = c->U ? c->s_cnt.as<u32>() : nullptr;
  if (d_first) *d_first = c->U ? c->s_first.as<u32>() : nullptr;
  return HUMID_OK;
}

int humid_stage_graph(humid_ctx *c, const uint64_t *d_g_word, const uint32_t *d_g_count,
                      uint64_t n_unique, uint32_t word_nt, uint32_t distance, uint32_t method,
                      const uint32_t **d_cluster_id, const uint8_t **d_is_max, humid_summary *summary) {
  if (!c) return fail(nullptr, HUMID_E_INVALID, "ctx is null");
  c->have_graph = false;
  c->graph_mode = false;
  TRY(check_run_args(c, n_unique, word_nt, method));
  HIPCHK(hipSetDevice(c->device));
  humid_summary s;
  memset(&s, 0, sizeof s);
  s.unique = n_unique;
  c->distance = distance; c->method = method;
  c->gU = 0; c->E = c->M = c->C = 0;
  if (d_cluster_id) *d_cluster_id = nullptr;
  if (d_is_max) *d_is_max = nullptr;
  if (n_unique) {
    if (!d_g_word || !d_g_count) return fail(c, HUMID_E_INVALID, "null buffer");
    u32 nps = 0;
    TRY(stage_graph(c, d_g_word, d_g_count, (u32)n_unique, word_nt, distance, method, s, nps));
    TRY(n_clusters_from_scan(c, (u32)n_unique, &c->C));
    s.clusters = c->C;
    if (d_cluster_id) *d_cluster_id = c->cid.as<u32>();
    if (d_is_max) *d_is_max = c->ismax.as<u8>();
  }
  HIPCHK(hipStreamSynchronize(c->stream));
  if (summary) *summary = s;
  c->have_graph = true;
  return HUMID_OK;
}

int humid_stage_map(humid_ctx *c, const uint32_t *d_local_cluster_id, const uint8_t *d_local_is_max,
                    uint64_t n_reads, uint32_t *d_cluster_id, uint8_t *d_keep) {
  if (!c) return fail(nullptr, HUMID_E_INVALID, "ctx is null");
  if (n_reads != c->N) return fail(c, HUMID_E_STATE, "n_reads differs from the preceding humid_stage_count");
  if (n_reads && (!d_cluster_id || !d_keep)) return fail(c, HUMID_E_INVALID, "null buffer");
  if (c->U && (!d_local_cluster_id || !d_local_is_max)) return fail(c, HUMID_E_INVALID, "null buffer");
  HIPCHK(hipSetDevice(c->device));
  if (n_reads) TRY(stage_map(c, d_local_cluster_id, d_local_is_max, (u32)n_reads, d_cluster_id, d_keep));
  HIPCHK(hipStreamSynchronize(c->stream));
  return HUMID_OK;
}

int humid_stage_pairs(humid_ctx *c, const uint64_t *d_g_word, uint64_t n_unique, uint32_t word_nt,
                      uint32_t distance, uint32_t part_rank, uint32_t part_world, const uint64_t **d_edges,
                      uint64_t *n_edges) {
  if (!c) return fail(nullptr, HUMID_E_INVALID, "ctx is null");
  if (!d_edges || !n_edges || part_world == 0 || part_rank >= part_world) return fail(c, HUMID_E_INVALID, "bad argument");
  TRY(check_run_args(c, n_unique, word_nt, 0));
  HIPCHK(hipSetDevice(c->device));
  *d_edges = nullptr;
  *n_edges = 0;
  if (n_unique && !d_g_word) return fail(c, HUMID_E_INVALID, "null buffer");
  u64 E = 0;
  if (n_unique) TRY(stage_pairs_share(c, d_g_word, (u32)n_unique, word_nt, distance, part_rank, part_world, &E));
  HIPCHK(hipStreamSynchronize(c->stream));
  *n_edges = E;
  *d_edges = E ? c->share_edges.as<u64>() : nullptr;
  return HUMID_OK;
}

int humid_stage_graph_edges(humid_ctx *c, const uint64_t *d_g_word, const uint32_t *d_g_count, uint64_t n_unique,
                            const uint64_t *d_edges, uint64_t n_edges, uint32_t word_nt, uint32_t distance,
                            uint32_t method, const uint32_t **d_cluster_id, const uint8_t **d_is_max,
                            humid_summary *summary) {
  if (!c) return fail(nullptr, HUMID_E_INVALID, "ctx is null");
  c->have_graph = false;
  c->graph_mode = false;
  TRY(check_run_args(c, n_unique, word_nt, method));
  if (n_edges >= 0x7fffffffull) return fail(c, HUMID_E_OVERFLOW, "2*edges exceeds 32 bits");
  HIPCHK(hipSetDevice(c->device));
  humid_summary s;
  memset(&s, 0, sizeof s);
  s.unique = n_unique;
  c->distance = distance; c->method = method;
  c->gU = 0; c->E = c->M = c->C = 0;
  if (d_cluster_id) *d_cluster_id = nullptr;
  if (d_is_max) *d_is_max = nullptr;
  if (n_unique) {
    if (!d_g_word || !d_g_count || (n_edges && !d_edges)) return fail(c, HUMID_E_INVALID, "null buffer");
    u32 nps = 0;
    static const u64 no_edges = 0;
    TRY(stage_graph(c, d_g_word, d_g_count, (u32)n_unique, word_nt, distance, method, s, nps,
                    n_edges ? d_edges : &no_edges, n_edges));
    TRY(n_clusters_from_scan(c, (u32)n_unique, &c->C));
    s.clusters = c->C;
    if (d_cluster_id) *d_cluster_id = c->cid.as<u32>();
    if (d_is_max) *d_is_max = c->ismax.as<u8>();
  }
  HIPCHK(hipStreamSynchronize(c->stream));
  if (summary) *summary = s;
  c->have_graph = true;
  return HUMID_OK;
}

// ---- multi-GPU result return ------------------------------------------------------------------
int humid_stage_owned_results(humid_ctx *c, const uint32_t *d_local_cluster_id, const uint8_t *d_local_is_max,
                              const uint64_t *shard_begin, uint32_t n_shards, const uint32_t **d_packed,
                              uint64_t *counts) {
  if (!c) return fail(nullptr, HUMID_E_INVALID, "ctx is null");
  if (c->last_count_lds) return fail(c, HUMID_E_STATE, "owned results need the global-table count variant (count_mode 1)");
  if (!shard_begin || !counts || !d_packed || n_shards == 0 || n_shards > 4096) return fail(c, HUMID_E_INVALID, "bad argument");
  HIPCHK(hipSetDevice(c->device));
  hipStream_t st = c->stream;
  const u32 N = (u32)c->N, U = (u32)c->U;
  *d_packed = nullptr;
  for (u32 q = 0; q < n_shards; q++) counts[q] = 0;
  if (N == 0) return HUMID_OK;
  for (u32 q = 0; q <= n_shards; q++)
    if (shard_begin[q] > N || (q && shard_begin[q] < shard_begin[q - 1])) return fail(c, HUMID_E_INVALID, "shard_begin must ascend within [0, n_reads]");
  if (U && (!d_local_cluster_id || !d_local_is_max)) return fail(c, HUMID_E_INVALID, "null buffer");
  if (U > 0)
    hipLaunchKernelGGL(k_slot_results, dim3(blocks_for(U)), dim3(256), 0, st, d_local_cluster_id, d_local_is_max,
                       c->s_first.as<u32>(), c->s_slot.as<u32>(), U, c->slot_out.as<u64>());
  ENSURE(c->opos, ((size_t)N + 1) * 4);
  {
    auto fin = rocprim::make_transform_iterator(rocprim::counting_iterator<u32>(0),
                                                OwnedFlagOp{c->slot_of_read.as<u32>(), N});
    size_t bytes = 0;
    HIPCHK(rocprim::exclusive_scan(nullptr, bytes, fin, c->opos.as<u32>(), 0u, (size_t)N + 1, rocprim::plus<u32>(), st));
    ENSURE(c->tmp, bytes);
    HIPCHK(rocprim::exclusive_scan(c->tmp.p, bytes, fin, c->opos.as<u32>(), 0u, (size_t)N + 1, rocprim::plus<u32>(), st));
  }
  // per-shard counts: opos at the shard boundaries (a handful of 4-byte copies, one sync)
  std::vector<u32> got(n_shards + 1);
  for (u32 q = 0; q <= n_shards; q++)
    HIPCHK(hipMemcpyAsync(&got[q], c->opos.as<u32>() + shard_begin[q], 4, hipMemcpyDeviceToHost, st));
  u32 total = 0;
  HIPCHK(hipMemcpyAsync(&total, c->opos.as<u32>() + N, 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  // reads outside [shard_begin[0], shard_begin[n_shards]) must not be owned
  if (got[0] != 0 || got[n_shards] != total) return fail(c, HUMID_E_INVALID, "owned reads outside the shard table");
  for (u32 q = 0; q < n_shards; q++) counts[q] = got[q + 1] - got[q];
  ENSURE(c->own_packed, ((size_t)total + 1) * 4);
  if (total)
    hipLaunchKernelGGL(k_owned_results, dim3(grid_stride_blocks(N)), dim3(256), 0, st, c->slot_of_read.as<u32>(),
                       c->opos.as<u32>(), c->slot_out.as<u64>(), N, c->own_packed.as<u32>());
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(st));
  *d_packed = c->own_packed.as<u32>();
  return HUMID_OK;
}

int humid_stage_owner_perm(humid_ctx *c, const uint64_t *d_words, const uint8_t *d_filtered, uint64_t n_reads,
                           const uint64_t *range_lo, const uint64_t *range_hi, uint32_t n_ranks,
                           const uint32_t **d_perm, uint64_t *counts) {
  if (!c) return fail(nullptr, HUMID_E_INVALID, "ctx is null");
  if (!range_lo || !range_hi || !counts || !d_perm || n_ranks == 0) return fail(c, HUMID_E_INVALID, "bad argument");
  if (n_ranks > MAX_RANKS) return fail(c, HUMID_E_UNSUPPORTED, "more than %d ranks", MAX_RANKS);
  if (n_reads > 0x7fffffffull) return fail(c, HUMID_E_OVERFLOW, "n_reads exceeds 2^31-1");
  HIPCHK(hipSetDevice(c->device));
  hipStream_t st = c->stream;
  const u32 n = (u32)n_reads;
  *d_perm = nullptr;
  for (u32 q = 0; q < n_ranks; q++) counts[q] = 0;
  if (n == 0) return HUMID_OK;
  if (!d_words || !d_filtered) return fail(c, HUMID_E_INVALID, "null buffer");
  OwnerRanges rg;
  for (u32 q = 0; q < MAX_RANKS; q++) { rg.lo[q] = q < n_ranks ? range_lo[q] : 1; rg.hi[q] = q < n_ranks ? range_hi[q] : 0; }
  ENSURE(c->owner, (size_t)n);
  ENSURE(c->owner_sorted, (size_t)n);
  ENSURE(c->perm, (size_t)n * 4);
  hipLaunchKernelGGL(k_owner_of, dim3(blocks_for(n)), dim3(256), 0, st, d_words, d_filtered, n, rg, n_ranks,
                     c->owner.as<u8>());
  {
    using part_cfg = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                                rocprim::default_config, 0>;   // never the merge path
    rocprim::counting_iterator<u32> vin(0);
    size_t bytes = 0;
    HIPCHK(rocprim::radix_sort_pairs<part_cfg>(nullptr, bytes, c->owner.as<u8>(), c->owner_sorted.as<u8>(), vin,
                                               c->perm.as<u32>(), (size_t)n, 0, 8, st));
    ENSURE(c->tmp, bytes);
    HIPCHK(rocprim::radix_sort_pairs<part_cfg>(c->tmp.p, bytes, c->owner.as<u8>(), c->owner_sorted.as<u8>(), vin,
                                               c->perm.as<u32>(), (size_t)n, 0, 8, st));
  }
  ENSURE(c->small, (size_t)(n_ranks + 2) * 4);
  hipLaunchKernelGGL(k_owner_bounds, dim3(1), dim3(64), 0, st, c->owner_sorted.as<u8>(), n, n_ranks,
                     c->small.as<u32>());
  std::vector<u32> b(n_ranks + 2);
  HIPCHK(hipMemcpyAsync(b.data(), c->small.p, (n_ranks + 2) * 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(st));
  for (u32 q = 0; q < n_ranks; q++) counts[q] = b[q + 1] - b[q];
  *d_perm = c->perm.as<u32>();
  return HUMID_OK;
}

int humid_stage_scatter(humid_ctx *c, const uint32_t *d_perm, const uint32_t *d_packed, uint64_t n_recv,
                        uint64_t n_reads, uint32_t *d_cluster_id, uint8_t *d_keep) {
  if (!c) return fail(nullptr, HUMID_E_INVALID, "ctx is null");
  if (n_recv > n_reads) return fail(c, HUMID_E_INVALID, "n_recv > n_reads");
  HIPCHK(hipSetDevice(c->device));
  hipStream_t st = c->stream;
  if (n_reads) {
    if (!d_cluster_id || !d_keep) return fail(c, HUMID_E_INVALID, "null buffer");
    HIPCHK(hipMemsetAsync(d_cluster_id, 0, (size_t)n_reads * 4, st));
    HIPCHK(hipMemsetAsync(d_keep, 0, (size_t)n_reads, st));
  }
  if (n_recv) {
    if (!d_perm || !d_packed) return fail(c, HUMID_E_INVALID, "null buffer");
    hipLaunchKernelGGL(k_scatter_results, dim3(grid_stride_blocks(n_recv)), dim3(256), 0, st, d_perm, d_packed,
                       (u32)n_recv, d_cluster_id, d_keep);
  }
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(st));
  return HUMID_OK;
}

}  // extern "C"
