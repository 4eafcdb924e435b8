// kernels_count.hip.h -- exact counts: global HBM table and hash-partitioned LDS tables (Trie::add)
// Part of libhumid_hip.so (see humid_hip.hip for the pipeline and the C ABI).  Device code for
// gfx950 only; included once, in this order, by humid_hip.hip.
#ifndef HUMID_KERNELS_COUNT_HIP_H
#define HUMID_KERNELS_COUNT_HIP_H

#include "common.hip.h"

// --------------------------------------------------------------------------------
// 0. word packing on the device (makeWord, /root/reference/src/fastq.cc:146-161)
// --------------------------------------------------------------------------------
// bases[n_reads][word_nt]: the symbols getNucleotides (src/fastq.cc:116-144) assembled for every
// record -- header UMI, then the leading bases of every file's read, 'N' where a read was short --
// as the ASCII the FastQ holds.  A -> 0, C -> 1, G -> 2, T -> 3; any other byte -> the code of 'G'
// and the word is filtered (src/fastq.cc:151-158).  First symbol in the most significant bits.
// A workgroup copies its 256 rows into LDS with coalesced 4-byte loads, then every thread packs one.
template <bool WIDE>
__global__ void __launch_bounds__(256)
k_pack_bases(const u8 *__restrict__ bases, u32 n_reads, u32 word_nt, u64 *__restrict__ words, u8 *__restrict__ filtered) {
  HUMID_GUARD_LAST_VGPR();
  __shared__ u32 rows[256 * 64 / 4];
  const u32 r0 = blockIdx.x * 256;
  const u32 n_rows = (n_reads - r0 < 256) ? n_reads - r0 : 256;
  const size_t byte0 = (size_t)r0 * word_nt;
  const u32 n_bytes = n_rows * word_nt;
  // the tile starts at byte0, which need not be 4-aligned: copy from the aligned address below it
  const u32 skew = (u32)((uintptr_t)(bases + byte0) & 3);
  const u32 *src = (const u32 *)(bases + byte0 - skew);
  const u32 n_dw = (skew + n_bytes + 3) / 4;
  for (u32 k = threadIdx.x; k < n_dw; k += 256) rows[k] = src[k];
  __syncthreads();
  if (threadIdx.x >= n_rows) return;
  const u8 *row = (const u8 *)rows + skew + threadIdx.x * word_nt;
  u64 hi = 0, lo = 0;
  bool filt = false;
  for (u32 i = 0; i < word_nt; i++) {
    const u32 ch = row[i];
    u32 code = ((ch >> 1) ^ (ch >> 2)) & 3u;                  // A 0, C 1, G 2, T 3
    if (ch != 'A' && ch != 'C' && ch != 'G' && ch != 'T') { code = 2; filt = true; }
    if (WIDE) hi = (hi << 2) | (lo >> 62);
    lo = (lo << 2) | code;
  }
  if (WIDE) { words[2 * (size_t)(r0 + threadIdx.x)] = hi; words[2 * (size_t)(r0 + threadIdx.x) + 1] = lo; }
  else words[r0 + threadIdx.x] = lo;
  filtered[r0 + threadIdx.x] = filt ? 1 : 0;
}

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

static __global__ void __launch_bounds__(256)
k_hash_insert(const u64 *__restrict__ words, const u8 *__restrict__ filtered, u32 n_reads,
              Slot *tab, u32 cap_log2, u32 *__restrict__ slot_of_read, u64 range_lo, u64 range_hi,
              u32 max_probe, ull *ctr) {
  HUMID_GUARD_LAST_VGPR();
  const u32 mask = (1u << cap_log2) - 1u;
  const u32 cap = 1u << cap_log2;
  for (u32 r = blockIdx.x * blockDim.x + threadIdx.x; r < n_reads; r += gridDim.x * blockDim.x) {
    if (filtered && filtered[r]) { slot_of_read[r] = NOSLOT; continue; }   // filtered == null: none
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
static __global__ void __launch_bounds__(256)
k_compact_table(const Slot *__restrict__ tab, u32 n_slots, u64 *__restrict__ uniq_word,
                u32 *__restrict__ uniq_slot, u32 uniq_cap, ull *ctr) {
  HUMID_GUARD_LAST_VGPR();
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
#define LDS_SLOT_BITS 10u
#define LDS_SLOTS (1u << LDS_SLOT_BITS)   // 16-byte entries: 16 KiB of LDS per workgroup
#define LDS_FILL_LIMIT (3u * LDS_SLOTS / 4u)   // unique words a bucket may hold (75 % load)
#define PART_TARGET 350u         // mean reads per bucket

// keys_input transform: word -> partition key.  Hashed: mix64(word) (a bijection; robust to any
// word distribution).  Ordered: (word - lo) * scale, a strictly increasing map of the value range
// [lo, hi] the reads lie in onto the whole 64-bit key space, so that buckets are runs of the word
// order -- used when the words are uniform over the range (UMI first), see stage_count.  A single
// GPU has lo = 0, scale = 2^(64 - 2n): the left-aligned word.
struct PartKeyOp {
  u32 ordered;
  u64 lo, scale;
  __host__ __device__ u64 operator()(u64 w) const { return ordered ? (w - lo) * scale : mix64(w); }
};
struct ReadTagOp {               // values_input transform: read index | excluded << 31
  const u64 *words;              // null: no value range to check (one GPU), the word is not loaded
  const u8 *filtered;
  u64 lo, hi;
  __device__ u32 operator()(u32 r) const {
    bool excl = filtered && filtered[r] != 0;                                // filtered == null: none
    if (words) { const u64 w = words[r]; excl = excl || w < lo || w > hi; }
    return r | (excl ? 0x80000000u : 0u);
  }
};

// first position of every bucket in the partitioned key array (binary search)
static __global__ void k_part_bounds(const u64 *__restrict__ keys, u32 n, u32 pb, u32 n_parts, u32 *__restrict__ pbeg,
                              u32 *__restrict__ ucount) {
  HUMID_GUARD_LAST_VGPR();
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
// ORDERED: the partition key is the left-aligned word; the bucket's unique words are additionally
// sorted (bitonic sort of the claim-order index by key, in LDS), so the padded arrays are in
// word order bucket after bucket = Trie::walk() order once the holes are squeezed out.
template <bool ORDERED>
__global__ void __launch_bounds__(256)
k_dedup_lds(const u64 *__restrict__ keys, const u32 *__restrict__ vals, const u32 *__restrict__ pbeg,
            u32 n_reads, u32 pb, u64 klo, u64 kscale, u32 kshift, u64 *__restrict__ pad_word,
            uint2 *__restrict__ pad_cf,
            u32 *__restrict__ ucount, u32 *__restrict__ pusable, u32 *__restrict__ pslot, ull *ctr) {
  HUMID_GUARD_LAST_VGPR();
  __shared__ u64 lkey[LDS_SLOTS + 1];
  __shared__ u32 lcnt[LDS_SLOTS + 1];
  __shared__ u32 lfirst[LDS_SLOTS + 1];
  __shared__ unsigned short lslot_of[LDS_SLOTS + 1];   // unique index (claim order) -> table entry
  __shared__ unsigned short lorder[ORDERED ? (LDS_SLOTS < 512 ? LDS_SLOTS : 512) : 1];   // ORDERED, small buckets: entries by rank
  __shared__ u32 lcount;                               // unique words claimed so far
  __shared__ u32 lds[8];
  const u32 b = blockIdx.x;
  const u32 beg = pbeg[b], end = pbeg[b + 1];
  if (beg >= end || end > n_reads) {
    if (beg > end || end > n_reads) ctr[CTR_OVERFULL] = 1;   // malformed partition: never index with it
    if (threadIdx.x == 0) { ucount[b] = 0; pusable[b] = 0; }
    return;
  }
  // a thread's first four positions are requested before the table is cleared (their latency hides
  // behind the clearing and its barrier: buckets hold ~350 reads, i.e. 1-2 positions per thread) and
  // stay in registers for the second pass below; the rest (buckets beyond 1024 reads) one by one
  u64 kq[4];
  u32 vq[4];
#pragma unroll
  for (u32 q = 0; q < 4; q++) {
    const u32 i = beg + threadIdx.x + 256u * q;
    if (i < end) { vq[q] = vals[i]; kq[q] = keys[i]; }
  }
  for (u32 s = threadIdx.x; s <= LDS_SLOTS; s += 256) { lkey[s] = EMPTY_KEY; lcnt[s] = 0; lfirst[s] = NONE32; }
  if (threadIdx.x == 0) { lcount = 0; lds[0] = 0; }
  __syncthreads();
  const u32 hshift = 64 - pb - LDS_SLOT_BITS;   // table index = the key bits just below the bucket bits
  u32 usable = 0;
  bool overflow = false;
  // one read into the table; false = the table is full / the index is malformed
  auto insert = [&](u32 i, u32 v, u64 k) -> bool {
    if ((v & 0x7fffffffu) >= n_reads) return false;                   // a malformed index is never used
    if (v & 0x80000000u) { pslot[i] = NOSLOT; return true; }
    usable++;
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
        if (++probes >= LDS_SLOTS) return false;
      }
    }
    // the first add to an entry (old count 0) registers it: its index is the claim order, so no
    // compaction scan over the table is needed afterwards
    if (atomicAdd(&lcnt[s], 1u) == 0u) lslot_of[atomicAdd(&lcount, 1u)] = (unsigned short)s;
    atomicMin(&lfirst[s], v);
    return true;
  };
#pragma unroll
  for (u32 q = 0; q < 4; q++) {
    const u32 i = beg + threadIdx.x + 256u * q;
    if (i < end && !overflow && !insert(i, vq[q], kq[q])) overflow = true;
  }
  for (u32 i = beg + threadIdx.x + 1024u; i < end && !overflow; i += 256)
    if (!insert(i, vals[i], keys[i])) overflow = true;
  if (overflow) ctr[CTR_OVERFULL] = 1;
  {                                           // usable reads of the bucket: one LDS add per wave, before the barrier
    u32 x = usable;
#pragma unroll
    for (u32 d = 32; d >= 1; d >>= 1) x += __shfl_xor(x, d);
    if ((threadIdx.x & 63) == 0 && x) atomicAdd(&lds[0], x);
  }
  __syncthreads();
  // registered entries -> padded arrays (index < unique words <= reads of the bucket = padded
  // room); lfirst[s] is then reused as entry -> index
  const u32 n_uniq = lcount;
  if (ORDERED && n_uniq <= 512) {
    // small bucket (the normal case): rank of an entry = number of smaller keys, counted directly
    // -- n_uniq broadcast LDS reads per thread, no barrier per step (a bitonic network costs 36
    // barriers at 256 entries)
    for (u32 li = threadIdx.x; li < n_uniq; li += 256) {
      const u32 s = lslot_of[li];
      const u64 k = lkey[s];
      u32 r = 0;
      for (u32 j = 0; j < n_uniq; j++) r += (lkey[lslot_of[j]] < k) ? 1u : 0u;
      lorder[r] = (unsigned short)s;            // keys are distinct: ranks are a permutation
    }
    __syncthreads();
    for (u32 li = threadIdx.x; li < n_uniq; li += 256) lslot_of[li] = lorder[li];
    __syncthreads();
  } else if (ORDERED) {
    if (n_uniq > LDS_SLOTS) ctr[CTR_OVERFULL] = 1;             // table + special entry all in use
    u32 npow = 1;
    while (npow < n_uniq) npow <<= 1;
    if (npow > LDS_SLOTS) npow = LDS_SLOTS;
    for (u32 i = n_uniq + threadIdx.x; i < npow; i += 256) lslot_of[i] = 0xffff;   // padding: +infinity
    __syncthreads();
    for (u32 k = 2; k <= npow; k <<= 1) {
      for (u32 j = k >> 1; j > 0; j >>= 1) {
        for (u32 t = threadIdx.x; t < npow; t += 256) {
          const u32 x = t ^ j;
          if (x > t) {
            const u32 a = lslot_of[t], bb = lslot_of[x];
            const bool pa = a == 0xffff, pbd = bb == 0xffff;
            const u64 ka = pa ? 0 : lkey[a], kb = pbd ? 0 : lkey[bb];
            const bool gt = pa ? !pbd : (!pbd && ka > kb);     // a > b, padding above every key
            if (gt == ((t & k) == 0)) { lslot_of[t] = (unsigned short)bb; lslot_of[x] = (unsigned short)a; }
          }
        }
        __syncthreads();
      }
    }
  }
  for (u32 li = threadIdx.x; li < n_uniq; li += 256) {
    const u32 s = lslot_of[li];
    // ordered key -> word: kshift < 64 when the scale is a power of two (always on one GPU)
    pad_word[beg + li] = ORDERED ? klo + (kshift < 64 ? (lkey[s] >> kshift) : lkey[s] / kscale) : unmix64(lkey[s]);
    pad_cf[beg + li] = make_uint2(lcnt[s], lfirst[s]);
    lfirst[s] = li;
  }
  if (threadIdx.x == 0) { ucount[b] = n_uniq; pusable[b] = lds[0]; }
  __syncthreads();
  // second pass: every position learns the padded slot of its word (coalesced store; the
  // per-read outputs are produced later in this same partition order, see k_read_map_bucket)
  auto locate = [&](u32 i, u32 v, u64 k) {
    if (v >= n_reads) return;            // excluded read (bit 31) or malformed index
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
  };
#pragma unroll
  for (u32 q = 0; q < 4; q++) {
    const u32 i = beg + threadIdx.x + 256u * q;
    if (i < end) locate(i, vq[q], kq[q]);
  }
  for (u32 i = beg + threadIdx.x + 1024u; i < end; i += 256) locate(i, vals[i], keys[i]);
}

// totals over the buckets: U = sum ucount, usable = sum pusable (a few blocks, two atomics each;
// the counters were zeroed at the start of the stage)
static __global__ void __launch_bounds__(256)
k_part_totals(const u32 *__restrict__ ucount, const u32 *__restrict__ pusable, u32 n_parts, ull *ctr) {
  HUMID_GUARD_LAST_VGPR();
  __shared__ u32 lds[4];
  u32 u = 0, us = 0;                    // both totals are < 2^32 (reads < 2^31)
  for (u32 p = blockIdx.x * blockDim.x + threadIdx.x; p < n_parts; p += gridDim.x * blockDim.x) { u += ucount[p]; us += pusable[p]; }
  const u32 tu = block_sum(u, lds);
  const u32 ts = block_sum(us, lds);
  if (threadIdx.x == 0) {
    if (tu) atomicAdd(&ctr[CTR_UNIQUE], (ull)tu);
    if (ts) atomicAdd(&ctr[CTR_USABLE], (ull)ts);
  }
}

// padded -> dense unique list (word, padded position), one wave per bucket.  Hashed buckets: the
// list is in bucket order and sorted afterwards.  Ordered buckets: it IS walk order, so count and
// first read are copied along and no sort follows.
template <bool ORDERED>
__global__ void __launch_bounds__(256)
k_compact_padded(const u64 *__restrict__ pad_word, const uint2 *__restrict__ pad_cf,
                 const u32 *__restrict__ pbeg, const u32 *__restrict__ ucount,
                 const u32 *__restrict__ ubase, u32 n_parts, u64 *__restrict__ uniq_word,
                 u32 *__restrict__ uniq_slot, u32 *__restrict__ s_cnt, u32 *__restrict__ s_first) {
  HUMID_GUARD_LAST_VGPR();
  const u32 wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const u32 lane = threadIdx.x & 63;
  if (wave >= n_parts) return;
  const u32 beg = pbeg[wave], uc = ucount[wave], ub = ubase[wave];
  for (u32 j = lane; j < uc; j += 64) {
    uniq_word[ub + j] = pad_word[beg + j];
    uniq_slot[ub + j] = beg + j;
    if (ORDERED) {
      const uint2 cf = pad_cf[beg + j];
      s_cnt[ub + j] = cf.x;
      s_first[ub + j] = cf.y;
    }
  }
}

// after the sort, padded variant: gather count / first read of rank i (one 8-byte gather)
static __global__ void k_post_sort_padded(const u32 *__restrict__ s_slot, const uint2 *__restrict__ pad_cf, u32 n,
                                   u32 *__restrict__ s_cnt, u32 *__restrict__ s_first) {
  HUMID_GUARD_LAST_VGPR();
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const uint2 cf = pad_cf[s_slot[i]];
    s_cnt[i] = cf.x;
    s_first[i] = cf.y;
  }
}

// after the sort: per rank i gather count / first read from the table
static __global__ void k_post_sort(const u32 *__restrict__ s_slot, const Slot *__restrict__ tab, u32 n,
                            u32 *__restrict__ s_cnt, u32 *__restrict__ s_first) {
  HUMID_GUARD_LAST_VGPR();
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const Slot sl = tab[s_slot[i]];
    s_cnt[i] = sl.cntm1 + 1u;
    s_first[i] = sl.first;
  }
}


#endif  // HUMID_KERNELS_COUNT_HIP_H
