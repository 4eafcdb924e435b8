// kernels_wide.hip.h -- exact counts for wide words (33 <= n <= 64 nucleotides, two uint64 each)
// Part of libhumid_hip.so (see humid_hip.hip for the pipeline and the C ABI).  Device code for
// gfx950 only; included once, in this order, by humid_hip.hip.
//
// Two ways.  The default for evenly spread words is the LDS-table count at the end of this file
// (k_dedup_lds_wide, round 2).  The general one, and its fallback: a 128-bit word does not fit the 64-bit
// LDS / HBM hash tables of kernels_count.hip.h, so the path counts by SORTING: two stable LSD radix passes (lo, then hi with the filtered flag on top)
// put the reads in (filtered, hi, lo, read index) order.  Runs of equal words are the leaves
// (Trie::add, /root/reference/src/humid.cc:95), their order is Trie::walk() order, the run length
// is the count and -- the sort being stable over ascending read indices -- the first element of a
// run is the leaf's first read.  The sorted order doubles as the "partition order" that
// k_read_map_part (kernels_map.hip.h) walks, so stage C is shared with the one-word path.
#ifndef HUMID_KERNELS_WIDE_HIP_H
#define HUMID_KERNELS_WIDE_HIP_H

#include "common.hip.h"
#include "kernels_count.hip.h"

// pass 1 input: key = lo, value = read index; also counts the usable reads (one atomic per block)
static __global__ void __launch_bounds__(256)
k_wide_keys_lo(const W2 *__restrict__ words, const u8 *__restrict__ filtered, u32 n, u64 *__restrict__ key,
               u32 *__restrict__ val, ull *ctr) {
  HUMID_GUARD_LAST_VGPR();
  __shared__ u32 lds[8];
  u32 usable = 0;
  for (u32 r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x) {
    key[r] = words[r].lo;
    val[r] = r;
    usable += (filtered && filtered[r]) ? 0u : 1u;
  }
  const u32 t = block_sum(usable, lds);
  if (threadIdx.x == 0 && t) atomicAdd(&ctr[CTR_USABLE], (ull)t);
}

// pass 2 input: key = hi (hbits = 2(n-32) significant bits), filtered reads above every word when
// there is a spare bit (hbits < 64; n = 64 takes a third one-bit pass instead)
static __global__ void __launch_bounds__(256)
k_wide_keys_hi(const W2 *__restrict__ words, const u8 *__restrict__ filtered, const u32 *__restrict__ v1, u32 n,
               u32 hbits, u64 *__restrict__ key) {
  HUMID_GUARD_LAST_VGPR();
  const u64 hmask = hbits >= 64 ? ~0ull : ((1ull << hbits) - 1ull);
  for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const u32 r = v1[i];
    u64 k = words[r].hi & hmask;
    if (hbits < 64 && filtered && filtered[r]) k = 1ull << hbits;
    key[i] = k;
  }
}

// pass 3 input (n = 64 only): key = filtered flag
static __global__ void __launch_bounds__(256)
k_wide_keys_flag(const u8 *__restrict__ filtered, const u32 *__restrict__ v2, u32 n, u32 *__restrict__ key) {
  HUMID_GUARD_LAST_VGPR();
  for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    key[i] = (filtered && filtered[v2[i]]) ? 1u : 0u;
}

// words in sorted order (one 16-byte gather per read)
static __global__ void __launch_bounds__(256)
k_wide_gather(const W2 *__restrict__ words, const u32 *__restrict__ v, u32 n, u32 hbits, W2 *__restrict__ sw) {
  HUMID_GUARD_LAST_VGPR();
  const u64 hmask = hbits >= 64 ? ~0ull : ((1ull << hbits) - 1ull);
  for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    W2 w = words[v[i]];
    w.hi &= hmask;
    sw[i] = w;
  }
}

// head[i] = 1 where a new unique word starts among the usable reads (the first ctr[CTR_USABLE]
// positions); head[n] = 0 is the sentinel of the exclusive scan
static __global__ void __launch_bounds__(256)
k_wide_heads(const W2 *__restrict__ sw, u32 n, const ull *__restrict__ ctr, u32 *__restrict__ head) {
  HUMID_GUARD_LAST_VGPR();
  const u32 usable = (u32)ctr[CTR_USABLE];
  for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i <= n; i += gridDim.x * blockDim.x)
    head[i] = (i < usable && i < n && (i == 0 || !w_eq(sw[i], sw[i - 1]))) ? 1u : 0u;
}

// per position: rank of its word (pslot) and the read tag (vals, bit 31 = filtered); per run head:
// the unique word, its first read and the run start.  hpos = exclusive scan of head.
static __global__ void __launch_bounds__(256)
k_wide_unique(const W2 *__restrict__ sw, const u32 *v, const u32 *__restrict__ head,
              const u32 *__restrict__ hpos, u32 n, const ull *__restrict__ ctr, W2 *__restrict__ s_word,
              u32 *__restrict__ s_first, u32 *__restrict__ start, u32 *__restrict__ pslot, u32 *vals) {
  HUMID_GUARD_LAST_VGPR();
  const u32 usable = (u32)ctr[CTR_USABLE];
  for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const u32 r = v[i];
    if (i < usable) {
      const u32 h = head[i];
      const u32 rank = hpos[i] + h - 1u;
      pslot[i] = rank;
      vals[i] = r;
      if (h) { s_word[rank] = sw[i]; s_first[rank] = r; start[rank] = i; }
    } else {
      pslot[i] = NOSLOT;
      vals[i] = r | 0x80000000u;
    }
    if (i == 0) start[hpos[n]] = usable < n ? usable : n;
  }
}

// count = run length; the slot of leaf u is u itself
static __global__ void k_wide_counts(const u32 *__restrict__ start, u32 n_unique, u32 *__restrict__ s_cnt,
                              u32 *__restrict__ s_slot) {
  HUMID_GUARD_LAST_VGPR();
  u32 u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u < n_unique) { s_cnt[u] = start[u + 1] - start[u]; s_slot[u] = u; }
}

// --------------------------------------------------------------------------------
// Wide words in LDS tables (round 2).  The sort above moves every read through five passes of a
// 64-bit radix sort; when the top 64 bits of the words (their "heads", k_wide_head64) spread evenly
// -- the same test as for one-word words, prefix_fits_ordered -- the reads are bucketed by their head
// with the two-level tile partition of kernels_part.hip.h exactly as one-word words are, and each
// bucket is counted by one workgroup here.  A 128-bit word cannot be claimed with one LDS compare-and-
// swap, so a table entry holds the POSITION (inside the bucket) of the read that claimed it: the
// bucket's words are staged in LDS by position before the first insert, an entry's word is the staged
// word of its claimer, and a probe compares against that.  Outputs and their layout are those of
// k_dedup_lds<true> with W2 words: unique words of a bucket ascending (hi, lo) in the bucket's padded
// room -- buckets are runs of the head order, so squeezing out the holes gives Trie::walk() order.
// A bucket of more than WL_STAGE reads does not fit the staging array: CTR_OVERFULL, and the caller
// counts by sorting instead.
// --------------------------------------------------------------------------------
#define WL_STAGE 1024u            // the longest bucket the large variant stages
#define WL_SMALL_LEN 384u         // buckets up to this many reads: the small variant (512 entries at <= 75 % load)

// Two size classes, launched one after the other over all buckets (a workgroup whose bucket belongs to the
// other class leaves at once): SB = 9, STAGE = 512 for buckets of up to WL_SMALL_LEN reads -- nearly all of
// them at ~300 reads per bucket -- needs 17 KB of LDS, so eight workgroups share a CU as in k_dedup_lds;
// SB = 10, STAGE = 1024 (33 KB, four per CU) takes the rest.
template <u32 SB, u32 STAGE, u32 LEN_MIN, u32 LEN_MAX>
__global__ void __launch_bounds__(256)
k_dedup_lds_wide(const u64 *__restrict__ keys, const u32 *__restrict__ vals, const u32 *__restrict__ pbeg,
                 const W2 *__restrict__ words, u32 hbits, u32 n_reads, u32 pb, W2 *__restrict__ pad_word,
                 uint2 *__restrict__ pad_cf, u32 *__restrict__ ucount, u32 *__restrict__ pusable,
                 u32 *__restrict__ pslot, ull *ctr) {
  HUMID_GUARD_LAST_VGPR();
  constexpr u32 SLOTS = 1u << SB, Q = STAGE / 256u;
  static_assert(STAGE % 256u == 0 && (STAGE & (STAGE - 1)) == 0 && LEN_MAX <= STAGE && LEN_MAX <= SLOTS, "size class");
  __shared__ W2 wk[STAGE];                             // the bucket's words by position
  __shared__ u32 ltag[SLOTS];                          // position of the claiming read, NONE32 = empty
  __shared__ u32 lcnt[SLOTS];
  __shared__ u32 lfirst[SLOTS];
  __shared__ unsigned short lslot_of[SLOTS];           // unique index (claim order, then rank) -> table entry
  __shared__ unsigned short lpos[SLOTS];               // unique index -> position of its claimer
  __shared__ unsigned short lorder[512];
  __shared__ u32 lcount;
  __shared__ u32 lds[8];
  const u32 b = blockIdx.x;
  const u32 beg = pbeg[b], end = pbeg[b + 1];
  if (beg >= end || end > n_reads || end - beg > WL_STAGE) {
    // empty, malformed or too long for any class: reported once, by the class that starts at 0
    if (LEN_MIN == 0 && threadIdx.x == 0) {
      if (beg > end || end > n_reads || (beg < end && end - beg > WL_STAGE)) ctr[CTR_OVERFULL] = 1;
      ucount[b] = 0;
      pusable[b] = 0;
    }
    return;
  }
  const u32 len = end - beg;
  if (len <= LEN_MIN || len > LEN_MAX) return;         // the other class's bucket
  const u64 hmask = hbits >= 64 ? ~0ull : ((1ull << hbits) - 1ull);
  u64 kq[Q];
  u32 vq[Q];
  W2 wq[Q];
#pragma unroll
  for (u32 q = 0; q < Q; q++) {
    const u32 p = threadIdx.x + 256u * q;
    vq[q] = NONE32;
    if (p < len) { vq[q] = vals[beg + p]; kq[q] = keys[beg + p]; }
  }
  for (u32 s = threadIdx.x; s < SLOTS; s += 256) { ltag[s] = NONE32; lcnt[s] = 0; lfirst[s] = NONE32; }
  if (threadIdx.x == 0) { lcount = 0; lds[0] = 0; }
  bool overflow = false;
#pragma unroll
  for (u32 q = 0; q < Q; q++) {                        // the words themselves: one 16-byte gather per read
    const u32 p = threadIdx.x + 256u * q;
    if (p < len) {
      W2 w{0, 0};
      if (vq[q] < n_reads) { w = words[vq[q]]; w.hi &= hmask; }
      wq[q] = w;
      wk[p] = w;
    }
  }
  __syncthreads();
  const u32 hshift = 64 - pb - SB;
  u32 usable = 0;
#pragma unroll
  for (u32 q = 0; q < Q; q++) {
    const u32 p = threadIdx.x + 256u * q;
    if (p >= len || overflow) continue;
    const u32 v = vq[q];
    if ((v & 0x7fffffffu) >= n_reads) { overflow = true; continue; }     // a malformed index is never used
    if (v & 0x80000000u) { pslot[beg + p] = NOSLOT; continue; }
    usable++;
    const W2 w = wq[q];
    u32 s = (u32)(kq[q] >> hshift) & (SLOTS - 1);
    u32 probes = 0;
    bool placed = false;
    while (probes++ <= SLOTS) {
      u32 cur = ltag[s];
      if (cur == NONE32) cur = atomicCAS(&ltag[s], NONE32, p);
      if (cur == NONE32 || w_eq(wk[cur & (STAGE - 1)], w)) { placed = true; break; }
      s = (s + 1) & (SLOTS - 1);
    }
    if (!placed) { overflow = true; continue; }
    if (atomicAdd(&lcnt[s], 1u) == 0u) lslot_of[atomicAdd(&lcount, 1u)] = (unsigned short)s;
    atomicMin(&lfirst[s], v);
  }
  if (overflow) ctr[CTR_OVERFULL] = 1;
  {
    u32 x = usable;
#pragma unroll
    for (u32 d = 32; d >= 1; d >>= 1) x += __shfl_xor(x, d);
    if ((threadIdx.x & 63) == 0 && x) atomicAdd(&lds[0], x);
  }
  __syncthreads();
  const u32 n_uniq = lcount < SLOTS ? lcount : SLOTS;
  for (u32 li = threadIdx.x; li < n_uniq; li += 256) lpos[li] = (unsigned short)ltag[lslot_of[li]];
  __syncthreads();
  if (n_uniq <= 512) {
    // rank of an entry = number of smaller words among the bucket's unique words (all distinct)
    for (u32 li = threadIdx.x; li < n_uniq; li += 256) {
      const W2 w = wk[lpos[li]];
      u32 r = 0;
      for (u32 j = 0; j < n_uniq; j++) r += w_less(wk[lpos[j]], w) ? 1u : 0u;
      lorder[r] = lslot_of[li];
    }
    __syncthreads();
    for (u32 li = threadIdx.x; li < n_uniq; li += 256) lslot_of[li] = lorder[li];
    __syncthreads();
  } else if (LEN_MAX > 512) {
    // bitonic network over the claim-order list, keys through the claimer's staged word
    u32 npow = 1;
    while (npow < n_uniq) npow <<= 1;
    for (u32 i = n_uniq + threadIdx.x; i < npow; i += 256) lslot_of[i] = 0xffff;      // padding: above every word
    __syncthreads();
    for (u32 k = 2; k <= npow; k <<= 1) {
      for (u32 j = k >> 1; j > 0; j >>= 1) {
        for (u32 t = threadIdx.x; t < npow; t += 256) {
          const u32 x = t ^ j;
          if (x > t) {
            const u32 a = lslot_of[t], bb = lslot_of[x];
            const bool pa = a == 0xffff, pbd = bb == 0xffff;
            bool gt;
            if (pa) gt = !pbd;
            else if (pbd) gt = false;
            else gt = w_less(wk[ltag[bb] & (STAGE - 1)], wk[ltag[a] & (STAGE - 1)]);
            if (gt == ((t & k) == 0)) { lslot_of[t] = (unsigned short)bb; lslot_of[x] = (unsigned short)a; }
          }
        }
        __syncthreads();
      }
    }
  }
  for (u32 li = threadIdx.x; li < n_uniq; li += 256) {
    const u32 s = lslot_of[li];
    pad_word[beg + li] = wk[ltag[s] & (STAGE - 1)];
    pad_cf[beg + li] = make_uint2(lcnt[s], lfirst[s]);
    lfirst[s] = li;
  }
  if (threadIdx.x == 0) { ucount[b] = n_uniq; pusable[b] = lds[0]; }
  __syncthreads();
  // second pass: every position learns the padded slot of its word
#pragma unroll
  for (u32 q = 0; q < Q; q++) {
    const u32 p = threadIdx.x + 256u * q;
    if (p >= len || vq[q] >= n_reads) continue;        // excluded read (bit 31) or malformed index
    const W2 w = wq[q];
    u32 s = (u32)(kq[q] >> hshift) & (SLOTS - 1);
    u32 probes = 0, li = NONE32;
    while (probes++ <= SLOTS) {
      const u32 t = ltag[s];
      if (t == NONE32) break;                          // (cannot happen for a word that was inserted)
      if (w_eq(wk[t & (STAGE - 1)], w)) { li = lfirst[s]; break; }
      s = (s + 1) & (SLOTS - 1);
    }
    pslot[beg + p] = (li < len) ? beg + li : NOSLOT;
  }
}

// padded -> dense unique list of wide words, one wave per bucket (k_compact_padded<true> with W2 words)
static __global__ void __launch_bounds__(256)
k_compact_padded_wide(const W2 *__restrict__ pad_word, const uint2 *__restrict__ pad_cf,
                      const u32 *__restrict__ pbeg, const u32 *__restrict__ ucount,
                      const u32 *__restrict__ ubase, u32 n_parts, W2 *__restrict__ uniq_word,
                      u32 *__restrict__ uniq_slot, u32 *__restrict__ s_cnt, u32 *__restrict__ s_first) {
  HUMID_GUARD_LAST_VGPR();
  const u32 wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const u32 lane = threadIdx.x & 63;
  if (wave >= n_parts) return;
  const u32 beg = pbeg[wave], uc = ucount[wave], ub = ubase[wave];
  for (u32 j = lane; j < uc; j += 64) {
    uniq_word[ub + j] = pad_word[beg + j];
    uniq_slot[ub + j] = beg + j;
    const uint2 cf = pad_cf[beg + j];
    s_cnt[ub + j] = cf.x;
    s_first[ub + j] = cf.y;
  }
}

#endif  // HUMID_KERNELS_WIDE_HIP_H
