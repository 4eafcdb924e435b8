// kernels_wide.hip.h -- exact counts for wide words (33 <= n <= 64 nucleotides, two uint64 each)
// Part of libhumid_hip.so (see humid_hip.hip for the pipeline and the C ABI).  Device code for
// gfx950 only; included once, in this order, by humid_hip.hip.
//
// A 128-bit word does not fit the 64-bit LDS / HBM hash tables of kernels_count.hip.h, so the wide
// path counts by SORTING: two stable LSD radix passes (lo, then hi with the filtered flag on top)
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
__global__ void __launch_bounds__(256)
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
__global__ void __launch_bounds__(256)
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
__global__ void __launch_bounds__(256)
k_wide_keys_flag(const u8 *__restrict__ filtered, const u32 *__restrict__ v2, u32 n, u32 *__restrict__ key) {
  HUMID_GUARD_LAST_VGPR();
  for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    key[i] = (filtered && filtered[v2[i]]) ? 1u : 0u;
}

// words in sorted order (one 16-byte gather per read)
__global__ void __launch_bounds__(256)
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
__global__ void __launch_bounds__(256)
k_wide_heads(const W2 *__restrict__ sw, u32 n, const ull *__restrict__ ctr, u32 *__restrict__ head) {
  HUMID_GUARD_LAST_VGPR();
  const u32 usable = (u32)ctr[CTR_USABLE];
  for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i <= n; i += gridDim.x * blockDim.x)
    head[i] = (i < usable && i < n && (i == 0 || !w_eq(sw[i], sw[i - 1]))) ? 1u : 0u;
}

// per position: rank of its word (pslot) and the read tag (vals, bit 31 = filtered); per run head:
// the unique word, its first read and the run start.  hpos = exclusive scan of head.
__global__ void __launch_bounds__(256)
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
__global__ void k_wide_counts(const u32 *__restrict__ start, u32 n_unique, u32 *__restrict__ s_cnt,
                              u32 *__restrict__ s_slot) {
  HUMID_GUARD_LAST_VGPR();
  u32 u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u < n_unique) { s_cnt[u] = start[u + 1] - start[u]; s_slot[u] = u; }
}

#endif  // HUMID_KERNELS_WIDE_HIP_H
