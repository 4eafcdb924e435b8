// kernels_part.hip.h -- LDS-staged multi-way partition: reads -> buckets (front of the exact counts)
// and per-read results -> read order (the un-permute at the end of the path)
// Part of libhumid_hip.so (see humid_hip.hip for the pipeline and the C ABI).  Device code for
// gfx950 only; included once, in this order, by humid_hip.hip.
//
// Both ends of the path move one small record per read to a place that depends on its value: the
// (key, read) pair into the bucket of its key, the (read, result) pair back to the position of the
// read.  Done as a plain scatter that is one random 4-12 byte store per read -- 32-byte write
// granules for 4 bytes of payload and half-filled requests (round 1: 0.22 ms and 3.4x the
// algorithmic bytes for the un-permute, 0.31 ms in library radix passes for the partition).  Here a
// workgroup takes a TILE of 8192 records, sorts it by destination bin inside LDS (counting sort:
// LDS atomics give every record its rank), reserves room in every bin with ONE global atomic per
// non-empty bin, and writes each bin's records as one contiguous run: hundreds of bytes per run at a
// fan-out of 128-512 bins.  Order inside a bin is not kept (nothing downstream needs it: the exact
// counts take the minimum read index of a word, the un-permute writes disjoint positions).
#ifndef HUMID_KERNELS_PART_HIP_H
#define HUMID_KERNELS_PART_HIP_H

#include "common.hip.h"
#include "kernels_count.hip.h"

#define PT_THREADS 1024u
#define PT_TILE 8192u            // records per tile: 8 per thread
#define PT_IPT (PT_TILE / PT_THREADS)
#define PT_MAXBINS 512u          // fan-out of one level (<= 9 key bits)
#define PT_MAXBINS1 1024u        // ... of the record path's FIRST level when a record would not fit 64 bits otherwise (10 bits)

// the reads of the count stage
struct PtInput {
  const u64 *words;
  const u8 *filtered;            // null: none filtered
  u64 rlo, rhi;                  // value range this rank counts (multi-GPU); check_range = 0: everything
  u32 check_range;
  PartKeyOp key;
};
__device__ __forceinline__ bool pt_load(const PtInput &in, u32 r, u64 &key) {
  if (in.filtered && in.filtered[r]) return false;
  const u64 w = in.words[r];
  if (in.check_range && (w < in.rlo || w > in.rhi)) return false;
  key = in.key(w);
  return true;
}
// A SOURCE of the partition says what item j of the input is (load: false = not partitioned; the 64-bit
// payload that travels with the item's index) and what 64-bit key, bins read from its top bits, a
// payload has.  ReadsSrc: the reads of the count stage, payload = the partition key itself.
struct ReadsSrc {
  PtInput in;
  __device__ __forceinline__ bool load(u32 j, u64 &payload) const { return pt_load(in, j, payload); }
  __device__ __forceinline__ u64 key(u64 payload) const { return payload; }
};
// Two-word words (33 <= n <= 64 nucleotides): the partition key is taken from the word's HEAD, the top 64
// bits of its 2n-bit value (hbits = 2 (n - 32) of them live in .hi), computed as the word is read.  Only
// the head's top 48 bits enter the key (WIDE_KEY_DROP low bits are shifted out, "a 24-nt word"): buckets
// and table homes need 28 bits at most, and a value range of 48-bit numbers stretches over the 64-bit key
// space with an integer factor of ~2^16 -- a range of 64-bit heads (a rank's share on 2 GPUs: half of
// them) would have to make do with a factor of 1 or 2 and leave up to half of the buckets empty.
#define WIDE_KEY_DROP 16u
struct WideReadsSrc {
  const W2 *words;
  const u8 *filtered;            // null: none filtered
  u32 hbits;
  PartKeyOp pkey;
  __device__ __forceinline__ bool load(u32 j, u64 &payload) const {
    if (filtered && filtered[j]) return false;
    const W2 x = words[j];
    payload = pkey((hbits >= 64 ? x.hi : ((x.hi << (64 - hbits)) | (x.lo >> hbits))) >> WIDE_KEY_DROP);
    return true;
  }
  __device__ __forceinline__ u64 key(u64 payload) const { return payload; }
};

// exclusive scan of cnt[0, nb) (nb <= 512) by the first 512 threads of the block -> off[0, nb],
// off[nb] = total.  All threads of the block must call it.
__device__ __forceinline__ void block_exscan_512(const u32 *cnt, u32 *off, u32 nb, u32 *wsum /* >= 8 */) {
  const u32 t = threadIdx.x, lane = t & 63, wv = t >> 6;
  u32 x = (t < nb) ? cnt[t] : 0u, incl = x;
  if (t < 512) {
    incl = wave_incl_scan(incl);
    if (lane == 63) wsum[wv] = incl;
  }
  __syncthreads();
  if (t < 512) {
    u32 before = 0;
    for (u32 k = 0; k < wv; k++) before += wsum[k];
    if (t < nb) off[t] = before + incl - x;
    if (t == nb - 1) off[nb] = before + incl;
  }
  __syncthreads();
}

// the same for nb <= 1024 (the 10-bit first level of the record path, kernels_part8.hip.h), by a block of 1024 threads
__device__ __forceinline__ void block_exscan_1024(const u32 *cnt, u32 *off, u32 nb, u32 *wsum /* >= 16 */) {
  const u32 t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const u32 x = (t < nb) ? cnt[t] : 0u;
  const u32 incl = wave_incl_scan(x);
  if (lane == 63) wsum[wv] = incl;
  __syncthreads();
  u32 before = 0;
  for (u32 k = 0; k < wv; k++) before += wsum[k];
  if (t < nb) off[t] = before + incl - x;
  if (t == nb - 1) off[nb] = before + incl;
  __syncthreads();
}

// tile -> (coarse bin, first record, record count): tiles never cross a coarse bin
// cap1 > 0: the level-1 output is PADDED -- coarse bin c owns the fixed room [c * cap1, (c + 1) * cap1) and
// holds cbase[c + 1] - cbase[c] records at its start (level 1 then needs no histogram pass of its own)
__device__ __forceinline__ void pt_tile_of(const u32 *__restrict__ tprefix, const u32 *__restrict__ cbase, u32 nb1,
                                           u32 tile, u32 cap1, u32 &c, u32 &beg, u32 &cnt) {
  u32 lo = 0, hi = nb1;                    // largest c with tprefix[c] <= tile
  while (hi - lo > 1) {
    const u32 mid = (lo + hi) >> 1;
    if (tprefix[mid] <= tile) lo = mid; else hi = mid;
  }
  c = lo;
  const u32 k = tile - tprefix[c];
  const u32 n_c = cbase[c + 1] - cbase[c], done = k * PT_TILE;
  beg = (cap1 ? c * cap1 : cbase[c]) + done;
  cnt = (done >= n_c) ? 0u : ((n_c - done < PT_TILE) ? n_c - done : PT_TILE);
}

// ---- level 1 histogram: usable reads per coarse bin (the top d1 key bits) ----
template <class SRC>
__global__ void __launch_bounds__(1024)
k_pt_hist1(SRC src, u32 n_reads, u32 d1, u32 *__restrict__ hist1) {
  HUMID_GUARD_LAST_VGPR();
  __shared__ u32 h[PT_MAXBINS];
  const u32 nb = 1u << d1;
  for (u32 b = threadIdx.x; b < nb; b += blockDim.x) h[b] = 0;
  __syncthreads();
  const u32 stride = gridDim.x * blockDim.x;
  u32 r = blockIdx.x * blockDim.x + threadIdx.x;
  for (; r + 3 * stride < n_reads; r += 4 * stride) {         // four independent loads in flight
    u64 key[4];
    bool ok[4];
#pragma unroll
    for (u32 q = 0; q < 4; q++) ok[q] = src.load(r + q * stride, key[q]);
#pragma unroll
    for (u32 q = 0; q < 4; q++)
      if (ok[q]) atomicAdd(&h[(u32)(src.key(key[q]) >> (64 - d1))], 1u);
  }
  for (; r < n_reads; r += stride) {
    u64 key;
    if (src.load(r, key)) atomicAdd(&h[(u32)(src.key(key) >> (64 - d1))], 1u);
  }
  __syncthreads();
  for (u32 b = threadIdx.x; b < nb; b += blockDim.x)
    if (h[b]) atomicAdd(&hist1[b], h[b]);
}

// ---- one block: coarse bin bases, tile table of level 2, and the bucket boundaries pbeg[] that no
// level-2 tile will write: all of them when there is no second level (d2 = 0), else those of the
// EMPTY coarse bins (no tile) and the final pbeg[2^(d1+d2)] = number of records ----
static __global__ void __launch_bounds__(1024)
k_pt_scan1(const u32 *__restrict__ hist1, u32 d1, u32 d2, u32 *__restrict__ cbase, u32 *__restrict__ tprefix,
           u32 *__restrict__ pbeg, u32 *__restrict__ ucount_tail, u32 cap1, u32 *clear = nullptr) {
  HUMID_GUARD_LAST_VGPR();
  __shared__ u32 cnt[PT_MAXBINS1], off[PT_MAXBINS1 + 1], wsum[16];
  const u32 nb = 1u << d1;
  // cap1 > 0: hist1 = the cursors of a padded level 1 (what each coarse bin received; a bin that
  // overflowed its room is cut to it -- the run is discarded by the caller, nothing may leave its room)
  auto count_of = [&](u32 b) { const u32 v = hist1[b]; return (cap1 && v > cap1) ? cap1 : v; };
  if (threadIdx.x < nb) cnt[threadIdx.x] = count_of(threadIdx.x);
  __syncthreads();
  block_exscan_1024(cnt, off, nb, wsum);
  if (threadIdx.x < nb) cbase[threadIdx.x] = off[threadIdx.x];
  if (threadIdx.x == 0) { cbase[nb] = off[nb]; pbeg[nb << d2] = off[nb]; *ucount_tail = 0; }
  if (d2 == 0) {
    if (threadIdx.x < nb) pbeg[threadIdx.x] = off[threadIdx.x];
  } else {
    if (threadIdx.x < nb && cnt[threadIdx.x] == 0)               // empty coarse bins are rare
      for (u32 f = 0; f < (1u << d2); f++) pbeg[(threadIdx.x << d2) | f] = off[threadIdx.x];
  }
  __syncthreads();
  if (threadIdx.x < nb) cnt[threadIdx.x] = (count_of(threadIdx.x) + PT_TILE - 1) / PT_TILE;
  __syncthreads();
  block_exscan_1024(cnt, off, nb, wsum);
  if (threadIdx.x < nb) tprefix[threadIdx.x] = off[threadIdx.x];
  if (threadIdx.x == 0) tprefix[nb] = off[nb];
  // clear (= hist1, the cursors of a padded level 1 that nobody reads after this): left at zero for the next use
  if (clear && threadIdx.x < nb) clear[threadIdx.x] = 0;
}

// ---- the scatter of one level ----
// LEVEL 1: tile = 8192 consecutive reads of the input; bin = top d1 key bits; bin b's region starts at
//          base[b] (cbase).  LEVEL 2: tile = up to 8192 records of ONE coarse bin c; bin = the d2 key bits
//          below; base = the level-2 histogram: the region of (c, f) starts at cbase[c] + the fine
//          counts of c before f, which the first tile of c also writes to pbeg_out[c << d2 | f].
//          cursor[]: records already placed in each region (zeroed before the launch).
// THREADS: 1024 (tiles of PT_TILE records: the tile bookkeeping of level 2 assumes it), or 512 for a LEVEL 1 whose 120 KB
// of LDS per 1024-thread workgroup would leave one workgroup per CU (the graph stage's grouping: 333 tiles on 256 CUs)
template <int LEVEL, class SRC, u32 THREADS = 1024u>
__global__ void __launch_bounds__(THREADS)
k_pt_scatter(SRC src, u32 n_reads, const u64 *__restrict__ k_in, const u32 *__restrict__ v_in,
             const u32 *__restrict__ tprefix, const u32 *__restrict__ cbase, u32 d1, u32 d2,
             const u32 *__restrict__ base, u32 *cursor, u64 *__restrict__ k_out, u32 *__restrict__ v_out,
             u32 *__restrict__ pbeg_out, u32 cap1, ull *over) {
  HUMID_GUARD_LAST_VGPR();
  static_assert(LEVEL == 1 || THREADS == PT_THREADS, "level 2 works on tiles of PT_TILE records");
  constexpr u32 TILE = THREADS * PT_IPT;
  __shared__ u64 skey[TILE];
  __shared__ u32 sval[TILE];
  __shared__ unsigned short sbin[TILE];
  __shared__ u32 cnt[PT_MAXBINS], loff[PT_MAXBINS + 1], goff[PT_MAXBINS], room[PT_MAXBINS], fbase[PT_MAXBINS + 1], wsum[8];
  __shared__ u32 s_c, s_beg, s_cnt, s_first;
  const u32 nb = 1u << (LEVEL == 1 ? d1 : d2);
  u32 t_beg, t_cnt, coarse = 0;
  if (LEVEL == 1) {
    t_beg = blockIdx.x * TILE;
    t_cnt = (t_beg >= n_reads) ? 0u : ((n_reads - t_beg < TILE) ? n_reads - t_beg : TILE);
  } else {
    if (blockIdx.x >= tprefix[1u << d1]) return;             // beyond the last tile (uniform exit)
    if (threadIdx.x == 0) {
      u32 c, b, n;
      pt_tile_of(tprefix, cbase, 1u << d1, blockIdx.x, cap1, c, b, n);
      s_c = c; s_beg = b; s_cnt = n;
      s_first = (blockIdx.x == tprefix[c]) ? 1u : 0u;          // first tile of its coarse bin
    }
    __syncthreads();
    // where the fine buckets of this coarse bin start: the coarse base + a scan of the bin's fine
    // counts (base = the level-2 histogram); its first tile also publishes them as pbeg[]
    coarse = s_c;
    if (threadIdx.x < nb) cnt[threadIdx.x] = base[(coarse << d2) | threadIdx.x];
    __syncthreads();
    block_exscan_512(cnt, fbase, nb, wsum);
    if (s_first && threadIdx.x < nb) pbeg_out[(coarse << d2) | threadIdx.x] = cbase[coarse] + fbase[threadIdx.x];
  }
  for (u32 b = threadIdx.x; b < nb; b += THREADS) cnt[b] = 0;
  __syncthreads();
  if (LEVEL == 2) { t_beg = s_beg; t_cnt = s_cnt; }
  u64 key[PT_IPT];
  u32 val[PT_IPT], binrank[PT_IPT];                           // bin << 16 | rank inside (tile, bin); ~0: none
#pragma unroll
  for (u32 q = 0; q < PT_IPT; q++) {
    const u32 j = threadIdx.x + q * THREADS;
    binrank[q] = NONE32;
    if (j < t_cnt) {
      bool ok = true;
      if (LEVEL == 1) { ok = src.load(t_beg + j, key[q]); val[q] = t_beg + j; }
      else { key[q] = k_in[t_beg + j]; val[q] = v_in[t_beg + j]; }
      if (ok) {
        const u64 kk = src.key(key[q]);
        const u32 bin = (LEVEL == 1) ? (u32)(kk >> (64 - d1)) : (u32)(kk >> (64 - d1 - d2)) & (nb - 1);
        binrank[q] = bin << 16 | atomicAdd(&cnt[bin], 1u);   // rank < 8192 < 2^16
      }
    }
  }
  __syncthreads();
  block_exscan_512(cnt, loff, nb, wsum);
  if (threadIdx.x < nb) {
    const u32 c = cnt[threadIdx.x];
    const u32 g = (LEVEL == 1) ? threadIdx.x : ((coarse << d2) | threadIdx.x);
    // LEVEL 1 with cap1: the bin's fixed room; what does not fit is dropped and reported (*over):
    // the caller repeats the partition with a histogram pass in front
    const u32 b0 = (LEVEL == 1) ? (cap1 ? g * cap1 : base[g]) : cbase[coarse] + fbase[threadIdx.x];
    const u32 had_before = c ? atomicAdd(&cursor[g], c) : 0u;
    goff[threadIdx.x] = b0 + had_before;
    room[threadIdx.x] = (LEVEL == 1 && cap1) ? (had_before >= cap1 ? 0u : cap1 - had_before) : 0xffffffffu;
    if (LEVEL == 1 && cap1 && had_before + c > cap1) *over = 1;
  }
#pragma unroll
  for (u32 q = 0; q < PT_IPT; q++)
    if (binrank[q] != NONE32) {
      const u32 bin = binrank[q] >> 16, p = loff[bin] + (binrank[q] & 0xffffu);
      skey[p] = key[q];
      sval[p] = val[q];
      sbin[p] = (unsigned short)bin;
    }
  __syncthreads();
  const u32 total = loff[nb];
  for (u32 s = threadIdx.x; s < total; s += THREADS) {
    const u32 bin = sbin[s];
    const u32 within = s - loff[bin];
    if (within >= room[bin]) continue;
    const u32 d = goff[bin] + within;
    k_out[d] = skey[s];
    v_out[d] = sval[s];
  }
}

// ---- level 2 histogram: records per fine bucket, one tile of one coarse bin per block ----
template <class SRC>
__global__ void __launch_bounds__(1024)
k_pt_hist2(SRC src, const u64 *__restrict__ k_in, const u32 *__restrict__ tprefix, const u32 *__restrict__ cbase, u32 d1,
           u32 d2, u32 *__restrict__ hist_fine, u32 cap1) {
  HUMID_GUARD_LAST_VGPR();
  __shared__ u32 h[PT_MAXBINS];
  __shared__ u32 s_c, s_beg, s_cnt;
  if (blockIdx.x >= tprefix[1u << d1]) return;
  const u32 nb = 1u << d2;
  if (threadIdx.x == 0) {
    u32 c, b, n;
    pt_tile_of(tprefix, cbase, 1u << d1, blockIdx.x, cap1, c, b, n);
    s_c = c; s_beg = b; s_cnt = n;
  }
  for (u32 b = threadIdx.x; b < nb; b += PT_THREADS) h[b] = 0;
  __syncthreads();
  const u32 beg = s_beg, n = s_cnt;
  for (u32 j = threadIdx.x; j < n; j += PT_THREADS)
    atomicAdd(&h[(u32)(src.key(k_in[beg + j]) >> (64 - d1 - d2)) & (nb - 1)], 1u);
  __syncthreads();
  for (u32 b = threadIdx.x; b < nb; b += PT_THREADS)
    if (h[b]) atomicAdd(&hist_fine[(s_c << d2) | b], h[b]);
}

// ---- grouping by a full key: level 2 of the graph stage's bucket order ----
// The neighbour search needs the words of equal combination key NEXT TO each other -- in any order,
// and the buckets in any order: a grouping, not a sort.  Level 1 (k_pt_hist1 / k_pt_scan1 /
// k_pt_scatter<1> with FieldsSrc, kernels_graph.hip.h) brings the words into 2^d1 coarse bins by the top key bits; here one
// workgroup per coarse bin groups its words by the remaining d2 <= 15 key bits: count per fine value
// in LDS (2^d2 counters), scan, place through the running counters.  Two streaming passes over the
// bin, so a bin of any size is handled (slowly when one key prefix holds 10^6 words).  Replaces three
// library radix passes with their per-pass memsets, the copy of the input they start with and the
// gather of the words behind them: 0.18 -> 0.05 ms at 2.7 M words and 24 key bits.
#define GF_THREADS 1024u
#define GF_MAXBITS 15u
#define GF_SMALL 7872u           // a coarse bin of up to this many words: 79.4 KB of LDS, two workgroups per CU (2 x 80 KB is ALL of a CU's LDS: the second workgroup did not get in)
#define GF_MID 32768u            // up to this many: 128 KB, still coalesced stores
// BIG = false: the normal case, a few thousand words per coarse bin -- 2^15 16-bit counters and the inverse
// permutation (output position -> input position, 16 bits each) in 80 KB of LDS, so two workgroups share a
// CU; the stores of the last pass are coalesced and its scattered LOADS hit the lines the two passes before
// just read.  BIG = true (a second launch that only takes the bins the first one left): 2^15 32-bit
// counters, a bin of any size, the words placed with scattered stores.
// SIZE 0: bins of <= GF_SMALL words, SIZE 1: <= GF_MID (the same road with room for a longer permutation),
// SIZE 2: the rest.  Three launches, each takes its own bins and leaves the others at once.
template <class SRC, int SIZE>
__global__ void __launch_bounds__(GF_THREADS, SIZE == 0 ? 8 : 4)       // SIZE 0: <= 64 registers, two workgroups per CU
k_group_fine(SRC src, const u64 *__restrict__ k_in, const u32 *__restrict__ v_in, const u32 *__restrict__ cbase, u32 d1,
             u32 d2, u64 *__restrict__ k_out, u32 *__restrict__ v_out, u32 cap1) {
  HUMID_GUARD_LAST_VGPR();
  constexpr bool BIG = SIZE == 2;
  constexpr u32 INV_WORDS = SIZE == 0 ? GF_SMALL / 2u + 16u : (1u << (GF_MAXBITS - 1)) + 16u;       // 16-bit entries, two per word, + the wave sums
  __shared__ u32 gf_lds[BIG ? (1u << GF_MAXBITS) + 16 : (1u << (GF_MAXBITS - 1)) + INV_WORDS];
  const u32 nb = 1u << d2, c = blockIdx.x;
  PH_DECL;
  PH(0);
  const u32 beg = cbase[c], end = cbase[c + 1];
  if (beg >= end) return;
  const u32 n = end - beg;
  if ((n <= GF_SMALL ? 0 : (n <= GF_MID ? 1 : 2)) != SIZE) return;                    // another launch's bin
  // cap1 > 0: level 1 was the PADDED scatter -- coarse bin c sits in its fixed room, the grouped words go
  // to the dense positions cbase[] (the scan of the bins' counts)
  const u32 ibeg = cap1 ? c * cap1 : beg;
  k_in += ibeg; v_in += ibeg; k_out += beg; v_out += beg;
  u32 *wsum = BIG ? gf_lds + (1u << GF_MAXBITS) : gf_lds + (1u << (GF_MAXBITS - 1)) + INV_WORDS - 16;
  const u32 lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  auto fine = [&](u64 w) { return (u32)(src.key(w) >> (64 - d1 - d2)) & (nb - 1); };
  if (!BIG) {
    unsigned short *inv = (unsigned short *)(gf_lds + (1u << (GF_MAXBITS - 1)));      // GF_SMALL / GF_MID entries, then the 16 wave sums
    // SIZE 0: the bin's words (at most 8 per thread) are requested ONCE, all together, and stay in registers for both
    // counting passes -- a loop of load / count / load / count is a chain of memory round trips (this kernel is one
    // round of 512 workgroups: its time IS the latency of one workgroup, 49 us before, see profiles/r03e)
    constexpr u32 GF_RPT = 8;
    u64 kk[GF_RPT];
    u32 ff[GF_RPT];
    if constexpr (SIZE == 0) {
#pragma unroll
      for (u32 q = 0; q < GF_RPT; q++) {
        const u32 j = threadIdx.x + q * GF_THREADS;
        kk[q] = j < n ? k_in[j] : 0ull;
      }
    }
    for (u32 b = threadIdx.x; b < (1u << (GF_MAXBITS - 1)); b += GF_THREADS) gf_lds[b] = 0;   // 2^15 16-bit counters, two per word
    __syncthreads();
    PH(1);
    if constexpr (SIZE == 0) {
#pragma unroll
      for (u32 q = 0; q < GF_RPT; q++) {
        const u32 j = threadIdx.x + q * GF_THREADS;
        ff[q] = fine(kk[q]);
        if (j < n) atomicAdd(&gf_lds[ff[q] >> 1], 1u << (16 * (ff[q] & 1)));          // (no carry: n < 2^16)
      }
    } else {
      for (u32 j = threadIdx.x; j < n; j += GF_THREADS) {
        const u32 f = fine(k_in[j]);
        atomicAdd(&gf_lds[f >> 1], 1u << (16 * (f & 1)));                             // (no carry: n < 2^16)
      }
    }
    PH(2);
    __syncthreads();
    PH(3);
    // exclusive scan of the 16-bit counters in place.  A wave owns a stretch of words / 16 consecutive words and walks it
    // in rows of 64 (lane l takes word 64 i + l: no bank conflict; a thread owning 16 CONSECUTIVE words, as before,
    // hit two banks with 64 lanes -- 10.7 of the kernel's 20 us per workgroup): totals first, then the offsets.
    const u32 words = nb >= 2 ? nb / 2 : 1u;
    const u32 n_waves = GF_THREADS / 64u;
    const u32 per_wave = (words + n_waves - 1) / n_waves, wbeg = wv * per_wave, wend = wbeg + per_wave < words ? wbeg + per_wave : words;
    u32 tot = 0;
    for (u32 w = wbeg + lane; w < wend; w += 64) { const u32 x = gf_lds[w]; tot += (x & 0xffffu) + (x >> 16); }
#pragma unroll
    for (u32 dd = 32; dd; dd >>= 1) tot += __shfl_xor(tot, dd);
    if (lane == 0) wsum[wv] = tot;
    __syncthreads();
    u32 run = 0;
    for (u32 q = 0; q < wv; q++) run += wsum[q];
    for (u32 w0 = wbeg; w0 < wend; w0 += 64) {                                        // (wave-uniform bounds)
      const u32 w = w0 + lane;
      const u32 x = w < wend ? gf_lds[w] : 0u;
      const u32 mine = (x & 0xffffu) + (x >> 16);
      const u32 incl = wave_incl_scan(mine);
      const u32 lo = run + incl - mine, hi = lo + (x & 0xffffu);
      if (w < wend) gf_lds[w] = lo | (hi << 16);                                      // offsets < n <= 2^15
      run += (u32)__builtin_amdgcn_readlane((int)incl, 63);
    }
    __syncthreads();
    PH(4);
    if constexpr (SIZE == 0) {
      u32 oldq[GF_RPT];
#pragma unroll
      for (u32 q = 0; q < GF_RPT; q++) {
        const u32 j = threadIdx.x + q * GF_THREADS;
        oldq[q] = j < n ? atomicAdd(&gf_lds[ff[q] >> 1], 1u << (16 * (ff[q] & 1))) : 0u;
      }
#pragma unroll
      for (u32 q = 0; q < GF_RPT; q++) {
        const u32 j = threadIdx.x + q * GF_THREADS;
        if (j < n) inv[(oldq[q] >> (16 * (ff[q] & 1))) & 0xffffu] = (unsigned short)j;
      }
      PH(5);
      __syncthreads();
      PH(6);
      // output position q <- input position inv[q]: eight gathers in flight per thread (the lines were just read), coalesced stores
      u32 jj[GF_RPT], vo[GF_RPT];
      u64 ko[GF_RPT];
#pragma unroll
      for (u32 t = 0; t < GF_RPT; t++) {
        const u32 q = threadIdx.x + t * GF_THREADS;
        jj[t] = q < n ? inv[q] : 0u;
      }
#pragma unroll
      for (u32 t = 0; t < GF_RPT; t++) {
        const u32 q = threadIdx.x + t * GF_THREADS;
        if (q < n) { ko[t] = k_in[jj[t]]; vo[t] = v_in[jj[t]]; }
      }
#pragma unroll
      for (u32 t = 0; t < GF_RPT; t++) {
        const u32 q = threadIdx.x + t * GF_THREADS;
        if (q < n) { k_out[q] = ko[t]; v_out[q] = vo[t]; }
      }
      PH(7);
      PH_END(4, 7, (c & 7u) == 3u);    // 1 bounds + loads issued + clear | 2 count | 3 barrier | 4 scan | 5 place | 6 barrier | 7 gather + store
      return;
    }
    for (u32 j = threadIdx.x; j < n; j += GF_THREADS) {
      const u32 f = fine(k_in[j]);
      const u32 old = atomicAdd(&gf_lds[f >> 1], 1u << (16 * (f & 1)));
      inv[(old >> (16 * (f & 1))) & 0xffffu] = (unsigned short)j;
    }
    __syncthreads();
    for (u32 q = threadIdx.x; q < n; q += GF_THREADS) {
      const u32 j = inv[q];
      k_out[q] = k_in[j];
      v_out[q] = v_in[j];
    }
    return;
  }
  for (u32 b = threadIdx.x; b < nb; b += GF_THREADS) gf_lds[b] = 0;
  __syncthreads();
  for (u32 j = threadIdx.x; j < n; j += GF_THREADS) atomicAdd(&gf_lds[fine(k_in[j])], 1u);
  __syncthreads();
  const u32 per = nb >= GF_THREADS ? nb / GF_THREADS : 1u;
  const u32 b0 = threadIdx.x * per;
  u32 mine = 0;
  if (b0 < nb)
    for (u32 q = 0; q < per; q++) mine += gf_lds[b0 + q];
  u32 incl = mine;
  incl = wave_incl_scan(incl);
  if (lane == 63) wsum[wv] = incl;
  __syncthreads();
  u32 run = incl - mine;
  for (u32 q = 0; q < wv; q++) run += wsum[q];
  if (b0 < nb)
    for (u32 q = 0; q < per; q++) {
      const u32 x = gf_lds[b0 + q];
      gf_lds[b0 + q] = run;
      run += x;
    }
  __syncthreads();
  for (u32 j = threadIdx.x; j < n; j += GF_THREADS) {
    const u64 w = k_in[j];
    const u32 p = atomicAdd(&gf_lds[fine(w)], 1u);
    k_out[p] = w;
    v_out[p] = v_in[j];
  }
}

// --------------------------------------------------------------------------------
// the un-permute: results from partition order back to read order, in two coalesced passes
// --------------------------------------------------------------------------------
// Pass 1 (k_unperm_bins): position i of the partition order holds read vals[i] and the padded slot of
// its word; its packed result (cluster id | keep << 31) travels as the record (result << 32 | read)
// into bin read >> wshift.  A bin covers 2^wshift consecutive reads, every read occurs at most once,
// so bin b owns the fixed room [b << wshift, (b + 1) << wshift) of the record array: no histogram.
// Pass 2 (k_unperm_window): one workgroup per bin scatters the bin's records into an LDS window of
// 2^wshift results (excluded reads keep 0: cluster 0, never kept) and writes cluster_id / keep -- or
// the packed word for the multi-GPU return stream -- with coalesced stores.
#define UW_MAXBINS 2048u
#define UW_MAXSHIFT 15u

static __global__ void __launch_bounds__(1024)
k_unperm_bins(const u32 *__restrict__ vals, const u32 *__restrict__ pslot, const u64 *__restrict__ slot_out,
              const u32 *__restrict__ n_pos_dev, u32 n_pos_max, u32 n_reads, u32 wshift, u32 n_bins, u32 *ucur,
              u64 *__restrict__ rec) {
  HUMID_GUARD_LAST_VGPR();
  __shared__ u64 srec[PT_TILE];
  __shared__ unsigned short sbin[PT_TILE];
  __shared__ u32 cnt[UW_MAXBINS], loff[UW_MAXBINS + 1], goff[UW_MAXBINS], wsum[16];
  u32 n_pos = n_pos_dev ? *n_pos_dev : n_pos_max;
  if (n_pos > n_pos_max) n_pos = n_pos_max;
  const u32 t_beg = blockIdx.x * PT_TILE;
  if (t_beg >= n_pos) return;
  const u32 t_cnt = (n_pos - t_beg < PT_TILE) ? n_pos - t_beg : PT_TILE;
  for (u32 b = threadIdx.x; b < n_bins; b += PT_THREADS) cnt[b] = 0;
  __syncthreads();
  u64 r64[PT_IPT];
  u32 binrank[PT_IPT];
  u32 vq[PT_IPT], sq[PT_IPT];
#pragma unroll
  for (u32 q = 0; q < PT_IPT; q++) {                          // all loads of the thread in flight together
    const u32 j = threadIdx.x + q * PT_THREADS;
    if (j < t_cnt) { vq[q] = vals[t_beg + j]; sq[q] = pslot[t_beg + j]; }
  }
#pragma unroll
  for (u32 q = 0; q < PT_IPT; q++) {
    const u32 j = threadIdx.x + q * PT_THREADS;
    binrank[q] = NONE32;
    if (j < t_cnt) {
      const u32 r = vq[q] & 0x7fffffffu;
      if (r < n_reads && !(vq[q] & 0x80000000u) && sq[q] != NOSLOT) {
        const u64 o = slot_out[sq[q]];
        const u32 c = (u32)o | (((u32)(o >> 32) == r) ? 0x80000000u : 0u);
        r64[q] = ((u64)c << 32) | r;
        const u32 bin = r >> wshift;
        binrank[q] = atomicAdd(&cnt[bin], 1u);
        vq[q] = bin;
      }
    }
  }
  __syncthreads();
  // exclusive scan of up to 2048 counters: two per thread
  {
    const u32 a = 2 * threadIdx.x, b = a + 1;
    const u32 ca = a < n_bins ? cnt[a] : 0u, cb = b < n_bins ? cnt[b] : 0u;
    const u32 s = ca + cb;
    const u32 lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    u32 incl = s;
    incl = wave_incl_scan(incl);
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    u32 before = 0;
    for (u32 k = 0; k < wv; k++) before += wsum[k];
    const u32 ex = before + incl - s;
    if (a < n_bins) loff[a] = ex;
    if (b < n_bins) loff[b] = ex + ca;
    if (threadIdx.x == 1023) loff[n_bins] = before + incl;
  }
  __syncthreads();
  for (u32 b = threadIdx.x; b < n_bins; b += PT_THREADS) {
    const u32 c = cnt[b];
    goff[b] = (b << wshift) + (c ? atomicAdd(&ucur[b], c) : 0u);
  }
#pragma unroll
  for (u32 q = 0; q < PT_IPT; q++)
    if (binrank[q] != NONE32) {
      const u32 p = loff[vq[q]] + binrank[q];
      srec[p] = r64[q];
      sbin[p] = (unsigned short)vq[q];
    }
  __syncthreads();
  const u32 total = loff[n_bins];
  for (u32 s = threadIdx.x; s < total; s += PT_THREADS) {
    const u32 bin = sbin[s];
    rec[goff[bin] + (s - loff[bin])] = srec[s];
  }
}

// PACKED: one u32 per read (multi-GPU return stream); else cluster_id u32 + keep u8.  WSHIFT: 14
// (64 KiB window, two workgroups per CU) or 15 (read sets beyond 32 M reads)
#define UW_THREADS 1024u
template <bool PACKED, u32 WSHIFT>
__global__ void __launch_bounds__(UW_THREADS)
k_unperm_window(const u64 *__restrict__ rec, u32 *__restrict__ ucur, u32 n_reads,
                u32 *__restrict__ cluster_id, u8 *__restrict__ keep) {
  HUMID_GUARD_LAST_VGPR();
  __shared__ u32 win[1u << WSHIFT];
  constexpr u32 wshift = WSHIFT;
  const u32 W = 1u << wshift;
  const u32 bin = blockIdx.x;
  const u32 r0 = bin << wshift;
  if (r0 >= n_reads) return;
  u32 n = ucur[bin];                                          // (requested first: it is back when the window is clear)
  for (u32 j = threadIdx.x; j < W; j += UW_THREADS) win[j] = 0;
  if (n > W) n = W;                                           // never beyond the bin's room
  __syncthreads();
  if (threadIdx.x == 0) ucur[bin] = 0;                        // the next pass finds its cursors at zero (no memset in front of it)
  const u64 *rb = rec + r0;
  for (u32 j0 = 0; j0 < n; j0 += 8 * UW_THREADS) {            // eight records per thread in flight
    u64 x[8];
#pragma unroll
    for (u32 q = 0; q < 8; q++) {
      const u32 j = j0 + q * UW_THREADS + threadIdx.x;
      x[q] = j < n ? rb[j] : ~0ull;
    }
#pragma unroll
    for (u32 q = 0; q < 8; q++) {
      const u32 j = j0 + q * UW_THREADS + threadIdx.x;
      if (j < n) win[(u32)x[q] & (W - 1)] = (u32)(x[q] >> 32);
    }
  }
  __syncthreads();
  const u32 cntw = (n_reads - r0 < W) ? n_reads - r0 : W;
  if (PACKED) {
    for (u32 j = threadIdx.x; j < cntw; j += UW_THREADS) cluster_id[r0 + j] = win[j];
  } else {
    for (u32 j = threadIdx.x; j < cntw; j += UW_THREADS) cluster_id[r0 + j] = win[j] & 0x7fffffffu;
    // keep flags: four per thread, one 4-byte store (r0 and W are multiples of 4)
    u32 *k4 = (u32 *)(keep + r0);
    const u32 n4 = (((uintptr_t)keep & 3) == 0) ? cntw >> 2 : 0u;
    for (u32 j = threadIdx.x; j < n4; j += UW_THREADS)
      k4[j] = (win[4 * j] >> 31) | ((win[4 * j + 1] >> 31) << 8) | ((win[4 * j + 2] >> 31) << 16) |
              ((win[4 * j + 3] >> 31) << 24);
    for (u32 j = (n4 << 2) + threadIdx.x; j < cntw; j += UW_THREADS) keep[r0 + j] = (u8)(win[j] >> 31);
  }
}

#endif  // HUMID_KERNELS_PART_HIP_H
