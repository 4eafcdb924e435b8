// kernels_part8.hip.h -- the count stage's partition and LDS count on 8-BYTE records (round 3)
// Part of libhumid_hip.so (see humid_hip.hip for the pipeline and the C ABI).  Device code for
// gfx950 only; included once, in this order, by humid_hip.hip.
//
// kernels_part.hip.h moves a (64-bit key, 32-bit read index) pair per read through both partition levels:
// 12 bytes, in two arrays.  A word-ordered key of an n-nucleotide word has only 2n significant bits, and
// once a record sits in coarse bin c its top d1 key bits are known from WHERE it is -- so
//     record = (key bits below the top d1) << ibits | read index,        ibits = bits of the read count
// fits ONE 64-bit word whenever 2n - d1 + ibits <= 64 (the metric workload: 48 - 8 + 24 = 64).  What this
// buys: 8 instead of 12 bytes per read and level through HBM, and a tile of 8192 records in 64 KB of
// LDS instead of 112 KB, so that TWO workgroups of the scatter share a CU.  Both levels scatter into
// PADDED bins of fixed room (keys that were checked to spread evenly: no histogram pass in front of
// either level, the bins' counts are the cursors left behind); any bin that outgrows its room is
// reported and the caller takes the exact, histogram-based kernels of kernels_part.hip.h instead.
#ifndef HUMID_KERNELS_PART8_HIP_H
#define HUMID_KERNELS_PART8_HIP_H

#include "common.hip.h"
#include "kernels_count.hip.h"
#include "kernels_part.hip.h"

// room of a fine bucket: the mean is <= PART_TARGET = 350 reads, but reads come in FAMILIES of equal words,
// so the spread is that of a compound distribution -- sigma ~ sqrt(mean x E[k^2]/E[k]) for family sizes k
// (46 for the benchmark's families of 4 on average, 110 for families of 20): 1024 leaves 6 sigma even then.
#define P8_CAP2_LOG 10u
#define P8_CAP2 (1u << P8_CAP2_LOG)
#define P8_RPT (P8_CAP2 / 256u)              // records per thread of k_dedup_rec

// word -> key' : the word-ordered partition key with its insignificant low bits dropped.
// key = (w - lo) * scale stretches the value range [lo, hi] over 64 bits (PartKeyOp); z = floor(log2 scale)
// of its low bits carry no information (distinct words stay distinct after >> z), kbits = 64 - z remain.
// scale = 2^z (always on one GPU): key' = w - lo.
struct RecKey {
  u64 lo, scale;
  u32 z, kbits;
  u32 pow2;
  __host__ __device__ __forceinline__ u64 operator()(u64 w) const { return ((w - lo) * scale) >> z; }
  // key' -> word
  __device__ __forceinline__ u64 word(u64 k) const {
    if (pow2) return lo + k;
    const u64 x = k << z, q = x / scale;
    return lo + q + ((x - q * scale) ? 1u : 0u);               // the smallest v with v * scale >= k 2^z
  }
};

// the reads of the count stage as a source of (key', index)
struct Reads8 {
  const u64 *words;
  const u8 *filtered;            // null: none filtered
  u64 rlo, rhi;                  // value range this rank counts (multi-GPU); check_range = 0: everything
  u32 check_range;
  RecKey key;
  __device__ __forceinline__ bool load(u32 j, u64 &k) const {
    const u64 w = words[j];                            // (requested together with the flag, not behind it)
    if (filtered && filtered[j]) return false;
    if (check_range && (w < rlo || w > rhi)) return false;
    k = key(w);
    return true;
  }
};

// a tile's records, sorted by bin in LDS (srec, loff), written out bin by bin: a group of G lanes (a power of
// two, 8 .. 64, about the mean number of records per bin and tile) takes one bin at a time, so that every
// store instruction of a wave writes 64 / G contiguous runs.  dst(bin, k): where record k of the bin goes
// (~0: dropped -- the bin's room is used up).
template <class Dst>
__device__ __forceinline__ void p8_write_bins(const u64 *srec, const u32 *loff, u32 nb, u32 G, u64 *__restrict__ out, Dst dst) {
  const u32 lane = threadIdx.x & 63, wv = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
  const u32 per_wave = 64 / G, sub = lane / G, gl = lane % G;
  for (u32 b0 = wv * per_wave; b0 < nb; b0 += n_waves * per_wave) {
    const u32 bin = b0 + sub;
    if (bin >= nb) continue;
    const u32 beg = loff[bin], n = loff[bin + 1] - beg;
    for (u32 k = gl; k < n; k += G) {
      const u64 d = dst(bin, k);
      if (d != ~0ull) out[d] = srec[beg + k];
    }
  }
}
__device__ __forceinline__ u32 p8_group(u32 records, u32 nb) {
  u32 g = 8;
  while (g < 64 && g * nb < records) g <<= 1;
  return g;
}

// ---- level 1: reads -> padded coarse bins (top d1 key bits), records as described above ----
// THREADS x PT_IPT reads per tile (1024: the tile of the other levels; 512: twice the workgroups per CU, whose
// load / rank / write phases then overlap more)
template <class SRC, u32 THREADS>
__global__ void __launch_bounds__(THREADS)
k_p8_scatter1(SRC src, u32 n_reads, u32 kbits, u32 d1, u32 ibits, u32 cap1, u32 *cursor, u64 *__restrict__ out, ull *ctr) {
  HUMID_GUARD_LAST_VGPR();
  constexpr u32 TILE = THREADS * PT_IPT;
  __shared__ u64 srec[TILE];
  // up to 1024 coarse bins (d1 = 10: chosen when a record would not fit 64 bits with 9, i.e. 24-nt words beyond 33 M
  // reads); the room left in a bin is what separates its next free place from the end of its room: no array of its own
  __shared__ u32 cnt[PT_MAXBINS1], loff[PT_MAXBINS1 + 1], goff[PT_MAXBINS1], wsum[16];
  const u32 nb = 1u << d1;
  PH_DECL;
  PH(0);
  const u32 t_beg = blockIdx.x * TILE;
  const u32 t_cnt = (t_beg >= n_reads) ? 0u : ((n_reads - t_beg < TILE) ? n_reads - t_beg : TILE);
  for (u32 b = threadIdx.x; b < nb; b += THREADS) cnt[b] = 0;
  __syncthreads();
  PH(1);
  const u32 rbits = kbits - d1;                               // key bits a record keeps
  u64 rec[PT_IPT];
  u32 binrank[PT_IPT];                                        // bin << 16 | rank inside (tile, bin); ~0: none
#pragma unroll
  for (u32 q = 0; q < PT_IPT; q++) {
    const u32 j = threadIdx.x + q * THREADS;
    binrank[q] = NONE32;
    u64 k;
    if (j < t_cnt && src.load(t_beg + j, k)) {
      const u32 bin = (u32)(k >> rbits);
      rec[q] = ((k & ((1ull << rbits) - 1ull)) << ibits) | (u64)(t_beg + j);
      binrank[q] = bin << 16 | atomicAdd(&cnt[bin], 1u);
    }
  }
  PH(2);
  __syncthreads();
  PH(3);
  block_exscan_1024(cnt, loff, nb, wsum);
  PH(4);
  if (threadIdx.x < nb) {
    const u32 c = cnt[threadIdx.x];
    const u32 had = c ? atomicAdd(&cursor[threadIdx.x], c) : 0u;
    goff[threadIdx.x] = threadIdx.x * cap1 + (had < cap1 ? had : cap1);     // (a full bin: its end, room 0)
    if (had + c > cap1) ctr[CTR_SPECIAL] = 1;                 // the bin outgrew its room: the caller repartitions
  }
  PH(5);
#pragma unroll
  for (u32 q = 0; q < PT_IPT; q++)
    if (binrank[q] != NONE32) srec[loff[binrank[q] >> 16] + (binrank[q] & 0xffffu)] = rec[q];
  __syncthreads();
  PH(6);
  p8_write_bins(srec, loff, nb, p8_group(loff[nb], nb), out,
                [&](u32 bin, u32 k) -> u64 { return goff[bin] + k < (bin + 1) * cap1 ? (u64)goff[bin] + k : ~0ull; });
  PH(7);
  PH_END(1, 7, (blockIdx.x & 15u) == 3u);  // 1 clear | 2 loads + LDS ranks | 3 barrier | 4 scan | 5 global cursors | 6 sort in LDS + barrier | 7 write out
}

// ---- level 2: one tile of one coarse bin -> padded fine buckets (the next d2 key bits) ----
// bucket g = c << d2 | f owns the positions [g << P8_CAP2_LOG, (g + 1) << P8_CAP2_LOG); cursor2[g] = its reads
static __global__ void __launch_bounds__(1024, 8)
k_p8_scatter2(const u64 *__restrict__ in, const u32 *__restrict__ tprefix, const u32 *__restrict__ cbase, u32 kbits, u32 d1,
              u32 d2, u32 ibits, u32 cap1, u32 *cursor2, u64 *__restrict__ out, ull *ctr) {
  HUMID_GUARD_LAST_VGPR();
  __shared__ u64 srec[PT_TILE];
  __shared__ u32 cnt[PT_MAXBINS], loff[PT_MAXBINS + 1], goff[PT_MAXBINS], room[PT_MAXBINS], wsum[8];
  __shared__ u32 s_c, s_beg, s_cnt;
  PH_DECL;
  PH(0);
  if (blockIdx.x >= tprefix[1u << d1]) return;                // beyond the last tile (uniform exit)
  const u32 nb = 1u << d2;
  if (threadIdx.x == 0) {
    u32 c, b, n;
    pt_tile_of(tprefix, cbase, 1u << d1, blockIdx.x, cap1, c, b, n);
    s_c = c; s_beg = b; s_cnt = n;
  }
  for (u32 b = threadIdx.x; b < nb; b += PT_THREADS) cnt[b] = 0;
  __syncthreads();
  PH(1);
  const u32 coarse = s_c, t_beg = s_beg, t_cnt = s_cnt;
  const u32 fshift = ibits + kbits - d1 - d2;                 // where a record keeps the fine bits
  u64 rec[PT_IPT];
  u32 binrank[PT_IPT];
#pragma unroll
  for (u32 q = 0; q < PT_IPT; q++) {
    const u32 j = threadIdx.x + q * PT_THREADS;
    binrank[q] = NONE32;
    if (j < t_cnt) rec[q] = in[t_beg + j];
  }
#pragma unroll
  for (u32 q = 0; q < PT_IPT; q++) {
    const u32 j = threadIdx.x + q * PT_THREADS;
    if (j < t_cnt) {
      const u32 bin = (u32)(rec[q] >> fshift) & (nb - 1);
      binrank[q] = bin << 16 | atomicAdd(&cnt[bin], 1u);
    }
  }
  PH(2);
  __syncthreads();
  PH(3);
  block_exscan_512(cnt, loff, nb, wsum);
  PH(4);
  if (threadIdx.x < nb) {
    const u32 c = cnt[threadIdx.x];
    const u32 g = (coarse << d2) | threadIdx.x;
    const u32 had = c ? atomicAdd(&cursor2[g], c) : 0u;
    goff[threadIdx.x] = had;
    room[threadIdx.x] = had >= P8_CAP2 ? 0u : P8_CAP2 - had;
    if (had + c > P8_CAP2) ctr[CTR_SPECIAL] = 1;
  }
  PH(5);
#pragma unroll
  for (u32 q = 0; q < PT_IPT; q++)
    if (binrank[q] != NONE32) srec[loff[binrank[q] >> 16] + (binrank[q] & 0xffffu)] = rec[q];
  __syncthreads();
  PH(6);
  const u64 gbase = (u64)(coarse << d2) << P8_CAP2_LOG;
  p8_write_bins(srec, loff, nb, p8_group(loff[nb], nb), out, [&](u32 bin, u32 k) -> u64 {
    return k < room[bin] ? gbase + ((u64)bin << P8_CAP2_LOG) + goff[bin] + k : ~0ull;
  });
  PH(7);
  PH_END(2, 7, (blockIdx.x & 15u) == 3u);  // 1 tile lookup + clear | 2 loads + LDS ranks | 3 barrier | 4 scan | 5 global cursors | 6 sort in LDS | 7 write out
}

// ---- the LDS count of one bucket, on records (k_dedup_lds<true> of kernels_count.hip.h) ----
// Bucket g reads its n = cursor2[g] <= P8_CAP2 records at g << P8_CAP2_LOG; key' = the bucket's top d1 bits | the record's
// key bits.  Outputs as k_dedup_lds: pad_word / pad_cf at [g << P8_CAP2_LOG, + unique words) in word order,
// agg[g] = reads << 32 | unique words; and, IN PLACE of every record, (padded slot of its word << 32 | read index) -- what the
// un-permute wants of a position (k_unperm_bins8), so that no separate slot array is written.
// What bounds this kernel (profiles/r03e_dedup_rec.md): a workgroup lives ~7 us and the CU holds all the waves it can
// (8 workgroups), and those waves keep the vector unit half busy: both the NUMBER of instructions and the LENGTH of
// the chains of dependent LDS operations count, HBM does not (the same loads and stores alone: 40 us).  Clocks inside
// the kernel: fill + records + clear 1.35 us, insert 1.4, ranks 3.4 (comparing every unique word with every other),
// store drain 0.7.
//
// Ranks without comparing: the table is NOT circular -- a probe that passes entry LDS_SLOTS - 1 goes on into DR_SPILL
// more entries (a bucket whose keys run past those is reported as overfull, like a full table) -- and the home of a
// key is monotone in the key (the key bits right below the bucket bits), so the TABLE ORDER IS THE KEY ORDER except
// inside a run of neighbouring occupied entries, where probing may have swapped keys.  The thread that claims an
// entry sets its bit in an occupancy bitmap; rank of an entry = bits set in front of its run + smaller keys inside the
// run (runs are ~1.1 entries long at the usual load of 8 %).
#define DR_SPILL 64u
#define DR_SLOTS (LDS_SLOTS + DR_SPILL)
// the run of occupied entries around entry sl = 32 w + bit, [start, end), read off the occupancy bitmap (bw = its word
// w; one word of zeros lies behind the bitmap): no walk through the table, so the keys of the run can be requested
// together
__device__ __forceinline__ void dr_run_bounds(const u32 *lbits, u32 w, u32 bit, u32 bw, u32 &start, u32 &end) {
  const u32 zb = ~bw & ((1u << bit) - 1u);                 // empty entries of the word in front of sl
  if (zb) start = (w << 5) + (32u - (u32)__clz((int)zb));
  else {
    u32 ww = w;
    start = w << 5;
    while (ww > 0) {
      const u32 pz = ~lbits[ww - 1];
      if (pz) { start = ((ww - 1) << 5) + (32u - (u32)__clz((int)pz)); break; }
      ww--;
      start = ww << 5;
    }
  }
  const u32 za = bit < 31 ? (~bw & ~((2u << bit) - 1u)) : 0u;   // empty entries of the word behind sl
  if (za) end = (w << 5) + (u32)__ffs((int)za) - 1u;
  else {
    u32 ww = w + 1;
    while (true) {                                         // (ends at the zero word behind the bitmap at the latest)
      const u32 nz = ~lbits[ww];
      if (nz) { end = (ww << 5) + (u32)__ffs((int)nz) - 1u; break; }
      ww++;
    }
  }
}
#define DR_WORDS (DR_SLOTS / 32u)                  // 34 words of the occupancy bitmap
#define DR_EARLY 1u                                // of the P8_RPT records per thread: requested before the fill is known
static __global__ void __launch_bounds__(256)
k_dedup_rec(u64 *recs, const u32 *__restrict__ cursor2, u32 n_reads, u32 pb, u32 d1, u32 ibits, RecKey rk,
            u64 *__restrict__ pad_word, uint2 *__restrict__ pad_cf, u64 *__restrict__ agg, ull *ctr) {
  HUMID_GUARD_LAST_VGPR();
  PH_DECL;
  __shared__ u64 lkey[DR_SLOTS];
  __shared__ uint2 lcf[DR_SLOTS];                      // (count, first read); after the rank phase .y = the entry's rank
  __shared__ unsigned short lslot_of[P8_CAP2];         // claim order -> table entry
  __shared__ u32 lbits[DR_WORDS + 2], lpre[DR_WORDS + 2];   // occupancy bitmap (one word of zeros behind it), set bits in front of each word
  __shared__ u32 lcount;
  const u32 g = blockIdx.x;
  const size_t beg = (size_t)g << P8_CAP2_LOG;
  PH(0);
  // the first quarter of the bucket's room (256 records; a bucket holds 305 on average) is requested before its fill
  // is known (the room exists whatever it holds; what lies behind the fill is never looked at): the two round trips
  // overlap.  (512 records up front fetched 26 MB more per launch of the empty part of the rooms and were 2 us slower.)
  u64 rq[P8_RPT];
#pragma unroll
  for (u32 q = 0; q < DR_EARLY; q++) rq[q] = recs[beg + threadIdx.x + 256u * q];
  u32 n = cursor2[g];
  if (n > P8_CAP2) n = P8_CAP2;                        // (an overfull bucket was reported by the scatter: the run is discarded)
  if (n == 0) {
    if (threadIdx.x == 0) agg[g] = 0;
    return;
  }
#pragma unroll
  for (u32 q = DR_EARLY; q < P8_RPT; q++) {
    const u32 i = threadIdx.x + 256u * q;
    if (i < n) rq[q] = recs[beg + i];
  }
  for (u32 s = threadIdx.x; s < DR_SLOTS; s += 256) { lkey[s] = EMPTY_KEY; lcf[s] = make_uint2(0u, NONE32); }
  if (threadIdx.x < DR_WORDS + 2) lbits[threadIdx.x] = 0;
  if (threadIdx.x == 0) lcount = 0;
  __syncthreads();
  PH(1);
  const u32 rbits = rk.kbits - d1, d2 = pb - d1;
  const u64 top = (u64)(g >> d2) << rbits;             // the coarse bin: the key's top d1 bits
  const u64 imask = (1ull << ibits) - 1ull;
  // table home = the key bits just below the bucket bits (fewer than LDS_SLOT_BITS of them left: all of them)
  const int hs = (int)rk.kbits - (int)pb - (int)LDS_SLOT_BITS;
  auto home = [&](u64 k) -> u32 { return hs >= 0 ? (u32)(k >> hs) & (LDS_SLOTS - 1) : (u32)k & ((1u << (rk.kbits - pb)) - 1u); };
  bool bad = false;
  u32 sq[P8_RPT];
#pragma unroll
  for (u32 q = 0; q < P8_RPT; q++) {
    const u32 i = threadIdx.x + 256u * q;
    sq[q] = NONE32;
    if (i >= n) continue;
    const u64 k = top | (rq[q] >> ibits);
    const u32 v = (u32)(rq[q] & imask);
    if (v >= n_reads) { bad = true; continue; }        // a malformed index is never used
    u32 s = home(k);
    bool claimed = false;
    while (true) {
      u64 cur = lkey[s];
      if (cur == EMPTY_KEY) { cur = atomicCAS((ull *)&lkey[s], EMPTY_KEY, (ull)k); claimed = cur == EMPTY_KEY; }
      if (claimed || cur == k) break;
      if (++s >= DR_SLOTS) break;
    }
    if (s >= DR_SLOTS) { bad = true; continue; }
    sq[q] = s;
    atomicAdd(&lcf[s].x, 1u);
    atomicMin(&lcf[s].y, v);
    if (claimed) {                                     // the entry is this record's: its place in the list and its bit
      lslot_of[atomicAdd(&lcount, 1u)] = (unsigned short)s;
      atomicOr(&lbits[s >> 5], 1u << (s & 31));
    }
  }
  if (bad) ctr[CTR_OVERFULL] = 1;
  PH(2);
  __syncthreads();
  PH(3);
  const u32 n_uniq = lcount;                           // <= n <= P8_CAP2
  // set bits in front of every bitmap word: each wave that has entries to rank scans the 34 counts for itself
  // (identical values: the waves may overwrite each other)
  const u32 lane = threadIdx.x & 63;
  if ((threadIdx.x & ~63u) < n_uniq) {
    const u32 pc = lane < DR_WORDS ? (u32)__popc(lbits[lane]) : 0u;
    u32 incl = pc;
    incl = wave_incl_scan(incl);
    if (lane < DR_WORDS) lpre[lane] = incl - pc;
  }
  for (u32 li = threadIdx.x; li < n_uniq; li += 256) {
    const u32 sl = lslot_of[li];
    const u32 w = sl >> 5, bit = sl & 31;
    const u32 bw = lbits[w];
    u32 r = lpre[w] + (u32)__popc(bw & ((1u << bit) - 1u));     // occupied entries in front of sl
    const bool left = bit ? ((bw >> (bit - 1)) & 1u) != 0 : (w > 0 && (lbits[w - 1] >> 31) != 0);
    const bool right = bit < 31 ? ((bw >> (bit + 1)) & 1u) != 0 : (lbits[w + 1] & 1u) != 0;
    const u64 k = lkey[sl];
    if (left || right) {                               // a run of several entries: order inside it by comparing
      u32 start, end, smaller = 0;
      dr_run_bounds(lbits, w, bit, bw, start, end);
      if (end > DR_SLOTS) end = DR_SLOTS;
      for (u32 j = start; j < end; j++) smaller += lkey[j] < k ? 1u : 0u;
      r = r - (sl - start) + smaller;                  // (all of [start, sl) is occupied)
    }
    const uint2 cf = lcf[sl];
    pad_word[beg + r] = rk.word(k);
    pad_cf[beg + r] = cf;
    lcf[sl].y = r;                                     // entry -> rank
  }
  if (threadIdx.x == 0) agg[g] = ((u64)n << 32) | n_uniq;     // one scan of these gives both prefixes and both totals
  PH(4);
  __syncthreads();
  PH(5);
#pragma unroll
  for (u32 q = 0; q < P8_RPT; q++) {
    const u32 i = threadIdx.x + 256u * q;
    if (i >= n) continue;
    const u32 li = sq[q] != NONE32 ? lcf[sq[q]].y : NONE32;    // (the entry of the record is still in its register)
    recs[beg + i] = ((u64)(li < n ? (u32)beg + li : NOSLOT) << 32) | (u32)(rq[q] & imask);
  }
  PH(6);
  PH_END(0, 6, (g & 255u) == 77u);       // 1 fill + records + clear | 2 insert | 3 barrier | 4 ranks + out | 5 barrier | 6 final
}

// padded (fixed room per bucket) -> dense unique arrays in walk order, one wave per bucket
static __global__ void __launch_bounds__(256)
k_compact_padded8(const u64 *__restrict__ pad_word, const uint2 *__restrict__ pad_cf, const u64 *__restrict__ agg,
                  const u64 *__restrict__ abase, u32 n_parts, u64 *__restrict__ s_word, u32 *__restrict__ s_slot,
                  u32 *__restrict__ s_cnt, u32 *__restrict__ s_first) {
  HUMID_GUARD_LAST_VGPR();
  const u32 wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const u32 lane = threadIdx.x & 63;
  if (wave >= n_parts) return;
  const u32 beg = wave << P8_CAP2_LOG, uc = (u32)agg[wave], ub = (u32)abase[wave];
  for (u32 j = lane; j < uc; j += 64) {
    s_word[ub + j] = pad_word[beg + j];
    s_slot[ub + j] = beg + j;
    const uint2 cf = pad_cf[beg + j];
    s_cnt[ub + j] = cf.x;
    s_first[ub + j] = cf.y;
  }
}

// ---- un-permute, pass 1, from the in-place records of k_dedup_rec ----
// A workgroup takes B consecutive buckets (about a tile's worth of records: the positions in use are the
// first cursor2[g] of every bucket's room, the holes behind them are never read); record = (padded slot <<
// 32 | read).  Otherwise as k_unperm_bins (kernels_part.hip.h): the packed result travels as (result << 32
// | read) into bin read >> wshift, a tile's records leave bin by bin as contiguous runs.  NBMAX = 1024 (read
// sets up to 16 M reads): 77 KB of LDS, two workgroups per CU.
// UIPT records per thread and chunk (a chunk of 1024 x UIPT records is staged at a time): 8 with up to 1024 bins; 7 with
// up to 1536 bins (33 M .. 50 M reads), so that the staging area and the three bin tables still leave room for TWO
// workgroups per CU; 8 again (and one workgroup per CU) beyond
template <u32 NBMAX, u32 UIPT = 8u>
__global__ void __launch_bounds__(1024, NBMAX <= 1536 ? 8 : 4)
k_unperm_bins8(const u64 *__restrict__ recs, const u32 *__restrict__ cursor2, const u64 *__restrict__ slot_out, u32 n_parts,
               u32 B, u32 n_reads, u32 wshift, u32 n_bins, u32 *ucur, u64 *__restrict__ rec) {
  HUMID_GUARD_LAST_VGPR();
  constexpr u32 UTILE = PT_THREADS * UIPT;
  __shared__ u64 srec[UTILE];
  __shared__ u32 cnt[NBMAX], loff[NBMAX + 1], goff[NBMAX], wsum[16];
  __shared__ u32 bpre[65];                                    // records before bucket g0 + b in this workgroup's stretch
  PH_DECL;
  PH(0);
  const u32 g0 = blockIdx.x * B;
  if (g0 >= n_parts) return;
  const u32 nbk = (n_parts - g0 < B) ? n_parts - g0 : B;      // <= 64
  if (threadIdx.x < 64) {
    u32 c = threadIdx.x < nbk ? cursor2[g0 + threadIdx.x] : 0u;
    if (c > P8_CAP2) c = P8_CAP2;
    u32 incl = c;
    incl = wave_incl_scan(incl);
    bpre[threadIdx.x + 1] = incl;
    if (threadIdx.x == 0) bpre[0] = 0;
  }
  __syncthreads();
  const u32 T = bpre[nbk];
  const u64 inv_mean = T ? (((u64)nbk << 32) / T) : 0;        // buckets per record, 32.32 fixed point
  PH(1);
  for (u32 c0 = 0; c0 < T; c0 += UTILE) {
    for (u32 b = threadIdx.x; b < n_bins; b += PT_THREADS) cnt[b] = 0;
    __syncthreads();
    u64 in[UIPT], r64[UIPT];
    u32 binrank[UIPT];                                      // bin << 16 | rank inside (tile, bin); ~0: none
#pragma unroll
    for (u32 q = 0; q < UIPT; q++) {                        // all loads of the thread in flight together
      const u32 r = c0 + threadIdx.x + q * PT_THREADS;
      binrank[q] = NONE32;
      if (r < T) {
        // the bucket holding record r (bpre[lo] <= r < bpre[lo + 1]): the buckets of a stretch are about equally
        // full, so r x buckets / records is the right one or a neighbour -- a step or two instead of a bisection
        u32 lo = (u32)(((u64)r * inv_mean) >> 32);
        if (lo >= nbk) lo = nbk - 1;
        while (bpre[lo] > r) lo--;
        while (bpre[lo + 1] <= r) lo++;
        in[q] = recs[((size_t)(g0 + lo) << P8_CAP2_LOG) + (r - bpre[lo])];
        binrank[q] = 0;
      }
    }
    PH(2);
#pragma unroll
    for (u32 q = 0; q < UIPT; q++) {
      if (binrank[q] == NONE32) continue;
      const u32 r = (u32)in[q], sl = (u32)(in[q] >> 32);
      binrank[q] = NONE32;
      if (r < n_reads && sl != NOSLOT) {
        const u64 o = slot_out[sl];
        const u32 c = (u32)o | (((u32)(o >> 32) == r) ? 0x80000000u : 0u);
        r64[q] = ((u64)c << 32) | r;
        const u32 bin = r >> wshift;
        binrank[q] = bin << 16 | atomicAdd(&cnt[bin], 1u);    // (bin < 2048, rank < 8192)
      }
    }
    PH(3);
    __syncthreads();
    PH(4);
    {
      const u32 a = 2 * threadIdx.x, b = a + 1;               // exclusive scan of up to 2048 counters: two per thread
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
    PH(5);
    for (u32 b = threadIdx.x; b < n_bins; b += PT_THREADS) {
      const u32 c = cnt[b];
      goff[b] = (b << wshift) + (c ? atomicAdd(&ucur[b], c) : 0u);
    }
    PH(6);
#pragma unroll
    for (u32 q = 0; q < UIPT; q++)
      if (binrank[q] != NONE32) srec[loff[binrank[q] >> 16] + (binrank[q] & 0xffffu)] = r64[q];
    __syncthreads();
    PH(7);
    p8_write_bins(srec, loff, n_bins, p8_group(loff[n_bins], n_bins), rec,
                  [&](u32 bin, u32 k) -> u64 { return (u64)goff[bin] + k; });
    __syncthreads();
    PH(8);
  }
  // 1 fills of the buckets | 2 bucket lookup + record loads issued | 3 gather of the results + LDS ranks | 4 barrier | 5 scan |
  // 6 global cursors | 7 sort in LDS | 8 write out
  PH_END(3, 8, (blockIdx.x & 15u) == 3u);
}

// --------------------------------------------------------------------------------
// two-word words (33 <= n <= 64 nucleotides): the WORD travels through the partition
// --------------------------------------------------------------------------------
// Round 2 partitioned (key, read index) pairs of two-word words and let the count kernel GATHER the 16-byte
// words by read index: one scattered load per read that fetches 64 bytes for 16 (k_dedup_lds_wide: 1.4 GB of
// traffic per 10 M reads, 0.27 ms).  Here a record is the word itself (16 B) + its read index (4 B, a second
// array), bins come from the word's head (the key is recomputed from the word at both levels, it is not
// stored), and the count kernel reads its bucket as two contiguous streams.  Tiles of PW_TILE records: 64 KB of
// words + 16 KB of indices in LDS; a workgroup takes the PT_TILE positions of a tile slot in two halves, so the
// tile bookkeeping (k_pt_scan1, pt_tile_of) is that of the one-word kernels.
#define PW_TILE 4096u
#define PW_IPT (PW_TILE / PT_THREADS)
#include "kernels_wide.hip.h"

// the word-ordered key of a two-word word: its head's top 48 bits through RecKey (see WideReadsSrc)
__device__ __forceinline__ u64 pw_key(const W2 &x, u32 hbits, const RecKey &rk) {
  return rk((hbits >= 64 ? x.hi : ((x.hi << (64 - hbits)) | (x.lo >> hbits))) >> WIDE_KEY_DROP);
}

// bin by bin, words and indices (p8_write_bins for the two arrays of a wide record)
template <class Dst>
__device__ __forceinline__ void pw_write_bins(const W2 *sw, const u32 *si, const u32 *loff, u32 nb, u32 G, W2 *__restrict__ ow,
                                              u32 *__restrict__ oi, Dst dst) {
  const u32 lane = threadIdx.x & 63, wv = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
  const u32 per_wave = 64 / G, sub = lane / G, gl = lane % G;
  for (u32 b0 = wv * per_wave; b0 < nb; b0 += n_waves * per_wave) {
    const u32 bin = b0 + sub;
    if (bin >= nb) continue;
    const u32 beg = loff[bin], n = loff[bin + 1] - beg;
    for (u32 k = gl; k < n; k += G) {
      const u64 d = dst(bin, k);
      if (d != ~0ull) { ow[d] = sw[beg + k]; oi[d] = si[beg + k]; }
    }
  }
}

// LEVEL 1: tile slot = PT_TILE consecutive reads; bins = top d1 key bits; padded coarse bins of cap1 records.
// LEVEL 2: tile slot = up to PT_TILE records of one coarse bin; bins = the next d2 key bits; padded buckets of P8_CAP2.
template <int LEVEL>
__global__ void __launch_bounds__(1024)
k_pw_scatter(const W2 *__restrict__ w_in, const u8 *__restrict__ filtered, const u32 *__restrict__ i_in, u32 n_reads, u32 hbits,
             RecKey rk, const u32 *__restrict__ tprefix, const u32 *__restrict__ cbase, u32 d1, u32 d2, u32 cap1, u32 *cursor,
             W2 *__restrict__ w_out, u32 *__restrict__ i_out, ull *ctr) {
  HUMID_GUARD_LAST_VGPR();
  __shared__ W2 sw[PW_TILE];
  __shared__ u32 si[PW_TILE];
  __shared__ u32 cnt[PT_MAXBINS], loff[PT_MAXBINS + 1], goff[PT_MAXBINS], room[PT_MAXBINS], wsum[8];
  __shared__ u32 s_c, s_beg, s_cnt;
  const u32 nb = 1u << (LEVEL == 1 ? d1 : d2);
  u32 t_beg, t_cnt, coarse = 0;
  if (LEVEL == 1) {
    t_beg = blockIdx.x * PT_TILE;
    t_cnt = (t_beg >= n_reads) ? 0u : ((n_reads - t_beg < PT_TILE) ? n_reads - t_beg : PT_TILE);
  } else {
    if (blockIdx.x >= tprefix[1u << d1]) return;              // beyond the last tile (uniform exit)
    if (threadIdx.x == 0) {
      u32 c, b, n;
      pt_tile_of(tprefix, cbase, 1u << d1, blockIdx.x, cap1, c, b, n);
      s_c = c; s_beg = b; s_cnt = n;
    }
    __syncthreads();
    coarse = s_c; t_beg = s_beg; t_cnt = s_cnt;
  }
  const u32 kshift = LEVEL == 1 ? rk.kbits - d1 : rk.kbits - d1 - d2;
  for (u32 h0 = 0; h0 < t_cnt; h0 += PW_TILE) {               // the tile slot in halves of PW_TILE records
    const u32 h_cnt = t_cnt - h0 < PW_TILE ? t_cnt - h0 : PW_TILE;
    for (u32 b = threadIdx.x; b < nb; b += PT_THREADS) cnt[b] = 0;
    __syncthreads();
    W2 w[PW_IPT];
    u32 idx[PW_IPT], binrank[PW_IPT];
#pragma unroll
    for (u32 q = 0; q < PW_IPT; q++) {
      const u32 j = threadIdx.x + q * PT_THREADS;
      binrank[q] = NONE32;
      if (j < h_cnt) {
        const u32 p = t_beg + h0 + j;
        bool ok = true;
        if (LEVEL == 1) { ok = !(filtered && filtered[p]); idx[q] = p; }
        else idx[q] = i_in[p];
        if (ok) {
          W2 x = w_in[p];
          if (LEVEL == 1 && hbits < 64) x.hi &= (1ull << hbits) - 1ull;
          w[q] = x;
          const u32 bin = (u32)(pw_key(x, hbits, rk) >> kshift) & (nb - 1);
          binrank[q] = bin << 16 | atomicAdd(&cnt[bin], 1u);
        }
      }
    }
    __syncthreads();
    block_exscan_512(cnt, loff, nb, wsum);
    if (threadIdx.x < nb) {
      const u32 c = cnt[threadIdx.x];
      const u32 g = LEVEL == 1 ? threadIdx.x : ((coarse << d2) | threadIdx.x);
      const u32 cap = LEVEL == 1 ? cap1 : P8_CAP2;
      const u32 had = c ? atomicAdd(&cursor[g], c) : 0u;
      goff[threadIdx.x] = had;
      room[threadIdx.x] = had >= cap ? 0u : cap - had;
      if (had + c > cap) ctr[CTR_SPECIAL] = 1;                // the bin outgrew its room: the caller takes the other road
    }
#pragma unroll
    for (u32 q = 0; q < PW_IPT; q++)
      if (binrank[q] != NONE32) {
        const u32 p = loff[binrank[q] >> 16] + (binrank[q] & 0xffffu);
        sw[p] = w[q];
        si[p] = idx[q];
      }
    __syncthreads();
    const u64 stride = LEVEL == 1 ? (u64)cap1 : (u64)P8_CAP2;
    const u64 gbase = LEVEL == 1 ? 0ull : (u64)(coarse << d2) * stride;
    pw_write_bins(sw, si, loff, nb, p8_group(loff[nb], nb), w_out, i_out, [&](u32 bin, u32 k) -> u64 {
      return k < room[bin] ? gbase + (u64)bin * stride + goff[bin] + k : ~0ull;
    });
    __syncthreads();
  }
}

// ---- the LDS count of one bucket of two-word words, from its contiguous records (k_dedup_lds_wide) ----
// A 128-bit word cannot be claimed with one LDS compare-and-swap.  k_dedup_lds_wide therefore keeps the POSITION of
// the claiming read in the table and compares through it (two dependent LDS reads, 16 bytes, per probe).  Here the
// table holds a 64-bit FINGERPRINT of the word (mix64 of both halves), claimed and compared like a one-word key
// -- one 8-byte LDS read per probe -- and exactness is restored afterwards: every position checks that its word
// EQUALS the word of the read that registered its entry; two different words with one fingerprint (probability
// ~10^-15 per bucket) raise ctr[CTR_OVERFULL] and the caller counts this read set with the position-tag kernel.
// Two size classes (a launch takes its own buckets and leaves the others at once).  Outputs: pad_word / pad_cf
// at [g << P8_CAP2_LOG, + unique words) in word order, agg[g] = reads << 32 | unique words, and per position
// out8 = (padded slot of its word << 32 | read index): what k_unperm_bins8 reads.
__device__ __forceinline__ u64 w2_fingerprint(const W2 &w) {
  const u64 f = mix64(w.lo ^ mix64(w.hi + 0x9e3779b97f4a7c15ull));
  return f == EMPTY_KEY ? EMPTY_KEY - 1 : f;
}
// Ranks as in k_dedup_rec: the HOME of a word is the part of its partition key right below the bucket bits -- monotone in
// the word -- in a table that does not wrap (DR_SPILL entries behind it), so the table order is the word order except
// inside a run of neighbouring occupied entries; an occupancy bitmap gives the entries in front of a run, the words
// of a run are compared (through the claimers' staged words).  Round 3 first ranked by comparing all unique words with
// each other, three dependent LDS reads per compare.
template <u32 SB, u32 STAGE, u32 LEN_MIN, u32 LEN_MAX>
__global__ void __launch_bounds__(256)
k_dedup_wide_rec(const W2 *__restrict__ recw, const u32 *__restrict__ reci, const u32 *__restrict__ cursor2, u32 hbits, RecKey rk,
                 u32 n_reads, u32 pb, W2 *__restrict__ pad_word, uint2 *__restrict__ pad_cf, u64 *__restrict__ agg,
                 u64 *__restrict__ out8, ull *ctr) {
  HUMID_GUARD_LAST_VGPR();
  constexpr u32 SLOTS = 1u << SB, TS = SLOTS + DR_SPILL, WORDS = TS / 32u, Q = STAGE / 256u;
  static_assert(STAGE % 256u == 0 && (STAGE & (STAGE - 1)) == 0 && LEN_MAX <= STAGE && LEN_MAX <= SLOTS && WORDS <= 62, "size class");
  __shared__ W2 wk[STAGE];                             // the bucket's words by position
  __shared__ u64 lkey[TS];                             // fingerprint, EMPTY_KEY = free
  __shared__ uint2 lcf[TS];                            // (count, first read); after the rank phase .y = the entry's rank
  __shared__ unsigned short lclaim[TS];                // position of the read that registered the entry
  __shared__ unsigned short lslot_of[STAGE];           // claim order -> table entry
  __shared__ u32 lbits[WORDS + 2], lpre[WORDS + 2];    // occupancy bitmap (a word of zeros behind it), set bits in front of each word
  __shared__ u32 lcount;
  const u32 g = blockIdx.x;
  u32 len = cursor2[g];
  if (len > P8_CAP2) len = P8_CAP2;                    // (reported by the scatter: the run is discarded)
  if (len == 0) {
    if (LEN_MIN == 0 && threadIdx.x == 0) agg[g] = 0;
    return;
  }
  if (len <= LEN_MIN || len > LEN_MAX) return;         // the other class's bucket
  const size_t beg = (size_t)g << P8_CAP2_LOG;
  W2 wq[Q];
  u32 vq[Q], sq[Q];
#pragma unroll
  for (u32 q = 0; q < Q; q++) {
    const u32 p = threadIdx.x + 256u * q;
    vq[q] = NONE32;
    sq[q] = NONE32;
    if (p < len) { vq[q] = reci[beg + p]; wq[q] = recw[beg + p]; wk[p] = wq[q]; }
  }
  for (u32 s = threadIdx.x; s < TS; s += 256) { lkey[s] = EMPTY_KEY; lcf[s] = make_uint2(0u, NONE32); }
  if (threadIdx.x < WORDS + 2) lbits[threadIdx.x] = 0;
  if (threadIdx.x == 0) lcount = 0;
  __syncthreads();
  const int hs = (int)rk.kbits - (int)pb - (int)SB;
  auto home = [&](u64 k) -> u32 { return hs >= 0 ? (u32)(k >> hs) & (SLOTS - 1) : (u32)k & ((1u << (rk.kbits - pb)) - 1u); };
  bool bad = false;
#pragma unroll
  for (u32 q = 0; q < Q; q++) {
    const u32 p = threadIdx.x + 256u * q;
    if (p >= len || bad) continue;
    const u32 v = vq[q];
    if (v >= n_reads) { bad = true; continue; }        // a malformed index is never used
    const u64 f = w2_fingerprint(wq[q]);
    u32 s = home(pw_key(wq[q], hbits, rk));
    bool claimed = false;
    while (true) {
      u64 cur = lkey[s];
      if (cur == EMPTY_KEY) { cur = atomicCAS((ull *)&lkey[s], EMPTY_KEY, (ull)f); claimed = cur == EMPTY_KEY; }
      if (claimed || cur == f) break;
      if (++s >= TS) break;
    }
    if (s >= TS) { bad = true; continue; }
    sq[q] = s;
    atomicAdd(&lcf[s].x, 1u);
    atomicMin(&lcf[s].y, v);
    if (claimed) {
      lclaim[s] = (unsigned short)p;
      lslot_of[atomicAdd(&lcount, 1u)] = (unsigned short)s;
      atomicOr(&lbits[s >> 5], 1u << (s & 31));
    }
  }
  __syncthreads();
  // exactness: the word of every position against the word that registered its entry
#pragma unroll
  for (u32 q = 0; q < Q; q++)
    if (sq[q] != NONE32 && !w_eq(wk[lclaim[sq[q]]], wq[q])) bad = true;
  if (bad) ctr[CTR_OVERFULL] = 1;
  const u32 n_uniq = lcount < STAGE ? lcount : STAGE;
  const u32 lane = threadIdx.x & 63;
  if ((threadIdx.x & ~63u) < n_uniq) {                 // (identical values: the waves may overwrite each other)
    const u32 pc = lane < WORDS ? (u32)__popc(lbits[lane]) : 0u;
    const u32 incl = wave_incl_scan(pc);
    if (lane < WORDS) lpre[lane] = incl - pc;
  }
  for (u32 li = threadIdx.x; li < n_uniq; li += 256) {
    const u32 sl = lslot_of[li];
    const u32 w32 = sl >> 5, bit = sl & 31;
    const u32 bw = lbits[w32];
    u32 r = lpre[w32] + (u32)__popc(bw & ((1u << bit) - 1u));    // occupied entries in front of sl
    const bool left = bit ? ((bw >> (bit - 1)) & 1u) != 0 : (w32 > 0 && (lbits[w32 - 1] >> 31) != 0);
    const bool right = bit < 31 ? ((bw >> (bit + 1)) & 1u) != 0 : (lbits[w32 + 1] & 1u) != 0;
    const W2 w = wk[lclaim[sl]];
    if (left || right) {                               // a run of several entries: order inside it by comparing the words
      u32 start, end, smaller = 0;
      dr_run_bounds(lbits, w32, bit, bw, start, end);
      if (end > TS) end = TS;
      for (u32 j = start; j < end; j++) smaller += w_less(wk[lclaim[j]], w) ? 1u : 0u;
      r = r - (sl - start) + smaller;                  // (all of [start, sl) is occupied)
    }
    const uint2 cf = lcf[sl];
    if (r < STAGE) {
      pad_word[beg + r] = w;
      pad_cf[beg + r] = cf;
    }
    lcf[sl].y = r;                                     // entry -> rank
  }
  if (threadIdx.x == 0) agg[g] = ((u64)len << 32) | n_uniq;
  __syncthreads();
#pragma unroll
  for (u32 q = 0; q < Q; q++) {
    const u32 p = threadIdx.x + 256u * q;
    if (p >= len) continue;
    const u32 li = sq[q] != NONE32 ? lcf[sq[q]].y : NONE32;
    out8[beg + p] = ((u64)(li < len ? (u32)beg + li : NOSLOT) << 32) | vq[q];
  }
}

// padded -> dense unique arrays of two-word words, one wave per bucket
static __global__ void __launch_bounds__(256)
k_compact_padded8_wide(const W2 *__restrict__ pad_word, const uint2 *__restrict__ pad_cf, const u64 *__restrict__ agg,
                       const u64 *__restrict__ abase, u32 n_parts, W2 *__restrict__ s_word, u32 *__restrict__ s_slot,
                       u32 *__restrict__ s_cnt, u32 *__restrict__ s_first) {
  HUMID_GUARD_LAST_VGPR();
  const u32 wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const u32 lane = threadIdx.x & 63;
  if (wave >= n_parts) return;
  const u32 beg = wave << P8_CAP2_LOG, uc = (u32)agg[wave], ub = (u32)abase[wave];
  for (u32 j = lane; j < uc; j += 64) {
    s_word[ub + j] = pad_word[beg + j];
    s_slot[ub + j] = beg + j;
    const uint2 cf = pad_cf[beg + j];
    s_cnt[ub + j] = cf.x;
    s_first[ub + j] = cf.y;
  }
}

#endif  // HUMID_KERNELS_PART8_HIP_H
