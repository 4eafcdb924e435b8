// prims.hip.h -- device-wide exclusive scan and LSD radix sort, hand-written for gfx950
// Part of libhumid_hip.so (see humid_hip.hip for the pipeline and the C ABI).
//
// Why not rocPRIM (rounds 1-2 used it): (1) every kernel of this library must keep its last vector
// register empty (HUMID_GUARD_LAST_VGPR, common.hip.h) and a library's kernels
// cannot be made to; (2) rocPRIM instantiates every kernel once per architecture it knows (13 stubs
// per kernel: 1 624 of the 1 797 kernel symbols of the round-2 code object), which the loader walks
// at start-up of every process (the `humid` command line pays for it).  The scans of this path are
// small (buckets, unique words, graph nodes) and the sorts are fall-backs, so three plain kernels per
// scan / per radix pass are enough: nothing here is on the roofline-critical road.
#ifndef HUMID_PRIMS_HIP_H
#define HUMID_PRIMS_HIP_H

#include "common.hip.h"

// ---- input functors (what a transform / counting iterator was) ---------------------------------
template <class T>
struct PtrIn {
  const T *p;
  __device__ __forceinline__ T operator()(u64 i) const { return p[i]; }
};
template <class T, class S>
struct CastIn {                       // S array read as T
  const S *p;
  __device__ __forceinline__ T operator()(u64 i) const { return (T)p[i]; }
};
struct IotaIn {
  __device__ __forceinline__ u32 operator()(u64 i) const { return (u32)i; }
};
template <class F, class G>
struct ComposeIn {                    // f(g(i))
  F f;
  G g;
  __device__ __forceinline__ auto operator()(u64 i) const { return f(g(i)); }
};

// --------------------------------------------------------------------------------
// exclusive scan (sum)
// --------------------------------------------------------------------------------
#define PS_THREADS 256u
#define PS_ITEMS 16u
#define PS_TILE (PS_THREADS * PS_ITEMS)
#define PS_SMALL_THREADS 1024u
#define PS_SMALL_MAX (1u << 16)       // up to here one workgroup scans everything in one launch

// exclusive prefix of x over the block (THREADS a multiple of 64, <= 1024); *total = block sum.
// lds: THREADS / 64 + 1 entries.
template <class T, u32 THREADS>
__device__ __forceinline__ T ps_block_exscan(T x, T *lds, T *total) {
  const u32 lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  T incl = x;
  incl = wave_incl_scan(incl);
  if (lane == 63) lds[wv] = incl;
  __syncthreads();
  T before = 0, tot = 0;
#pragma unroll
  for (u32 k = 0; k < THREADS / 64; k++) {
    const T s = lds[k];
    if (k < wv) before += s;
    tot += s;
  }
  *total = tot;
  __syncthreads();
  return before + incl - x;
}

// one workgroup, n <= PS_SMALL_MAX: chunks of 1024 x PS_SMALL_ITEMS items staged through LDS (coalesced
// loads and stores, thread t scans PS_SMALL_ITEMS consecutive items of the chunk), a running carry
#define PS_SMALL_ITEMS 8u
template <class T, class In>
__global__ void __launch_bounds__(PS_SMALL_THREADS)
// (out may be the array `in` reads: a chunk is staged before it is written)
k_ps_scan_small(In in, u32 n, T *out) {
  HUMID_GUARD_LAST_VGPR();
  // (one pad per PS_SMALL_ITEMS entries: thread t's items start at entry 9 t, so the lanes of a wave spread
  // over the banks when each reads ITS items -- unpadded, a stride of 8 entries put every fourth lane on one bank)
  __shared__ T stage[PS_SMALL_THREADS * (PS_SMALL_ITEMS + 1)];
  __shared__ T lds[PS_SMALL_THREADS / 64 + 1];
  auto at = [](u32 j) { return j + j / PS_SMALL_ITEMS; };
  T carry = 0;
  for (u32 c0 = 0; c0 < n; c0 += PS_SMALL_THREADS * PS_SMALL_ITEMS) {
#pragma unroll
    for (u32 k = 0; k < PS_SMALL_ITEMS; k++) {
      const u32 i = c0 + k * PS_SMALL_THREADS + threadIdx.x;
      stage[at(k * PS_SMALL_THREADS + threadIdx.x)] = i < n ? in(i) : (T)0;
    }
    __syncthreads();
    T v[PS_SMALL_ITEMS];
    T s = 0;
#pragma unroll
    for (u32 k = 0; k < PS_SMALL_ITEMS; k++) { v[k] = stage[at(threadIdx.x * PS_SMALL_ITEMS + k)]; s += v[k]; }
    T tot;
    T run = carry + ps_block_exscan<T, PS_SMALL_THREADS>(s, lds, &tot);
#pragma unroll
    for (u32 k = 0; k < PS_SMALL_ITEMS; k++) { stage[at(threadIdx.x * PS_SMALL_ITEMS + k)] = run; run += v[k]; }
    carry += tot;
    __syncthreads();
#pragma unroll
    for (u32 k = 0; k < PS_SMALL_ITEMS; k++) {
      const u32 i = c0 + k * PS_SMALL_THREADS + threadIdx.x;
      if (i < n) out[i] = stage[at(k * PS_SMALL_THREADS + threadIdx.x)];
    }
    __syncthreads();
  }
}

// tile sums
template <class T, class In>
__global__ void __launch_bounds__(PS_THREADS)
k_ps_reduce(In in, u64 n, T *__restrict__ tile_sum) {
  HUMID_GUARD_LAST_VGPR();
  __shared__ T lds[PS_THREADS / 64 + 1];
  const u64 base = (u64)blockIdx.x * PS_TILE;
  T s = 0;
#pragma unroll
  for (u32 k = 0; k < PS_ITEMS; k++) {                       // striped: coalesced
    const u64 i = base + k * PS_THREADS + threadIdx.x;
    if (i < n) s += in(i);
  }
#pragma unroll
  for (u32 d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d);
  if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    T t = 0;
    for (u32 k = 0; k < PS_THREADS / 64; k++) t += lds[k];
    tile_sum[blockIdx.x] = t;
  }
}

// exclusive scan of the tile sums in place, one workgroup walking chunks with a running carry
template <class T>
__global__ void __launch_bounds__(PS_SMALL_THREADS)
k_ps_top(T *tile_sum, u32 n_tiles) {
  HUMID_GUARD_LAST_VGPR();
  __shared__ T lds[PS_SMALL_THREADS / 64 + 1];
  T carry = 0;
  for (u32 c0 = 0; c0 < n_tiles; c0 += PS_SMALL_THREADS) {
    const u32 i = c0 + threadIdx.x;
    const T v = i < n_tiles ? tile_sum[i] : (T)0;
    T tot;
    const T ex = ps_block_exscan<T, PS_SMALL_THREADS>(v, lds, &tot);
    if (i < n_tiles) tile_sum[i] = carry + ex;
    carry += tot;
  }
}

// every tile again: thread t owns PS_ITEMS consecutive items (staged through LDS so that both the
// loads and the stores are coalesced)
template <class T, class In>
__global__ void __launch_bounds__(PS_THREADS)
// (in place allowed: a tile is staged before it is written)
k_ps_down(In in, u64 n, const T *__restrict__ tile_base, T *out) {
  HUMID_GUARD_LAST_VGPR();
  __shared__ T stage[PS_TILE + PS_THREADS];               // (padded as in k_ps_scan_small)
  __shared__ T lds[PS_THREADS / 64 + 1];
  auto at = [](u32 j) { return j + j / PS_ITEMS; };
  const u64 base = (u64)blockIdx.x * PS_TILE;
#pragma unroll
  for (u32 k = 0; k < PS_ITEMS; k++) {
    const u64 i = base + k * PS_THREADS + threadIdx.x;
    stage[at(k * PS_THREADS + threadIdx.x)] = i < n ? in(i) : (T)0;
  }
  __syncthreads();
  T v[PS_ITEMS];
  T s = 0;
#pragma unroll
  for (u32 k = 0; k < PS_ITEMS; k++) { v[k] = stage[at(threadIdx.x * PS_ITEMS + k)]; s += v[k]; }
  T tot;
  T run = tile_base[blockIdx.x] + ps_block_exscan<T, PS_THREADS>(s, lds, &tot);
#pragma unroll
  for (u32 k = 0; k < PS_ITEMS; k++) { stage[at(threadIdx.x * PS_ITEMS + k)] = run; run += v[k]; }
  __syncthreads();
#pragma unroll
  for (u32 k = 0; k < PS_ITEMS; k++) {
    const u64 i = base + k * PS_THREADS + threadIdx.x;
    if (i < n) out[i] = stage[at(k * PS_THREADS + threadIdx.x)];
  }
}

// one workgroup, n <= PS_TINY_MAX: thread t owns the items [t * per, (t + 1) * per) straight from memory
// (a few hundred KB at most: the rank blocks of a bitmap, bucket counts), ONE block scan; the items are read
// twice (they stay in the cache)
#define PS_TINY_MAX 16384u
template <class T, class In>
__global__ void __launch_bounds__(PS_SMALL_THREADS)
// (out may be the array `in` reads: a thread reads its items before it writes them)
k_ps_scan_tiny(In in, u32 n, T *out) {
  HUMID_GUARD_LAST_VGPR();
  __shared__ T lds[PS_SMALL_THREADS / 64 + 1];
  const u32 per = (n + PS_SMALL_THREADS - 1) / PS_SMALL_THREADS;
  const u32 lo = threadIdx.x * per, hi = lo + per < n ? lo + per : n;
  T s = 0;
  for (u32 i = lo; i < hi; i += 4) {                         // four independent loads in flight
    T v0 = in(i), v1 = i + 1 < hi ? in(i + 1) : (T)0, v2 = i + 2 < hi ? in(i + 2) : (T)0, v3 = i + 3 < hi ? in(i + 3) : (T)0;
    s += v0 + v1 + v2 + v3;
  }
  T tot;
  T run = ps_block_exscan<T, PS_SMALL_THREADS>(s, lds, &tot);
  for (u32 i = lo; i < hi; i++) {
    const T v = in(i);
    out[i] = run;
    run += v;
  }
}

// up to PS_CHAIN_WGS workgroups in ONE launch, n <= PS_CHAIN_WGS x 1024 x 8: a workgroup scans its tile of
// 1024 x ITEMS items, publishes the tile's sum (value, then a flag that holds this launch's EPOCH, so the
// chain is cleared only when the 32-bit epoch wraps: ps_exscan), and its first wave collects the sums of ALL tiles before it -- one lane per
// earlier tile, at most 63 -- instead of walking a chain.  The grid is at most 64 workgroups on 256 CUs,
// so every tile a workgroup waits for is running.  Launches that share a PsChain must be ordered (one stream).
#define PS_CHAIN_WGS 64u
#define PS_CHAIN_MAX (PS_CHAIN_WGS * PS_SMALL_THREADS * 8u)
#define PS_CHAIN_MAX32 (PS_CHAIN_WGS * PS_SMALL_THREADS * 16u)   // 4-byte items: 16 per thread still fit the LDS staging (70 KB)
struct PsChain {
  unsigned long long val[PS_CHAIN_WGS];
  u32 flag[PS_CHAIN_WGS];
};
template <class T, class In, u32 ITEMS>
__global__ void __launch_bounds__(PS_SMALL_THREADS)
// (out may be the array `in` reads: a workgroup reads its tile before it writes it, tiles are disjoint)
k_ps_scan_chain(In in, u32 n, T *out, PsChain *ch, u32 epoch) {
  HUMID_GUARD_LAST_VGPR();
  __shared__ T stage[ITEMS > 1 ? PS_SMALL_THREADS * (ITEMS + 1) : 1];
  __shared__ T lds[PS_SMALL_THREADS / 64 + 1];
  __shared__ T s_before;
  auto at = [](u32 j) { return j + j / ITEMS; };
  const u32 base = blockIdx.x * PS_SMALL_THREADS * ITEMS;
  T v[ITEMS];
  T s = 0;
  if (ITEMS == 1) {
    const u32 i = base + threadIdx.x;
    v[0] = i < n ? in(i) : (T)0;
    s = v[0];
  } else {
#pragma unroll
    for (u32 k = 0; k < ITEMS; k++) {
      const u32 i = base + k * PS_SMALL_THREADS + threadIdx.x;
      stage[at(k * PS_SMALL_THREADS + threadIdx.x)] = i < n ? in(i) : (T)0;
    }
    __syncthreads();
#pragma unroll
    for (u32 k = 0; k < ITEMS; k++) { v[k] = stage[at(threadIdx.x * ITEMS + k)]; s += v[k]; }
  }
  T tot;
  T run = ps_block_exscan<T, PS_SMALL_THREADS>(s, lds, &tot);
  if (threadIdx.x == 0) {
    __hip_atomic_store(&ch->val[blockIdx.x], (unsigned long long)tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&ch->flag[blockIdx.x], epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (threadIdx.x < 64) {
    T mine = 0;
    if (threadIdx.x < blockIdx.x) {
      while (__hip_atomic_load(&ch->flag[threadIdx.x], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != epoch) __builtin_amdgcn_s_sleep(1);
      mine = (T)__hip_atomic_load(&ch->val[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
    for (u32 d = 32; d; d >>= 1) mine += __shfl_xor(mine, d);
    if (threadIdx.x == 0) s_before = mine;
  }
  __syncthreads();
  run += s_before;
  if (ITEMS == 1) {
    const u32 i = base + threadIdx.x;
    if (i < n) out[i] = run;
  } else {
#pragma unroll
    for (u32 k = 0; k < ITEMS; k++) { stage[at(threadIdx.x * ITEMS + k)] = run; run += v[k]; }
    __syncthreads();
#pragma unroll
    for (u32 k = 0; k < ITEMS; k++) {
      const u32 i = base + k * PS_SMALL_THREADS + threadIdx.x;
      if (i < n) out[i] = stage[at(k * PS_SMALL_THREADS + threadIdx.x)];
    }
  }
}

// scratch (in T) the scan of n items needs
static inline size_t ps_scan_scratch_items(u64 n) { return n <= PS_SMALL_MAX ? 0 : (size_t)((n + PS_TILE - 1) / PS_TILE); }

// out[i] = sum of in(j), j < i, for i < n.  scratch: ps_scan_scratch_items(n) entries of T.
// chain + epoch (optional): a zero-initialised PsChain in device memory and a counter that is different for
// every launch on it (never 0); then up to PS_CHAIN_MAX items are scanned by k_ps_scan_chain in one launch.
template <class T, class In>
static inline hipError_t ps_exscan(In in, T *out, u64 n, T *scratch, hipStream_t st, PsChain *chain = nullptr, u32 *epoch = nullptr) {
  if (n == 0) return hipSuccess;
  if (chain && n > 2048 && n <= (sizeof(T) == 4 ? PS_CHAIN_MAX32 : PS_CHAIN_MAX)) {
    if (++*epoch == 0) {   // the 32-bit epoch wrapped: a flag some launch left 2^32 - 1 launches ago would pass for the next one's
      hipError_t e = hipMemsetAsync(chain, 0, sizeof(PsChain), st);
      if (e != hipSuccess) return e;
      ++*epoch;
    }
    const u32 per = (u32)((n + PS_CHAIN_WGS * PS_SMALL_THREADS - 1) / (PS_CHAIN_WGS * PS_SMALL_THREADS));   // items per thread at 64 workgroups
    if (per <= 1)
      hipLaunchKernelGGL((k_ps_scan_chain<T, In, 1>), dim3((u32)((n + 1023) / 1024)), dim3(PS_SMALL_THREADS), 0, st, in, (u32)n, out, chain, *epoch);
    else if (per <= 2)
      hipLaunchKernelGGL((k_ps_scan_chain<T, In, 2>), dim3((u32)((n + 2047) / 2048)), dim3(PS_SMALL_THREADS), 0, st, in, (u32)n, out, chain, *epoch);
    else if (per <= 4)
      hipLaunchKernelGGL((k_ps_scan_chain<T, In, 4>), dim3((u32)((n + 4095) / 4096)), dim3(PS_SMALL_THREADS), 0, st, in, (u32)n, out, chain, *epoch);
    else if (per <= 8)
      hipLaunchKernelGGL((k_ps_scan_chain<T, In, 8>), dim3((u32)((n + 8191) / 8192)), dim3(PS_SMALL_THREADS), 0, st, in, (u32)n, out, chain, *epoch);
    else
      hipLaunchKernelGGL((k_ps_scan_chain<T, In, (sizeof(T) == 4 ? 16 : 8)>), dim3((u32)((n + 16383) / 16384)), dim3(PS_SMALL_THREADS), 0, st, in, (u32)n, out, chain, *epoch);
    return hipGetLastError();
  }
  if (n <= PS_TINY_MAX) {
    hipLaunchKernelGGL((k_ps_scan_tiny<T, In>), dim3(1), dim3(PS_SMALL_THREADS), 0, st, in, (u32)n, out);
    return hipGetLastError();
  }
  if (n <= PS_SMALL_MAX) {
    hipLaunchKernelGGL((k_ps_scan_small<T, In>), dim3(1), dim3(PS_SMALL_THREADS), 0, st, in, (u32)n, out);
    return hipGetLastError();
  }
  const u32 n_tiles = (u32)((n + PS_TILE - 1) / PS_TILE);
  hipLaunchKernelGGL((k_ps_reduce<T, In>), dim3(n_tiles), dim3(PS_THREADS), 0, st, in, n, scratch);
  hipLaunchKernelGGL((k_ps_top<T>), dim3(1), dim3(PS_SMALL_THREADS), 0, st, scratch, n_tiles);
  hipLaunchKernelGGL((k_ps_down<T, In>), dim3(n_tiles), dim3(PS_THREADS), 0, st, in, n, (const T *)scratch, out);
  return hipGetLastError();
}

// --------------------------------------------------------------------------------
// LSD radix sort, 8 bits per pass, stable
// --------------------------------------------------------------------------------
// A pass = tile histograms (k_rs_hist), their exclusive scan in (digit, tile) order (ps_exscan), and the
// scatter (k_rs_scatter): a wave takes RS_ITEMS rounds of 64 consecutive keys; in a round every lane
// learns which lanes hold its digit (eight ballots), its rank among them and -- through the wave's own
// digit counters in LDS, which only that wave touches -- its rank in the wave's stretch of the tile.
// Stable by construction: (tile, wave, round, lane) is the input order.
#define RS_THREADS 512u
#define RS_WAVES (RS_THREADS / 64u)
#define RS_ITEMS 8u
#define RS_TILE (RS_THREADS * RS_ITEMS)

template <class K>
__device__ __forceinline__ u32 rs_digit(K key, u32 shift, u32 mask) { return (u32)((u64)key >> shift) & mask; }

template <class K, class KIn>
__global__ void __launch_bounds__(RS_THREADS)
k_rs_hist(KIn kin, u32 n, u32 shift, u32 mask, u32 n_tiles, u32 *__restrict__ thist) {
  HUMID_GUARD_LAST_VGPR();
  __shared__ u32 h[256];
  if (threadIdx.x < 256) h[threadIdx.x] = 0;
  __syncthreads();
  const u32 base = blockIdx.x * RS_TILE;
#pragma unroll
  for (u32 k = 0; k < RS_ITEMS; k++) {
    const u32 i = base + k * RS_THREADS + threadIdx.x;
    if (i < n) atomicAdd(&h[rs_digit<K>(kin(i), shift, mask)], 1u);
  }
  __syncthreads();
  if (threadIdx.x < 256) thist[(size_t)threadIdx.x * n_tiles + blockIdx.x] = h[threadIdx.x];
}

// The scatter of one pass.  Keys (and values) of a tile are first brought into DIGIT ORDER inside LDS -- position =
// keys of smaller digits in the tile + keys of the digit in the waves before + rank inside the wave, which is the
// input order, so the sort stays stable -- and leave from there: consecutive threads then write consecutive places
// of a digit's run (16 keys on average) instead of 64 lanes writing to 64 runs (round 3, first form: 1.6 TB/s on
// 100 M (u64, u32) pairs, the reason the fallbacks that sort were slower than with the library's Onesweep).
template <class K, class V, bool HAS_V, class KIn, class VIn>
__global__ void __launch_bounds__(RS_THREADS)
k_rs_scatter(KIn kin, VIn vin, u32 n, u32 shift, u32 mask, u32 n_tiles, const u32 *__restrict__ tbase,
             K *__restrict__ kout, V *__restrict__ vout) {
  HUMID_GUARD_LAST_VGPR();
  __shared__ K skey[RS_TILE];
  __shared__ V sval[HAS_V ? RS_TILE : 1];
  __shared__ u32 wc[RS_WAVES][256];              // per wave: keys of each digit seen so far; then: in the waves before
  __shared__ u32 gbase[256], loff[256], wsum[4];
  const u32 lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (u32 q = threadIdx.x; q < RS_WAVES * 256; q += RS_THREADS) (&wc[0][0])[q] = 0;
  if (threadIdx.x < 256) gbase[threadIdx.x] = tbase[(size_t)threadIdx.x * n_tiles + blockIdx.x];
  const u32 w0 = blockIdx.x * RS_TILE + wv * (64u * RS_ITEMS);
  K key[RS_ITEMS];
  V val[HAS_V ? RS_ITEMS : 1];
  u32 rank[RS_ITEMS];
#pragma unroll
  for (u32 k = 0; k < RS_ITEMS; k++) {
    const u32 i = w0 + k * 64 + lane;
    if (i < n) {
      key[k] = kin(i);
      if (HAS_V) val[k] = vin(i);
    }
  }
  __syncthreads();
  const u64 below = (1ull << lane) - 1ull;
#pragma unroll
  for (u32 k = 0; k < RS_ITEMS; k++) {
    const u32 i = w0 + k * 64 + lane;
    const bool valid = i < n;
    const u32 d = valid ? rs_digit<K>(key[k], shift, mask) : 0u;
    u64 same = __ballot(valid);
#pragma unroll
    for (u32 b = 0; b < 8; b++) {
      const u64 m = __ballot((d >> b) & 1u);
      same &= ((d >> b) & 1u) ? m : ~m;
    }
    u32 old = 0;
    const u32 leader = valid ? (u32)__ffsll((long long)same) - 1u : lane;
    if (valid && lane == leader) { old = wc[wv][d]; wc[wv][d] = old + (u32)__popcll(same); }
    old = __shfl(old, leader);
    rank[k] = old + (u32)__popcll(same & below);
  }
  __syncthreads();
  // digit d (thread d of the first four waves): wc[w][d] -> keys of digit d in the waves before w; loff[d] = keys of
  // smaller digits in the tile
  if (threadIdx.x < 256) {
    u32 run = 0;
#pragma unroll
    for (u32 w = 0; w < RS_WAVES; w++) { const u32 c = wc[w][threadIdx.x]; wc[w][threadIdx.x] = run; run += c; }
    const u32 incl = wave_incl_scan(run);          // (waves 0 .. 3 are whole)
    if (lane == 63) wsum[wv] = incl;
    loff[threadIdx.x] = incl - run;                // exclusive inside the wave; the waves in front are added below
  }
  __syncthreads();
  if (threadIdx.x < 256) {
    u32 before = 0;
    for (u32 w = 0; w < wv; w++) before += wsum[w];
    loff[threadIdx.x] += before;
  }
  __syncthreads();
#pragma unroll
  for (u32 k = 0; k < RS_ITEMS; k++) {
    const u32 i = w0 + k * 64 + lane;
    if (i < n) {
      const u32 d = rs_digit<K>(key[k], shift, mask);
      const u32 p = loff[d] + wc[wv][d] + rank[k];
      skey[p] = key[k];
      if (HAS_V) sval[p] = val[k];
    }
  }
  __syncthreads();
  const u32 t_beg = blockIdx.x * RS_TILE;
  const u32 t_cnt = t_beg >= n ? 0u : (n - t_beg < RS_TILE ? n - t_beg : RS_TILE);
  for (u32 j = threadIdx.x; j < t_cnt; j += RS_THREADS) {
    const K kk = skey[j];
    const u32 d = rs_digit<K>(kk, shift, mask);
    const u32 p = gbase[d] + (j - loff[d]);
    kout[p] = kk;
    if (HAS_V) vout[p] = sval[j];
  }
}

// temporary storage of a sort of n items (bytes): one more copy of keys (and values) for the
// ping-pong, the tile histograms and the scan's tile sums
template <class K, class V, bool HAS_V>
static inline size_t rs_temp_bytes(u64 n) {
  const size_t n_tiles = (size_t)((n + RS_TILE - 1) / RS_TILE);
  const size_t hist = 256 * n_tiles * 4, hscr = ps_scan_scratch_items(256 * n_tiles) * 4;
  auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
  return up((size_t)n * sizeof(K)) + (HAS_V ? up((size_t)n * sizeof(V)) : 0) + up(hist) + up(hscr) + 256;
}

// Sorts by the key bits [b0, b1) (b0 < b1 <= 8 sizeof(K)); the first pass reads through the functors,
// the result lands in kout / vout.  temp: rs_temp_bytes<K, V, HAS_V>(n) bytes.  n < 2^32.
template <class K, class V, bool HAS_V, class KIn, class VIn>
static inline hipError_t rs_sort(void *temp, KIn kin, K *kout, VIn vin, V *vout, u64 n, u32 b0, u32 b1, hipStream_t st) {
  if (n == 0) return hipSuccess;
  const u32 n_tiles = (u32)((n + RS_TILE - 1) / RS_TILE);
  auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
  char *p = (char *)temp;
  K *ktmp = (K *)p; p += up((size_t)n * sizeof(K));
  V *vtmp = (V *)p; if (HAS_V) p += up((size_t)n * sizeof(V));
  u32 *hist = (u32 *)p; p += up((size_t)256 * n_tiles * 4);
  u32 *hscr = (u32 *)p;
  const u32 passes = b1 > b0 ? (b1 - b0 + 7) / 8 : 1;
  // ping-pong so that the LAST pass writes kout: pass q (0-based) writes kout when (passes - 1 - q) is even
  for (u32 q = 0; q < passes; q++) {
    const u32 shift = b0 + 8 * q;
    const u32 bits = (b1 > shift) ? ((b1 - shift < 8) ? b1 - shift : 8) : 0;
    const u32 mask = (1u << bits) - 1u;
    const bool to_out = ((passes - 1 - q) & 1u) == 0;
    K *kd = to_out ? kout : ktmp;
    V *vd = to_out ? vout : vtmp;
    const K *ks = to_out ? ktmp : kout;        // source of every pass but the first: the other buffer
    const V *vs = to_out ? vtmp : vout;
    if (q == 0) {
      hipLaunchKernelGGL((k_rs_hist<K, KIn>), dim3(n_tiles), dim3(RS_THREADS), 0, st, kin, (u32)n, shift, mask, n_tiles, hist);
    } else {
      hipLaunchKernelGGL((k_rs_hist<K, PtrIn<K>>), dim3(n_tiles), dim3(RS_THREADS), 0, st, PtrIn<K>{ks}, (u32)n, shift, mask, n_tiles, hist);
    }
    hipError_t e = ps_exscan<u32>(PtrIn<u32>{hist}, hist, (u64)256 * n_tiles, hscr, st);
    if (e != hipSuccess) return e;
    if (q == 0) {
      hipLaunchKernelGGL((k_rs_scatter<K, V, HAS_V, KIn, VIn>), dim3(n_tiles), dim3(RS_THREADS), 0, st, kin, vin, (u32)n, shift, mask,
                         n_tiles, (const u32 *)hist, kd, vd);
    } else {
      hipLaunchKernelGGL((k_rs_scatter<K, V, HAS_V, PtrIn<K>, PtrIn<V>>), dim3(n_tiles), dim3(RS_THREADS), 0, st, PtrIn<K>{ks},
                         PtrIn<V>{vs}, (u32)n, shift, mask, n_tiles, (const u32 *)hist, kd, vd);
    }
  }
  return hipGetLastError();
}

// --------------------------------------------------------------------------------
// run-length encode of a sorted array (the histograms of humid_get_histogram)
// --------------------------------------------------------------------------------
// head flags -> scan -> (value, first position) per run; the length of a run is the distance to the
// next run's first position
static __global__ void k_rle_heads(const u64 *__restrict__ sorted, u32 n, u32 *__restrict__ head) {
  HUMID_GUARD_LAST_VGPR();
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i > n) return;
  head[i] = (i < n && (i == 0 || sorted[i] != sorted[i - 1])) ? 1u : 0u;
}
static __global__ void k_rle_runs(const u64 *__restrict__ sorted, const u32 *__restrict__ head, const u32 *__restrict__ hpos,
                           u32 n, u64 *__restrict__ uniq, u32 *__restrict__ start) {
  HUMID_GUARD_LAST_VGPR();
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i > n) return;
  if (i == n) { start[hpos[n]] = n; return; }
  if (head[i]) { uniq[hpos[i]] = sorted[i]; start[hpos[i]] = i; }
}
static __global__ void k_rle_counts(const u32 *__restrict__ start, const u32 *__restrict__ n_runs_dev, u32 *__restrict__ counts) {
  HUMID_GUARD_LAST_VGPR();
  const u32 j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < *n_runs_dev) counts[j] = start[j + 1] - start[j];
}

#endif  // HUMID_PRIMS_HIP_H
