// common.hip.h -- types, counters and device helpers shared by all kernels
// Part of libhumid_hip.so (see humid_hip.hip for the pipeline and the C ABI).  Device code for
// gfx950 only; included once, in this order, by humid_hip.hip.
#ifndef HUMID_COMMON_HIP_H
#define HUMID_COMMON_HIP_H

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <new>
#include <string>
#include <vector>

#include <hip/hip_runtime.h>

#include "../../include/humid_hip.h"

typedef uint64_t u64;
typedef int64_t i64;
typedef uint32_t u32;
typedef uint8_t u8;
typedef unsigned long long ull;

#define EMPTY_KEY 0xffffffffffffffffull
#define NOSLOT 0xffffffffu
#define NONE32 0xffffffffu

enum { CTR_UNIQUE = 0, CTR_USABLE, CTR_EDGES, CTR_NONSINGLE, CTR_MEMBERS, CTR_SPECIAL,
       CTR_CLUSTERS, CTR_OVERFULL, CTR_BIGMASK /* combos with a bucket beyond k_pairs' walk */,
       CTR_SMALLROOTS /* components of 3 .. SMALL_COMP leaves listed by k_comp_count */,
       CTR_EOVER /* low word: an append region of the pair list was full (kernels_cgraph.hip.h) */,
       CTR_GOVER /* a padded coarse bin of the graph stage's grouping was full (group_words_by_stretch) */, CTR_N = 16 };

// --------------------------------------------------------------------------------
// device helpers
// --------------------------------------------------------------------------------
// Every kernel of this library starts with HUMID_GUARD_LAST_VGPR().  Found in round 2 while chasing
// the edge loss recorded in round 1 (tools/uniform_load_probe.hip reproduces it
// in 40 lines of inline assembly): on this MI355X / ROCm 7.2 stack the LAST register of a wave's
// vector-register allocation is occasionally (about one wave in 10^5) overwritten from outside the
// wave with the lane number 0..63 while the wave waits for memory -- whatever the kernel keeps
// there is silently replaced.  The register file is unified: naming a0 as clobbered makes the
// kernel allocate one granule of accumulation registers it never uses BEHIND its last VGPR, so the
// register that can be hit holds nothing (1000 launches x 45 k waves: 0 hits with the guard, 34
// launches hit without, profiles/r02_edge_loss/).  No instruction is emitted.
#define HUMID_GUARD_LAST_VGPR() asm volatile("" ::: "a0")

// Inclusive prefix sum over the 64 lanes of a wave with DPP moves (row shifts inside the rows of 16 lanes, then the two
// row broadcasts): six vector instructions of a few cycles each, where six __shfl_up steps are six round trips through
// the LDS crossbar (ds_bpermute).  All lanes of the wave must be active.
template <int CTRL, int ROWS, class T>
__device__ __forceinline__ T wave_dpp_add(T x) {
  if constexpr (sizeof(T) == 4) {
    return x + (T)(u32)__builtin_amdgcn_update_dpp(0, (int)(u32)x, CTRL, ROWS, 0xf, false);
  } else {
    static_assert(sizeof(T) == 8, "32- or 64-bit integers");
    const u32 lo = (u32)__builtin_amdgcn_update_dpp(0, (int)(u32)(u64)x, CTRL, ROWS, 0xf, false);
    const u32 hi = (u32)__builtin_amdgcn_update_dpp(0, (int)(u32)((u64)x >> 32), CTRL, ROWS, 0xf, false);
    return x + (T)(((u64)hi << 32) | lo);
  }
}
template <class T>
__device__ __forceinline__ T wave_incl_scan(T x) {
  x = wave_dpp_add<0x111, 0xf>(x);       // row_shr:1
  x = wave_dpp_add<0x112, 0xf>(x);       // row_shr:2
  x = wave_dpp_add<0x114, 0xf>(x);       // row_shr:4
  x = wave_dpp_add<0x118, 0xf>(x);       // row_shr:8 -> inclusive inside every row of 16
  x = wave_dpp_add<0x142, 0xa>(x);       // row_bcast:15 -> rows 1 and 3 take the total of the row in front
  x = wave_dpp_add<0x143, 0xc>(x);       // row_bcast:31 -> rows 2 and 3 take the total of the first half
  return x;
}
// Experiment builds only (-DHUMID_PHASE_CLOCKS; tools/phase_clocks.sh): thread 0 of a sampled workgroup reads the
// 100 MHz wall clock at the phase boundaries PH(k) of a kernel and adds the differences to humid_phase[id][k];
// [id][0] counts the sampled workgroups.  humid_ctx_destroy prints the table.
#ifdef HUMID_PHASE_CLOCKS
#define PH_KERNELS 8
#define PH_MAX 12
__device__ unsigned long long humid_phase[PH_KERNELS][PH_MAX];
#define PH_DECL unsigned long long ph_t[PH_MAX]
#define PH(k) ph_t[k] = wall_clock64()
#define PH_END(id, last, sampled)                                                                       \
  if (threadIdx.x == 0 && (sampled)) {                                                                  \
    atomicAdd(&humid_phase[id][0], 1ull);                                                               \
    for (int ph_k = 1; ph_k <= (last); ph_k++) atomicAdd(&humid_phase[id][ph_k], ph_t[ph_k] - ph_t[ph_k - 1]); \
  }
#else
#define PH_DECL
#define PH(k)
#define PH_END(id, last, sampled)
#endif

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

// Wide words (33 <= n <= 64 nucleotides): hi = the first n-32 nucleotides (right-aligned), lo =
// the last 32.  (hi, lo) lexicographic order == word order.  The graph kernels are templates over
// the word type (u64 or W2) through these helpers.
struct __attribute__((aligned(16))) W2 {
  u64 hi, lo;
};
__host__ __device__ __forceinline__ u64 w_xor(u64 a, u64 b) { return a ^ b; }
__host__ __device__ __forceinline__ W2 w_xor(W2 a, W2 b) { return W2{a.hi ^ b.hi, a.lo ^ b.lo}; }
__host__ __device__ __forceinline__ bool w_hits(u64 x, u64 m) { return (x & m) != 0; }
__host__ __device__ __forceinline__ bool w_hits(W2 x, W2 m) { return ((x.hi & m.hi) | (x.lo & m.lo)) != 0; }
__device__ __forceinline__ u32 w_mismatch(u64 x) { return nt_mismatch(x); }
__device__ __forceinline__ u32 w_mismatch(W2 x) { return nt_mismatch(x.hi) + nt_mismatch(x.lo); }
// bits [shift, shift + width) of the word, width <= 64
__host__ __device__ __forceinline__ u64 w_field(u64 w, u32 shift, u32 width) {
  return (w >> shift) & ((width >= 64) ? ~0ull : ((1ull << width) - 1ull));
}
__host__ __device__ __forceinline__ u64 w_field(W2 w, u32 shift, u32 width) {
  const unsigned __int128 v = ((unsigned __int128)w.hi << 64) | w.lo;
  return (u64)(v >> shift) & ((width >= 64) ? ~0ull : ((1ull << width) - 1ull));
}
// nucleotide i (0 = first) of an n-nucleotide word
__host__ __device__ __forceinline__ u32 w_sym(u64 w, u32 n, u32 i) { return (u32)(w >> (2 * (n - 1 - i))) & 3u; }
__host__ __device__ __forceinline__ u32 w_sym(W2 w, u32 n, u32 i) {
  const u32 nh = n - 32;
  return i < nh ? (u32)(w.hi >> (2 * (nh - 1 - i))) & 3u : (u32)(w.lo >> (2 * (n - 1 - i))) & 3u;
}
__host__ __device__ __forceinline__ bool w_less(W2 a, W2 b) { return a.hi < b.hi || (a.hi == b.hi && a.lo < b.lo); }
__host__ __device__ __forceinline__ bool w_eq(W2 a, W2 b) { return a.hi == b.hi && a.lo == b.lo; }
template <class WT> __host__ __device__ __forceinline__ WT w_from(W2 m);
template <> __host__ __device__ __forceinline__ u64 w_from<u64>(W2 m) { return m.lo; }
template <> __host__ __device__ __forceinline__ W2 w_from<W2>(W2 m) { return m; }

__device__ __forceinline__ u32 ld_agent(const u32 *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// /root/reference/src/cluster.cc:31-33 atLeastDouble_
__device__ __forceinline__ bool at_least_double(u64 a, u64 b) { return a >= 2 * b; }


#endif  // HUMID_COMMON_HIP_H
