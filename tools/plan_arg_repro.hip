// plan_arg_repro.hip -- stand-alone reproduction attempt of the edge loss recorded in round 1
// (kernels_graph.hip.h): at 10 M reads, 5-25 of 219 k neighbour pairs were
// lost, run-to-run different, when the pigeonhole plan reached k_pairs
//   (B) as a ~0.5-1 KB struct passed BY VALUE and indexed dynamically (kernarg segment), or
//   (C) as a struct freshly uploaded with hipMemcpyAsync to a persistent device buffer and read
//       through wave-uniform (scalar) loads,
// while (A) small by-value structs with static indexing never lost one.
// This program runs the same pair-count loop over ~2.7 M sorted words with the plan handed over in
// each of those ways (plus host-side race variants of C) and reports, per variant:
//   - launches whose pair total differs from the CPU count,
//   - waves that saw a mask / earlier-mask value different from the one the host passed.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/plan_arg_repro tools/plan_arg_repro.hip
// Run:   tools/plan_arg_repro [reps]      (also with HIP_FORCE_DEV_KERNARG=0 / =1)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

typedef uint64_t u64;
typedef uint32_t u32;
typedef uint8_t u8;
typedef unsigned long long ull;

#define CHECK(x)                                                                              \
  do {                                                                                        \
    hipError_t e_ = (x);                                                                      \
    if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } \
  } while (0)

#define MAX_COMBOS 20
#define MAX_FIELDS 8
struct BigPlan {                        // the round-1 ComboPlan layout (u64 masks): 508 bytes
  u32 ncombo;
  u32 key_bits;
  u64 mask[MAX_COMBOS];
  u8 nfield[MAX_COMBOS];
  u8 shift[MAX_COMBOS][MAX_FIELDS];
  u8 width[MAX_COMBOS][MAX_FIELDS];
  u64 pad[64];                          // + 512 bytes: "a 1 KB plan struct"
};
struct SmallMasks { u64 m[MAX_COMBOS]; };

__device__ __forceinline__ u32 nt_mismatch(u64 x) { return (u32)__popcll((x | (x >> 1)) & 0x5555555555555555ull); }

enum { V_SMALL = 0, V_BYVALUE = 1, V_POINTER = 2 };

// seen[wave] = the bucket mask this wave used; seen_em[wave] = OR of the earlier masks it used
template <int VAR>
__global__ void __launch_bounds__(256)
k_count(const u64 *__restrict__ W, u32 n, u64 mask_small, SmallMasks em_small, BigPlan byval,
        const BigPlan *__restrict__ ptr, u32 cb, u32 distance, ull *total, u64 *seen, u64 *seen_em) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  u64 mask;
  if (VAR == V_SMALL) mask = mask_small;
  else if (VAR == V_BYVALUE) mask = byval.mask[cb];          // dynamic index into the kernarg segment
  else mask = ptr->mask[cb];                                 // wave-uniform load from device memory
  u32 found = 0;
  u64 em_or = 0;
  if (i < n) {
    const u64 wi = W[i];
    for (u32 j = i + 1; j < n; j++) {
      const u64 x = wi ^ W[j];
      if (x & mask) break;
      if (nt_mismatch(x) > distance) continue;
      bool first = true;
      if (VAR == V_SMALL) {
#pragma unroll
        for (u32 q = 0; q < MAX_COMBOS; q++) {
          first = first && !(q < cb && (x & em_small.m[q]) == 0);
          if (q < cb) em_or |= em_small.m[q];
        }
      } else if (VAR == V_BYVALUE) {
        for (u32 q = 0; q < cb; q++) { const u64 m = byval.mask[q]; em_or |= m; if ((x & m) == 0) first = false; }
      } else {
        for (u32 q = 0; q < cb; q++) { const u64 m = ptr->mask[q]; em_or |= m; if ((x & m) == 0) first = false; }
      }
      if (first) found++;
    }
  }
#pragma unroll
  for (u32 d = 32; d >= 1; d >>= 1) { found += __shfl_xor(found, d); em_or |= __shfl_xor(em_or, d); }
  if ((threadIdx.x & 63) == 0) {
    const u32 wave = i >> 6;
    seen[wave] = mask;
    seen_em[wave] = em_or;
    if (found) atomicAdd(total, (ull)found);
  }
}

// filler between the k_count launches: what a sort / scan does to the caches (streams 2 x 32 MB)
__global__ void k_filler(const u64 *__restrict__ a, u64 *__restrict__ b, u32 n) {
  for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) b[i] = a[i] * 3 + 1;
}

// ---- the kernel that DID lose edges in situ (tools/exp_plan: variant k1): bucket keys from the
// plan's u8 field tables, indexed dynamically (form 1: by-value struct, form 2: uploaded copy)
// FIX 0: as written.  FIX 1: the shift goes through readfirstlane like the width does.  FIX 2: every
// lane also stores what it saw (dbg[i] = shift | width << 8 | nfield << 16 | 1 << 24).  FIX 3: the
// instruction stream of FIX 0, but the kernel is made to ALLOCATE 16 VGPRs instead of 8 (an empty asm
// statement that names v15 as clobbered).
template <int VAR, int FIX>
__global__ void k_keys(const u64 *__restrict__ W, u32 n, BigPlan byval, const BigPlan *__restrict__ ptr, u32 cb,
                       u32 *__restrict__ key, u32 *__restrict__ dbg) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (FIX == 3) asm volatile("" ::: "v15");
  if (i >= n) return;
  const u64 w = W[i];
  u64 k = 0;
  const u32 nf = VAR == V_BYVALUE ? byval.nfield[cb] : ptr->nfield[cb];
  u32 seen = nf << 16 | 1u << 24;
  for (u32 f = 0; f < nf; f++) {
    const u32 wd = VAR == V_BYVALUE ? byval.width[cb][f] : ptr->width[cb][f];
    u32 sh = VAR == V_BYVALUE ? byval.shift[cb][f] : ptr->shift[cb][f];
    if (FIX == 1) sh = __builtin_amdgcn_readfirstlane(sh);
    if (FIX == 2) seen |= sh | wd << 8;
    k = ((wd >= 64) ? 0ull : (k << wd)) | ((w >> sh) & ((wd >= 64) ? ~0ull : ((1ull << wd) - 1ull)));
  }
  key[i] = (u32)k;
  if (FIX == 2) dbg[i] = seen;
}

// bad[0] = wrong keys; bad[1 + 2t], bad[2 + 2t] = index and value of the t-th (t < 8)
__global__ void k_check_keys(const u64 *__restrict__ W, const u32 *__restrict__ key, u32 n, u32 shift, u32 width,
                             u32 *bad) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const u32 want = (u32)((W[i] >> shift) & ((1ull << width) - 1ull));
  if (key[i] != want) {
    const u32 t = atomicAdd(&bad[0], 1u);
    if (t < 8) { bad[1 + 2 * t] = i; bad[2 + 2 * t] = key[i]; }
  }
}

static u64 cpu_count(const std::vector<u64> &W, u64 mask, const u64 *em, u32 cb, u32 d) {
  u64 tot = 0;
  const size_t n = W.size();
  for (size_t i = 0; i < n; i++)
    for (size_t j = i + 1; j < n; j++) {
      const u64 x = W[i] ^ W[j];
      if (x & mask) break;
      if ((u32)__builtin_popcountll((x | (x >> 1)) & 0x5555555555555555ull) > d) continue;
      bool first = true;
      for (u32 q = 0; q < cb; q++) if ((x & em[q]) == 0) first = false;
      if (first) tot++;
    }
  return tot;
}

int main(int argc, char **argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 30;
  const u32 target = 2700000;
  std::mt19937_64 rng(1002);
  std::vector<u64> w;
  w.reserve(target + target / 8);
  for (u32 i = 0; i < target; i++) w.push_back(rng() & ((1ull << 48) - 1));
  for (u32 i = 0; i < target / 12; i++) {                    // satellites at Hamming distance 1
    u64 x = w[rng() % target];
    const u32 pos = rng() % 24;
    x ^= (u64)(1 + rng() % 3) << (2 * pos);
    w.push_back(x);
  }
  std::sort(w.begin(), w.end());
  w.erase(std::unique(w.begin(), w.end()), w.end());
  const u32 U = (u32)w.size();
  // plan of the metric configuration: n = 24, d = 1: two combinations of 12 nt
  BigPlan plan;
  memset(&plan, 0, sizeof plan);
  plan.ncombo = 2;
  plan.key_bits = 24;
  plan.mask[0] = 0xffffffull << 24;
  plan.mask[1] = 0xffffffull;
  for (u32 q = 2; q < MAX_COMBOS; q++) plan.mask[q] = 0x1111111111111111ull * q;   // never used: recognisable
  // combination 1 walks the words in (suffix, prefix) order
  std::vector<u64> w1(w);
  std::sort(w1.begin(), w1.end(), [](u64 a, u64 b) {
    const u64 ka = ((a & 0xffffff) << 24) | (a >> 24), kb = ((b & 0xffffff) << 24) | (b >> 24);
    return ka < kb;
  });
  const u64 exp0 = cpu_count(w, plan.mask[0], plan.mask, 0, 1);
  const u64 exp1 = cpu_count(w1, plan.mask[1], plan.mask, 1, 1);
  printf("U = %u, expected pairs: combo 0 %llu, combo 1 %llu, sizeof(BigPlan) = %zu\n", U, (ull)exp0, (ull)exp1,
         sizeof(BigPlan));
  fflush(stdout);

  hipStream_t st;
  CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  u64 *dW0, *dW1, *dF0, *dF1, *d_seen, *d_seen_em;
  ull *d_tot;
  BigPlan *d_plan;
  const u32 n_waves = (U + 63) / 64;
  CHECK(hipMalloc(&dW0, (size_t)U * 8));
  CHECK(hipMalloc(&dW1, (size_t)U * 8));
  CHECK(hipMalloc(&dF0, (size_t)4 << 20 << 3));
  CHECK(hipMalloc(&dF1, (size_t)4 << 20 << 3));
  CHECK(hipMalloc(&d_seen, (size_t)n_waves * 8 * 4));
  CHECK(hipMalloc(&d_seen_em, (size_t)n_waves * 8 * 4));
  CHECK(hipMalloc(&d_tot, 8 * 4));
  CHECK(hipMalloc(&d_plan, sizeof(BigPlan)));
  CHECK(hipMemcpy(dW0, w.data(), (size_t)U * 8, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dW1, w1.data(), (size_t)U * 8, hipMemcpyHostToDevice));
  CHECK(hipMemset(dF0, 1, (size_t)4 << 20 << 3));

  SmallMasks em;
  for (u32 q = 0; q < MAX_COMBOS; q++) em.m[q] = plan.mask[q];
  BigPlan *h_plan = new BigPlan(plan);                 // pageable, heap: "the member of the context"
  BigPlan other;                                       // a different plan (d = 2 style masks)
  memset(&other, 0x5a, sizeof other);
  other.ncombo = 6;

  struct Variant { const char *name; int var; int upload; };
  // upload: 0 none, 1 hipMemcpyAsync from the pageable struct before every launch group,
  //         2 the same and the host struct is overwritten right after the call returns (and restored
  //           before the next upload): would expose an asynchronous read of pageable memory
  //         3 upload ONCE per repetition of a plan that alternates with `other` between repetitions
  //           (a stale copy is then a WRONG copy)
  const Variant variants[] = {{"A small by-value, static index", V_SMALL, 0},
                              {"B 1 KB by-value, dynamic index", V_BYVALUE, 0},
                              {"C uploaded, scalar loads", V_POINTER, 1},
                              {"C2 uploaded, host copy scribbled after the call", V_POINTER, 2},
                              {"C3 uploaded, alternating with another plan", V_POINTER, 3}};
  std::vector<u64> h_seen((size_t)n_waves * 4), h_seen_em((size_t)n_waves * 4);
  int any_bad = 0;
  for (const Variant &v : variants) {
    u32 bad_launches = 0, bad_waves = 0;
    long long lost = 0;
    for (int rep = 0; rep < reps; rep++) {
      CHECK(hipMemsetAsync(d_tot, 0, 32, st));
      if (v.upload == 3) {                             // the other plan was there before
        CHECK(hipMemcpyAsync(d_plan, &other, sizeof(BigPlan), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_filler, dim3(2048), dim3(256), 0, st, dF0, dF1, 4u << 20);
      }
      if (v.upload) {
        *h_plan = plan;
        CHECK(hipMemcpyAsync(d_plan, h_plan, sizeof(BigPlan), hipMemcpyHostToDevice, st));
        if (v.upload == 2) memset(h_plan, 0xee, sizeof(BigPlan));
      }
      // phase A (count) and phase B (fill) of the two combinations: four launches, as stage_graph does
      for (u32 phase = 0; phase < 2; phase++)
        for (u32 cb = 0; cb < 2; cb++) {
          const u32 slot = phase * 2 + cb;
          const u64 *W = cb ? dW1 : dW0;
          if (cb) hipLaunchKernelGGL(k_filler, dim3(2048), dim3(256), 0, st, dF0, dF1, 4u << 20);   // "the sort"
          const dim3 grid((U + 255) / 256), blk(256);
          if (v.var == V_SMALL)
            hipLaunchKernelGGL(k_count<V_SMALL>, grid, blk, 0, st, W, U, plan.mask[cb], em, plan, d_plan, cb, 1u,
                               d_tot + slot, d_seen + (size_t)slot * n_waves, d_seen_em + (size_t)slot * n_waves);
          else if (v.var == V_BYVALUE)
            hipLaunchKernelGGL(k_count<V_BYVALUE>, grid, blk, 0, st, W, U, plan.mask[cb], em, plan, d_plan, cb, 1u,
                               d_tot + slot, d_seen + (size_t)slot * n_waves, d_seen_em + (size_t)slot * n_waves);
          else
            hipLaunchKernelGGL(k_count<V_POINTER>, grid, blk, 0, st, W, U, plan.mask[cb], em, plan, d_plan, cb, 1u,
                               d_tot + slot, d_seen + (size_t)slot * n_waves, d_seen_em + (size_t)slot * n_waves);
        }
      ull tot[4];
      CHECK(hipMemcpyAsync(tot, d_tot, 32, hipMemcpyDeviceToHost, st));
      CHECK(hipMemcpyAsync(h_seen.data(), d_seen, (size_t)n_waves * 32, hipMemcpyDeviceToHost, st));
      CHECK(hipMemcpyAsync(h_seen_em.data(), d_seen_em, (size_t)n_waves * 32, hipMemcpyDeviceToHost, st));
      CHECK(hipStreamSynchronize(st));
      for (u32 slot = 0; slot < 4; slot++) {
        const u32 cb = slot & 1;
        const u64 want = cb ? exp1 : exp0;
        if (tot[slot] != want) { bad_launches++; lost += (long long)want - (long long)tot[slot]; }
        for (u32 wv = 0; wv < n_waves; wv++) {
          const u64 m = h_seen[(size_t)slot * n_waves + wv], e = h_seen_em[(size_t)slot * n_waves + wv];
          // a wave that found no candidate pair never loaded an earlier mask (e == 0)
          if (m != plan.mask[cb] || (e != 0 && e != (cb ? plan.mask[0] : 0))) {
            if (bad_waves < 5)
              printf("   rep %d launch %u wave %u: mask %016llx (want %016llx) earlier %016llx\n", rep, slot, wv,
                     (ull)m, (ull)plan.mask[cb], (ull)e);
            bad_waves++;
          }
        }
      }
    }
    printf("%-52s launches %d: wrong totals %u (pairs lost %lld), waves with a wrong mask %u\n", v.name, reps * 4,
           bad_launches, lost, bad_waves);
    fflush(stdout);
    if (bad_launches || bad_waves) any_bad = 1;
  }
  // ---- bucket keys of combination 1 from the u8 field tables ----
  plan.nfield[0] = 1; plan.shift[0][0] = 24; plan.width[0][0] = 24;
  plan.nfield[1] = 1; plan.shift[1][0] = 0;  plan.width[1][0] = 24;
  u32 *d_key, *d_bad, *d_dbg;
  CHECK(hipMalloc(&d_key, (size_t)U * 4));
  CHECK(hipMalloc(&d_dbg, (size_t)U * 4));
  CHECK(hipMalloc(&d_bad, 32 * 4));
  for (int fix = 0; fix < 4; fix++)
  for (int var = V_BYVALUE; var <= V_POINTER; var++) {
    u32 bad_launches = 0;
    unsigned long long bad_keys = 0;
    for (int rep = 0; rep < reps * 4; rep++) {
      *h_plan = plan;
      CHECK(hipMemcpyAsync(d_plan, h_plan, sizeof(BigPlan), hipMemcpyHostToDevice, st));
      hipLaunchKernelGGL(k_filler, dim3(2048), dim3(256), 0, st, dF0, dF1, 4u << 20);
      CHECK(hipMemsetAsync(d_bad, 0, 32 * 4, st));
      const dim3 grid((U + 255) / 256), blk(256);
      if (fix == 2) CHECK(hipMemsetAsync(d_dbg, 0, (size_t)U * 4, st));
#define LAUNCH_KEYS(V, F) hipLaunchKernelGGL((k_keys<V, F>), grid, blk, 0, st, dW0, U, plan, d_plan, 1u, d_key, d_dbg)
      if (var == V_BYVALUE) { if (fix == 0) LAUNCH_KEYS(V_BYVALUE, 0); else if (fix == 1) LAUNCH_KEYS(V_BYVALUE, 1); else if (fix == 2) LAUNCH_KEYS(V_BYVALUE, 2); else LAUNCH_KEYS(V_BYVALUE, 3); }
      else { if (fix == 0) LAUNCH_KEYS(V_POINTER, 0); else if (fix == 1) LAUNCH_KEYS(V_POINTER, 1); else if (fix == 2) LAUNCH_KEYS(V_POINTER, 2); else LAUNCH_KEYS(V_POINTER, 3); }
      hipLaunchKernelGGL(k_check_keys, grid, blk, 0, st, dW0, d_key, U, 0u, 24u, d_bad);
      u32 hb[32];
      CHECK(hipMemcpyAsync(hb, d_bad, 32 * 4, hipMemcpyDeviceToHost, st));
      CHECK(hipStreamSynchronize(st));
      if (hb[0]) {
        bad_launches++;
        bad_keys += hb[0];
        if (bad_launches <= 4) {
          printf("   keys form %d launch %d: %u wrong keys; first:", var, rep, hb[0]);
          for (u32 t = 0; t < 8 && t < hb[0]; t++)
            printf(" [i=%u lane %u got %06x want %06x]", hb[1 + 2 * t], hb[1 + 2 * t] & 63, hb[2 + 2 * t],
                   (u32)(w[hb[1 + 2 * t]] & 0xffffff));
          printf("\n");
          if (fix == 2) {                        // what the 64 lanes of the first affected wave saw
            u32 wv[64];
            const u32 i0 = hb[1] & ~63u;
            CHECK(hipMemcpy(wv, d_dbg + i0, 64 * 4, hipMemcpyDeviceToHost));
            printf("     wave at i = %u, per lane (1 << 24 | nfield << 16 | width << 8 | shift):\n     ", i0);
            for (int k = 0; k < 64; k++) printf("%07x%s", wv[k], (k & 7) == 7 ? "\n     " : " ");
            printf("\n");
          }
        }
      }
    }
    printf("%-52s %-28s launches %d: with wrong keys %u (wrong keys in total %llu)\n",
           var == V_BYVALUE ? "K1 bucket keys, u8 tables by value, dynamic index" : "K2 bucket keys, u8 tables uploaded",
           fix == 0 ? "" : fix == 1 ? "shift via readfirstlane" : fix == 2 ? "lanes store what they saw" : "same code, 16 VGPRs allocated",
           reps * 4, bad_launches, bad_keys);
    fflush(stdout);
    if (bad_launches) any_bad = 1;
  }
  printf(any_bad ? "REPRODUCED\n" : "not reproduced: every variant exact\n");
  return 0;
}
