// uniform_load_probe.hip -- what exactly goes wrong in the instruction sequence that lost edges
// (tools/plan_arg_repro.hip K1/K2 reproduce it from C++).
//
// hipcc turns a dynamically indexed u8 table in a by-value struct (or behind a uniform pointer) into
// VECTOR byte loads of a wave-uniform address:
//     global_load_ubyte v6, v1, s[0:1] offset:160      ; width  (v1 = 0 in every lane)
//     global_load_ubyte v7, v1, s[0:1]                  ; shift
//     s_waitcnt vmcnt(1) ; v_readfirstlane_b32 s6, v6 ; ... ; s_waitcnt vmcnt(0) ; use v7 per lane
// In the affected waves lane 0 was right and lanes 1..63 shifted by their lane number.  This probe
// issues that sequence from inline assembly with the destination registers pre-set to sentinels and
// stores what every lane sees, in several variants:
//   0  the sequence as generated (two ubyte loads, saddr + zero VGPR offset), table in device memory
//   1  the same, one load only
//   2  two DWORD loads of the same uniform addresses
//   3  two ubyte loads, address in a VGPR pair (no saddr)
//   4  variant 0 without the preceding global_load_dwordx2 in flight
//   5  the whole loop body as generated: the VALU consumer (v_lshrrev_b64 with the loaded byte as
//      shift amount, destination overlapping both load destinations) DIRECTLY after s_waitcnt vmcnt(0)
//   6  as 5 with one s_nop between the wait and the consumer
//   7  as 5 with dword loads
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/uniform_load_probe tools/uniform_load_probe.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef uint64_t u64;
typedef uint32_t u32;
typedef uint8_t u8;

#define CHECK(x)                                                                              \
  do {                                                                                        \
    hipError_t e_ = (x);                                                                      \
    if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } \
  } while (0)

struct Table {                 // 1 KB like the plan struct; the two bytes read sit 160 bytes apart
  u8 pad0[188];
  u8 shift[160];               // shift[8] is read: value 0x11
  u8 width[160];               // width[8] is read: value 0x18
  u8 pad1[516];
};
#define SENT_A 0xAAAAu
#define SENT_B 0xBBBBu

template <int MODE>
__global__ void __launch_bounds__(256)
k_probe(const u64 *__restrict__ W, u32 n, Table byval, const Table *__restrict__ ptr, u32 idx, u32 *__restrict__ out,
        u64 *__restrict__ sink) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const u8 *base = &ptr->shift[idx];                                          // wave-uniform
  (void)byval;
  u32 a = SENT_A, b = SENT_B, z = 0, sa = 0;
  u64 w = 0;
  const u64 *wa = W + i;
  if (MODE == 0) {
    asm volatile(
        "global_load_dwordx2 %[w], %[wa], off\n\t"
        "global_load_ubyte %[a], %[z], %[sb] offset:160\n\t"
        "global_load_ubyte %[b], %[z], %[sb]\n\t"
        "s_waitcnt vmcnt(1)\n\t"
        "v_readfirstlane_b32 %[sa], %[a]\n\t"
        "s_waitcnt vmcnt(0)\n\t"
        : [w] "=&v"(w), [a] "+&v"(a), [b] "+&v"(b), [sa] "=&s"(sa)
        : [wa] "v"(wa), [z] "v"(z), [sb] "s"(base)
        : "memory");
  } else if (MODE == 1) {
    asm volatile(
        "global_load_dwordx2 %[w], %[wa], off\n\t"
        "global_load_ubyte %[b], %[z], %[sb]\n\t"
        "s_waitcnt vmcnt(0)\n\t"
        : [w] "=&v"(w), [b] "+&v"(b)
        : [wa] "v"(wa), [z] "v"(z), [sb] "s"(base)
        : "memory");
    a = 0x18;
    sa = 0x18;
  } else if (MODE == 2) {
    const u8 *b4 = (const u8 *)((uintptr_t)base & ~(uintptr_t)3);
    asm volatile(
        "global_load_dwordx2 %[w], %[wa], off\n\t"
        "global_load_dword %[a], %[z], %[sb] offset:160\n\t"
        "global_load_dword %[b], %[z], %[sb]\n\t"
        "s_waitcnt vmcnt(1)\n\t"
        "v_readfirstlane_b32 %[sa], %[a]\n\t"
        "s_waitcnt vmcnt(0)\n\t"
        : [w] "=&v"(w), [a] "+&v"(a), [b] "+&v"(b), [sa] "=&s"(sa)
        : [wa] "v"(wa), [z] "v"(z), [sb] "s"(b4)
        : "memory");
    const u32 sh = 8 * (u32)((uintptr_t)base & 3);
    a = (a >> sh) & 0xff;
    b = (b >> sh) & 0xff;
    sa = (sa >> sh) & 0xff;
  } else if (MODE == 3) {
    const u8 *va = base + z;                     // the same address in a VGPR pair
    asm volatile(
        "global_load_dwordx2 %[w], %[wa], off\n\t"
        "global_load_ubyte %[a], %[va], off offset:160\n\t"
        "global_load_ubyte %[b], %[va], off\n\t"
        "s_waitcnt vmcnt(1)\n\t"
        "v_readfirstlane_b32 %[sa], %[a]\n\t"
        "s_waitcnt vmcnt(0)\n\t"
        : [w] "=&v"(w), [a] "+&v"(a), [b] "+&v"(b), [sa] "=&s"(sa)
        : [wa] "v"(wa), [va] "v"(va)
        : "memory");
  } else if (MODE >= 5) {
    // v6 / v7 as in the generated code; v7 is pre-set to 5: a consumer that reads it before the load
    // data is written shifts by 5 instead of 0x11
    u32 lo, hi;
    const u8 *b4 = (MODE == 7) ? (const u8 *)((uintptr_t)base & ~(uintptr_t)3) : base;
#define PROBE_BODY(LOAD, GAP)                                                     \
    asm volatile(                                                                 \
        "v_mov_b32 v7, 5\n\t"                                                     \
        "v_mov_b32 v6, 0xaa\n\t"                                                  \
        "global_load_dwordx2 %[w], %[wa], off\n\t"                                \
        LOAD " v6, %[z], %[sb] offset:160\n\t"                                    \
        LOAD " v7, %[z], %[sb]\n\t"                                               \
        "s_waitcnt vmcnt(1)\n\t"                                                  \
        "v_readfirstlane_b32 %[sa], v6\n\t"                                       \
        "s_waitcnt vmcnt(0)\n\t"                                                  \
        GAP                                                                       \
        "v_lshrrev_b64 v[6:7], v7, %[w]\n\t"                                      \
        "v_mov_b32 %[lo], v6\n\t"                                                 \
        "v_mov_b32 %[hi], v7\n\t"                                                 \
        : [w] "=&v"(w), [sa] "=&s"(sa), [lo] "=&v"(lo), [hi] "=&v"(hi)            \
        : [wa] "v"(wa), [z] "v"(z), [sb] "s"(b4)                                  \
        : "memory", "v6", "v7")
    if (MODE == 5) PROBE_BODY("global_load_ubyte", "");
    else if (MODE == 6) PROBE_BODY("global_load_ubyte", "s_nop 0\n\t");
    else PROBE_BODY("global_load_dword", "");
    // W is 0x0707...07: shifted by 0x11 the low word is 0x83838383, by 5 it is 0x38383838
    a = (lo == 0x83838383u && hi == 0x00000383u) ? 0x18 : (lo >> 16);
    b = (lo == 0x83838383u && hi == 0x00000383u) ? 0x11 : (lo & 0xff);
    sa = (MODE == 7) ? (sa >> (8 * (u32)((uintptr_t)base & 3))) : sa;
  } else {                                       // MODE 4: no other load in flight
    asm volatile(
        "global_load_ubyte %[a], %[z], %[sb] offset:160\n\t"
        "global_load_ubyte %[b], %[z], %[sb]\n\t"
        "s_waitcnt vmcnt(1)\n\t"
        "v_readfirstlane_b32 %[sa], %[a]\n\t"
        "s_waitcnt vmcnt(0)\n\t"
        : [a] "+&v"(a), [b] "+&v"(b), [sa] "=&s"(sa)
        : [z] "v"(z), [sb] "s"(base)
        : "memory");
  }
  out[i] = (a & 0xffffu) | ((b & 0xffu) << 16) | ((sa & 0xffu) << 24);   // want 0x18 | 0x11 << 16 | 0x18 << 24
  if (w == 0x123456789abcdef0ull) sink[0] = w;                           // keeps the word load alive
}

// The same loop body in a kernel that ALLOCATES exactly 8 VGPRs (PAD = 0; v7, the shift, is the last
// one) or 16 (PAD = 1: an empty asm statement names v15).  out[i] = low word of W[i] >> shift:
// 0x83838383 when the loaded shift (0x11) was used, 0x38383838 when v7 still held its preset 5.
// VAR 4: as VAR 1, and the kernel additionally allocates AGPRs it never uses (asm clobber a0), so
// that v7 is no longer the last register of the wave's allocation.
// VAR 0: as generated.  VAR 1: NO load into v7 (it keeps its preset 5; the other two loads stay): is v7
// overwritten from outside?  VAR 2: roles swapped (shift -> v6, width -> v7).  VAR 3: dword loads.
template <int PAD, int VAR = 0>
__global__ void __launch_bounds__(256)
k_probe8(const u64 *__restrict__ W, u32 n, const Table *__restrict__ ptr, u32 idx, u32 *__restrict__ out) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (PAD) asm volatile("" ::: "v15");
  if (i >= n) return;
  const u8 *base = &ptr->shift[idx];
  u32 sa;
  if (VAR == 4) asm volatile("" ::: "a0");        // allocation = 8 VGPRs + one granule of (unused) AGPRs
  if (VAR == 1 || VAR == 4) {
    asm volatile(
        "v_mov_b32 v1, 0\n\t"
        "v_lshlrev_b32 v2, 3, %[i]\n\t"
        "v_mov_b32 v7, 5\n\t"
        "v_mov_b32 v6, 0xaa\n\t"
        "global_load_dwordx2 v[4:5], v2, %[W]\n\t"
        "global_load_ubyte v6, v1, %[sb] offset:160\n\t"
        "global_load_ubyte v3, v1, %[sb]\n\t"
        "s_waitcnt vmcnt(1)\n\t"
        "v_readfirstlane_b32 %[sa], v6\n\t"
        "s_waitcnt vmcnt(0)\n\t"
        "v_lshrrev_b64 v[6:7], v7, v[4:5]\n\t"
        "v_lshlrev_b32 v2, 2, %[i]\n\t"
        "global_store_dword v2, v6, %[out]\n\t"
        : [sa] "=&s"(sa)
        : [i] "v"(i), [W] "s"(W), [sb] "s"(base), [out] "s"(out)
        : "v1", "v2", "v3", "v4", "v5", "v6", "v7", "memory");
    return;
  }
  if (VAR == 2) {
    asm volatile(
        "v_mov_b32 v1, 0\n\t"
        "v_lshlrev_b32 v2, 3, %[i]\n\t"
        "v_mov_b32 v6, 5\n\t"
        "v_mov_b32 v7, 0xaa\n\t"
        "global_load_dwordx2 v[4:5], v2, %[W]\n\t"
        "global_load_ubyte v7, v1, %[sb] offset:160\n\t"
        "global_load_ubyte v6, v1, %[sb]\n\t"
        "s_waitcnt vmcnt(1)\n\t"
        "v_readfirstlane_b32 %[sa], v7\n\t"
        "s_waitcnt vmcnt(0)\n\t"
        "v_lshrrev_b64 v[6:7], v6, v[4:5]\n\t"
        "v_lshlrev_b32 v2, 2, %[i]\n\t"
        "global_store_dword v2, v6, %[out]\n\t"
        : [sa] "=&s"(sa)
        : [i] "v"(i), [W] "s"(W), [sb] "s"(base), [out] "s"(out)
        : "v1", "v2", "v3", "v4", "v5", "v6", "v7", "memory");
    return;
  }
  if (VAR == 3) {
    asm volatile(
        "v_mov_b32 v1, 0\n\t"
        "v_lshlrev_b32 v2, 3, %[i]\n\t"
        "v_mov_b32 v7, 5\n\t"
        "v_mov_b32 v6, 0xaa\n\t"
        "global_load_dwordx2 v[4:5], v2, %[W]\n\t"
        "global_load_dword v6, v1, %[sb] offset:160\n\t"
        "global_load_dword v7, v1, %[sb]\n\t"
        "s_waitcnt vmcnt(1)\n\t"
        "v_readfirstlane_b32 %[sa], v6\n\t"
        "s_waitcnt vmcnt(0)\n\t"
        "v_lshrrev_b64 v[6:7], v7, v[4:5]\n\t"
        "v_lshlrev_b32 v2, 2, %[i]\n\t"
        "global_store_dword v2, v6, %[out]\n\t"
        : [sa] "=&s"(sa)
        : [i] "v"(i), [W] "s"(W), [sb] "s"(base), [out] "s"(out)
        : "v1", "v2", "v3", "v4", "v5", "v6", "v7", "memory");
    return;
  }
  asm volatile(
      "v_mov_b32 v1, 0\n\t"
      "v_lshlrev_b32 v2, 3, %[i]\n\t"
      "v_mov_b32 v7, 5\n\t"
      "v_mov_b32 v6, 0xaa\n\t"
      "global_load_dwordx2 v[4:5], v2, %[W]\n\t"
      "global_load_ubyte v6, v1, %[sb] offset:160\n\t"
      "global_load_ubyte v7, v1, %[sb]\n\t"
      "s_waitcnt vmcnt(1)\n\t"
      "v_readfirstlane_b32 %[sa], v6\n\t"
      "s_waitcnt vmcnt(0)\n\t"
      "v_lshrrev_b64 v[6:7], v7, v[4:5]\n\t"
      "v_lshlrev_b32 v2, 2, %[i]\n\t"
      "global_store_dword v2, v6, %[out]\n\t"
      : [sa] "=&s"(sa)
      : [i] "v"(i), [W] "s"(W), [sb] "s"(base), [out] "s"(out)
      : "v1", "v2", "v3", "v4", "v5", "v6", "v7", "memory");
  (void)sa;
}

// Is it the LAST register of any allocation, or of the 8-register one only?  The preset 5 lives in
// v15 of a kernel that allocates 16 VGPRs (LAST = 15) or in v23 of one that allocates 24 (LAST = 23).
#define STR2(x) #x
#define STR(x) STR2(x)
template <int LAST>
__global__ void __launch_bounds__(256)
k_probe_last(const u64 *__restrict__ W, u32 n, const Table *__restrict__ ptr, u32 idx, u32 *__restrict__ out) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const u8 *base = &ptr->shift[idx];
  u32 sa;
#define LAST_BODY(R)                                                           \
  asm volatile(                                                                \
      "v_mov_b32 v1, 0\n\t"                                                    \
      "v_lshlrev_b32 v2, 3, %[i]\n\t"                                          \
      "v_mov_b32 v" R ", 5\n\t"                                                 \
      "v_mov_b32 v6, 0xaa\n\t"                                                 \
      "global_load_dwordx2 v[4:5], v2, %[W]\n\t"                               \
      "global_load_ubyte v6, v1, %[sb] offset:160\n\t"                         \
      "global_load_ubyte v3, v1, %[sb]\n\t"                                    \
      "s_waitcnt vmcnt(1)\n\t"                                                 \
      "v_readfirstlane_b32 %[sa], v6\n\t"                                      \
      "s_waitcnt vmcnt(0)\n\t"                                                 \
      "v_lshrrev_b64 v[6:7], v" R ", v[4:5]\n\t"                                \
      "v_lshlrev_b32 v2, 2, %[i]\n\t"                                          \
      "global_store_dword v2, v6, %[out]\n\t"                                  \
      : [sa] "=&s"(sa)                                                         \
      : [i] "v"(i), [W] "s"(W), [sb] "s"(base), [out] "s"(out)                 \
      : "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v" R, "memory")
  if (LAST == 15) LAST_BODY("15"); else LAST_BODY("23");
  (void)sa;
}

__global__ void k_filler(const u64 *__restrict__ a, u64 *__restrict__ b, u32 n) {
  for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) b[i] = a[i] * 3 + 1;
}

// bad[0] = wrong lanes; the first wave with a wrong lane is copied whole to bad[1..64] (index in bad[65])
__global__ void k_check(const u32 *__restrict__ out, u32 n, u32 want, u32 *bad) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (out[i] != want) {
    const u32 t = atomicAdd(&bad[0], 1u);
    if (t == 0) bad[65] = i & ~63u;
  }
}

int main(int argc, char **argv) {
  const int launches = argc > 1 ? atoi(argv[1]) : 100;
  const u32 n = 2924858;
  hipStream_t st;
  CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  Table tab;
  memset(&tab, 0, sizeof tab);
  tab.shift[8] = 0x11;
  tab.width[8] = 0x18;
  u64 *dW, *dF0, *dF1, *d_sink;
  u32 *d_out, *d_bad;
  Table *d_tab;
  CHECK(hipMalloc(&dW, (size_t)n * 8));
  CHECK(hipMalloc(&dF0, (size_t)4 << 20 << 3));
  CHECK(hipMalloc(&dF1, (size_t)4 << 20 << 3));
  CHECK(hipMalloc(&d_out, (size_t)n * 4));
  CHECK(hipMalloc(&d_bad, 128 * 4));
  CHECK(hipMalloc(&d_sink, 64));
  CHECK(hipMalloc(&d_tab, sizeof(Table)));
  CHECK(hipMemset(dW, 7, (size_t)n * 8));
  CHECK(hipMemset(dF0, 1, (size_t)4 << 20 << 3));
  CHECK(hipMemcpy(d_tab, &tab, sizeof tab, hipMemcpyHostToDevice));
  const u32 want = 0x18u | (0x11u << 16) | (0x18u << 24);
  const char *names[16] = {"0 two ubyte loads, saddr + zero offset, device table", "1 one ubyte load",
                          "2 two dword loads", "3 two ubyte loads, VGPR address", "4 as 0, no word load in flight",
                          "5 whole loop body, consumer right after the wait", "6 as 5, s_nop before the consumer",
                          "7 as 5, dword loads", "8 loop body in a kernel that allocates  8 VGPRs",
                          "9 loop body in a kernel that allocates 16 VGPRs",
                          "10 8 VGPRs, NO load into v7 (preset 5 -> 38383838)", "11 8 VGPRs, roles swapped (shift in v6)",
                          "12 8 VGPRs, dword loads", "13 16 VGPRs, preset 5 parked in v15 (the last)",
                          "14 24 VGPRs, preset 5 parked in v23 (the last)",
                          "15 as 10 + unused AGPRs allocated behind v7"};
  std::vector<u32> h_out(n);
  for (int mode = (argc > 2 ? atoi(argv[2]) : 0); mode < 16; mode++) {
    const u32 want_m = (mode == 10 || mode >= 13) ? 0x38383838u : mode >= 8 ? 0x83838383u : want;
    u32 bad_launches = 0, shown = 0;
    unsigned long long bad_lanes = 0;
    for (int l = 0; l < launches; l++) {
      hipLaunchKernelGGL(k_filler, dim3(2048), dim3(256), 0, st, dF0, dF1, 4u << 20);
      CHECK(hipMemsetAsync(d_bad, 0, 128 * 4, st));
      CHECK(hipMemsetAsync(d_out, 0xff, (size_t)n * 4, st));
      const dim3 grid((n + 255) / 256), blk(256);
      switch (mode) {
        case 0: hipLaunchKernelGGL(k_probe<0>, grid, blk, 0, st, dW, n, tab, d_tab, 8u, d_out, d_sink); break;
        case 1: hipLaunchKernelGGL(k_probe<1>, grid, blk, 0, st, dW, n, tab, d_tab, 8u, d_out, d_sink); break;
        case 2: hipLaunchKernelGGL(k_probe<2>, grid, blk, 0, st, dW, n, tab, d_tab, 8u, d_out, d_sink); break;
        case 3: hipLaunchKernelGGL(k_probe<3>, grid, blk, 0, st, dW, n, tab, d_tab, 8u, d_out, d_sink); break;
        case 4: hipLaunchKernelGGL(k_probe<4>, grid, blk, 0, st, dW, n, tab, d_tab, 8u, d_out, d_sink); break;
        case 5: hipLaunchKernelGGL(k_probe<5>, grid, blk, 0, st, dW, n, tab, d_tab, 8u, d_out, d_sink); break;
        case 6: hipLaunchKernelGGL(k_probe<6>, grid, blk, 0, st, dW, n, tab, d_tab, 8u, d_out, d_sink); break;
        case 7: hipLaunchKernelGGL(k_probe<7>, grid, blk, 0, st, dW, n, tab, d_tab, 8u, d_out, d_sink); break;
        case 8: hipLaunchKernelGGL(k_probe8<0>, grid, blk, 0, st, dW, n, d_tab, 8u, d_out); break;
        case 9: hipLaunchKernelGGL(k_probe8<1>, grid, blk, 0, st, dW, n, d_tab, 8u, d_out); break;
        case 10: hipLaunchKernelGGL((k_probe8<0, 1>), grid, blk, 0, st, dW, n, d_tab, 8u, d_out); break;
        case 11: hipLaunchKernelGGL((k_probe8<0, 2>), grid, blk, 0, st, dW, n, d_tab, 8u, d_out); break;
        case 12: hipLaunchKernelGGL((k_probe8<0, 3>), grid, blk, 0, st, dW, n, d_tab, 8u, d_out); break;
        case 13: hipLaunchKernelGGL(k_probe_last<15>, grid, blk, 0, st, dW, n, d_tab, 8u, d_out); break;
        case 14: hipLaunchKernelGGL(k_probe_last<23>, grid, blk, 0, st, dW, n, d_tab, 8u, d_out); break;
        default: hipLaunchKernelGGL((k_probe8<0, 4>), grid, blk, 0, st, dW, n, d_tab, 8u, d_out); break;
      }
      hipLaunchKernelGGL(k_check, grid, blk, 0, st, d_out, n, want_m, d_bad);
      u32 hb[128];
      CHECK(hipMemcpyAsync(hb, d_bad, 128 * 4, hipMemcpyDeviceToHost, st));
      CHECK(hipStreamSynchronize(st));
      if (hb[0]) {
        bad_launches++;
        bad_lanes += hb[0];
        if (shown < 2) {
          shown++;
          u32 wv[64];
          CHECK(hipMemcpy(wv, d_out + hb[65], 64 * 4, hipMemcpyDeviceToHost));
          printf("   mode %d launch %d: %u wrong lanes; wave at i = %u (want %08x), lanes 0..63:\n     ", mode, l, hb[0],
                 hb[65], want_m);
          for (int k = 0; k < 64; k++) printf("%08x%s", wv[k], (k & 7) == 7 ? "\n     " : " ");
          printf("\n");
        }
      }
    }
    printf("mode %-55s launches %d: with wrong lanes %u (wrong lanes in total %llu)\n", names[mode], launches, bad_launches,
           bad_lanes);
    fflush(stdout);
  }
  return 0;
}
