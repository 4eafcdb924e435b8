// vgpr_clobber_probe.hip -- is the LAST allocated VGPR of a small kernel overwritten from outside?
//
// The keys kernel that lost edges (tools/plan_arg_repro.hip K1/K2) allocates
// exactly 8 VGPRs and keeps the loaded shift amount in v7, the last one.  In the affected waves
// every lane shifted by its own lane number -- the low bits of a work-item id -- although no
// instruction of the kernel computes such a value, and the same instruction sequence in a kernel
// with 16 allocated VGPRs (tools/uniform_load_probe.hip) never failed.  This probe parks a sentinel
// in the last allocated register of a kernel with 8 (and with 16) VGPRs, spins a while -- other waves
// are launched meanwhile -- and stores what the register holds afterwards.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/vgpr_clobber_probe tools/vgpr_clobber_probe.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef uint32_t u32;
#define CHECK(x)                                                                              \
  do {                                                                                        \
    hipError_t e_ = (x);                                                                      \
    if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } \
  } while (0)

#define SENT 0x5a5a5a5au

template <int NREG>   // 8: sentinel in v7, 16: in v15 (and v7)
__global__ void __launch_bounds__(256) k_park(u32 n, u32 spins, u32 *__restrict__ out) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u32 got7, got_last = SENT;
  if (NREG == 8) {
    asm volatile(
        "v_mov_b32 v7, 0x5a5a5a5a\n\t"
        "s_mov_b32 s20, %[sp]\n\t"
        "1:\n\t"
        "s_sleep 1\n\t"
        "s_sub_u32 s20, s20, 1\n\t"
        "s_cmp_lg_u32 s20, 0\n\t"
        "s_cbranch_scc1 1b\n\t"
        "v_mov_b32 %[g7], v7\n\t"
        : [g7] "=v"(got7)
        : [sp] "s"(spins)
        : "v7", "s20", "scc", "memory");
    got_last = got7;
  } else {
    asm volatile(
        "v_mov_b32 v7, 0x5a5a5a5a\n\t"
        "v_mov_b32 v15, 0x5a5a5a5a\n\t"
        "s_mov_b32 s20, %[sp]\n\t"
        "1:\n\t"
        "s_sleep 1\n\t"
        "s_sub_u32 s20, s20, 1\n\t"
        "s_cmp_lg_u32 s20, 0\n\t"
        "s_cbranch_scc1 1b\n\t"
        "v_mov_b32 %[g7], v7\n\t"
        "v_mov_b32 %[gl], v15\n\t"
        : [g7] "=v"(got7), [gl] "=v"(got_last)
        : [sp] "s"(spins)
        : "v7", "v15", "s20", "scc", "memory");
  }
  out[2 * (size_t)i] = got7;
  out[2 * (size_t)i + 1] = got_last;
}

int main(int argc, char **argv) {
  const int launches = argc > 1 ? atoi(argv[1]) : 50;
  const u32 spins = argc > 2 ? (u32)atoi(argv[2]) : 40;
  const u32 n = 2924858;
  u32 *d_out;
  CHECK(hipMalloc(&d_out, (size_t)n * 8));
  std::vector<u32> h((size_t)n * 2);
  for (int nreg = 8; nreg <= 16; nreg += 8) {
    unsigned long long bad7 = 0, badl = 0;
    u32 bad_launches = 0, shown = 0;
    for (int l = 0; l < launches; l++) {
      CHECK(hipMemset(d_out, 0, (size_t)n * 8));
      const dim3 grid((n + 255) / 256), blk(256);
      if (nreg == 8) hipLaunchKernelGGL(k_park<8>, grid, blk, 0, 0, n, spins, d_out);
      else hipLaunchKernelGGL(k_park<16>, grid, blk, 0, 0, n, spins, d_out);
      CHECK(hipMemcpy(h.data(), d_out, (size_t)n * 8, hipMemcpyDeviceToHost));
      bool any = false;
      for (u32 i = 0; i < n; i++) {
        const bool b7 = h[2 * (size_t)i] != SENT, bl = h[2 * (size_t)i + 1] != SENT;
        if (b7 || bl) {
          bad7 += b7; badl += bl; any = true;
          if (shown < 12) { shown++; printf("   %d VGPRs launch %d: i = %u (work-item %u, lane %u): v7 = %08x last = %08x\n", nreg, l, i, i & 255, i & 63, h[2 * (size_t)i], h[2 * (size_t)i + 1]); }
        }
      }
      bad_launches += any;
    }
    printf("kernel with %2d VGPRs: launches %d, with a changed sentinel %u; lanes with v7 changed %llu, with the last register changed %llu\n",
           nreg, launches, bad_launches, bad7, badl);
    fflush(stdout);
  }
  return 0;
}
