// sort_probe.hip -- which rocPRIM Onesweep configuration sorts the combination keys fastest?
// 2.7 M (key, rank) pairs, 24 key bits (the d = 1 second-half key), as in stage_graph.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/sort_probe tools/sort_probe.hip && tools/sort_probe
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <class Cfg, class K>
static void run(const char *name, K *kin, K *kout, unsigned *vin, unsigned *vout, size_t n, unsigned bits) {
  size_t bytes = 0;
  CK(rocprim::radix_sort_pairs<Cfg>(nullptr, bytes, kin, kout, vin, vout, n, 0, bits, 0));
  void *tmp;
  CK(hipMalloc(&tmp, bytes));
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  for (int i = 0; i < 3; i++) CK(rocprim::radix_sort_pairs<Cfg>(tmp, bytes, kin, kout, vin, vout, n, 0, bits, 0));
  CK(hipEventRecord(a, 0));
  const int reps = 20;
  for (int i = 0; i < reps; i++) CK(rocprim::radix_sort_pairs<Cfg>(tmp, bytes, kin, kout, vin, vout, n, 0, bits, 0));
  CK(hipEventRecord(b, 0));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  printf("%-44s key %zu B, %u bits: %.1f us per sort\n", name, sizeof(K), bits, 1e3f * ms / reps);
  CK(hipFree(tmp));
}

template <unsigned RB, unsigned BS, unsigned IPT, rocprim::block_radix_rank_algorithm ALG = rocprim::block_radix_rank_algorithm::match>
using os_cfg = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                          rocprim::radix_sort_onesweep_config<rocprim::kernel_config<256, 12>, rocprim::kernel_config<BS, IPT>, RB, ALG>, 0>;
using def_cfg = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 0>;

int main() {
  const size_t n = 2722206;
  std::vector<unsigned long long> h64(n);
  std::vector<unsigned> h32(n), hv(n);
  unsigned long long x = 88172645463325252ull;
  for (size_t i = 0; i < n; i++) {
    x ^= x << 13; x ^= x >> 7; x ^= x << 17;
    h64[i] = x & 0xffffffffffffull;
    h32[i] = (unsigned)(x & 0xffffff);
    hv[i] = (unsigned)i;
  }
  unsigned long long *k64, *o64;
  unsigned *k32, *o32, *v, *vo;
  CK(hipMalloc(&k64, n * 8)); CK(hipMalloc(&o64, n * 8)); CK(hipMalloc(&k32, n * 4)); CK(hipMalloc(&o32, n * 4));
  CK(hipMalloc(&v, n * 4)); CK(hipMalloc(&vo, n * 4));
  CK(hipMemcpy(k64, h64.data(), n * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(k32, h32.data(), n * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(v, hv.data(), n * 4, hipMemcpyHostToDevice));
  run<def_cfg>("default (library tuning)", k32, o32, v, vo, n, 24);
  run<def_cfg>("default (library tuning)", k64, o64, v, vo, n, 24);
  run<os_cfg<8, 256, 12>>("radix 8, 256 x 12, match", k32, o32, v, vo, n, 24);
  run<os_cfg<8, 512, 12>>("radix 8, 512 x 12, match", k32, o32, v, vo, n, 24);
  run<os_cfg<8, 1024, 8>>("radix 8, 1024 x 8, match", k32, o32, v, vo, n, 24);
  run<os_cfg<8, 1024, 4>>("radix 8, 1024 x 4, match", k32, o32, v, vo, n, 24);
  run<os_cfg<6, 512, 12>>("radix 6, 512 x 12, match (4 places)", k32, o32, v, vo, n, 24);
  run<os_cfg<8, 512, 12>>("radix 8, 512 x 12, match", k64, o64, v, vo, n, 24);
  run<os_cfg<8, 1024, 6>>("radix 8, 1024 x 6, match", k64, o64, v, vo, n, 24);
  run<os_cfg<8, 1024, 4>>("radix 8, 1024 x 4, match", k64, o64, v, vo, n, 24);
  run<os_cfg<8, 256, 12>>("radix 8, 256 x 12, match", k64, o64, v, vo, n, 24);
  return 0;
}
