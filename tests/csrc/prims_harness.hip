// prims_harness.hip -- TEST INFRASTRUCTURE, not part of the product library.
// The device-wide scan and radix sort of humid_amd/csrc/prims.hip.h behind a plain C interface, so that
// tests/test_gpu_prims.py can drive them at the sizes where their forms change (one workgroup / the
// chained single launch with 1, 2, 4, 8, 16 items per thread / three launches; one tile / many tiles;
// partial last pass of the sort).  The pipeline reaches most of these sizes only with specific inputs.
// Built by tests/prims_harness.py into tests/_build/ (git-ignored; travels to the GPU box).
#include <hip/hip_runtime.h>

#include "prims.hip.h"

namespace {
PsChain *g_chain = nullptr;
u32 g_epoch = 0;
void *g_tmp = nullptr;
size_t g_tmp_bytes = 0;

int grow(size_t bytes) {
  if (bytes <= g_tmp_bytes) return 0;
  if (g_tmp) (void)hipFree(g_tmp);
  g_tmp = nullptr;
  g_tmp_bytes = 0;
  hipError_t e = hipMalloc(&g_tmp, bytes);
  if (e != hipSuccess) return -1000 - (int)e;
  g_tmp_bytes = bytes;
  return 0;
}
int chain_ready() {
  if (g_chain) return 0;
  hipError_t e = hipMalloc((void **)&g_chain, sizeof(PsChain));
  if (e != hipSuccess) return -2000 - (int)e;
  e = hipMemset(g_chain, 0, sizeof(PsChain));
  return e == hipSuccess ? 0 : -3000 - (int)e;
}
template <class T>
int exscan(const T *in, T *out, u64 n, int chain) {
  int rc;
  if (chain && (rc = chain_ready())) return rc;
  if ((rc = grow(ps_scan_scratch_items(n) * sizeof(T) + 256))) return rc;
  if (ps_exscan<T>(PtrIn<T>{in}, out, n, (T *)g_tmp, nullptr, chain ? g_chain : nullptr, &g_epoch) != hipSuccess) return -2;
  return hipStreamSynchronize(nullptr) == hipSuccess ? 0 : -3;
}
template <class K>
int sort(const K *kin, K *kout, const u32 *vin, u32 *vout, u64 n, u32 b0, u32 b1, int has_v, int iota) {
  hipError_t e;
  if (has_v) {
    if (int rc = grow(rs_temp_bytes<K, u32, true>(n))) return rc;
    if (iota)
      e = rs_sort<K, u32, true>(g_tmp, PtrIn<K>{kin}, kout, IotaIn{}, vout, n, b0, b1, nullptr);
    else
      e = rs_sort<K, u32, true>(g_tmp, PtrIn<K>{kin}, kout, PtrIn<u32>{vin}, vout, n, b0, b1, nullptr);
  } else {
    if (int rc = grow(rs_temp_bytes<K, u32, false>(n))) return rc;
    e = rs_sort<K, u32, false>(g_tmp, PtrIn<K>{kin}, kout, IotaIn{}, (u32 *)nullptr, n, b0, b1, nullptr);
  }
  if (e != hipSuccess) return -2;
  return hipStreamSynchronize(nullptr) == hipSuccess ? 0 : -3;
}
}  // namespace

extern "C" {
// the epoch counter of the chained scan, to start a test just below its wrap-around
void ph_set_epoch(unsigned e) { g_epoch = e; }
unsigned ph_epoch() { return g_epoch; }
int ph_exscan_u32(const unsigned *in, unsigned *out, unsigned long long n, int chain) { return exscan<u32>(in, out, n, chain); }
int ph_exscan_u64(const unsigned long long *in, unsigned long long *out, unsigned long long n, int chain) {
  return exscan<u64>((const u64 *)in, (u64 *)out, n, chain);
}
int ph_sort_u32(const unsigned *kin, unsigned *kout, const unsigned *vin, unsigned *vout, unsigned long long n, unsigned b0,
                unsigned b1, int has_v, int iota) {
  return sort<u32>(kin, kout, vin, vout, n, b0, b1, has_v, iota);
}
int ph_sort_u64(const unsigned long long *kin, unsigned long long *kout, const unsigned *vin, unsigned *vout, unsigned long long n,
                unsigned b0, unsigned b1, int has_v, int iota) {
  return sort<u64>((const u64 *)kin, (u64 *)kout, vin, vout, n, b0, b1, has_v, iota);
}
void ph_release() {
  if (g_tmp) (void)hipFree(g_tmp);
  if (g_chain) (void)hipFree(g_chain);
  g_tmp = nullptr; g_chain = nullptr; g_tmp_bytes = 0;
}
}
