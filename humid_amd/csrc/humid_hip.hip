// humid_hip.hip -- HUMID's neighbour-search-and-cluster hot path for MI355X (gfx950): the host
// side of libhumid_hip.so (context, stages, C ABI of include/humid_hip.h).  Kernels live in the
// headers included below.
//
// Pipeline (all device-side; integer/bit work, HBM/latency bound, no MFMA):
//   A. exact counts   kernels_count.hip.h   reads radix-partitioned by mix64(word), one LDS-resident
//                     open-address table per bucket (k_dedup_lds); fallback: one table in HBM
//                     (k_hash_insert).  Replaces Trie::add, /root/reference/src/humid.cc:95.
//                     Unique words are then sorted: Trie::walk() order.
//   B. graph          kernels_graph.hip.h   generalised pigeonhole buckets (k_combo_keys, k_pairs),
//                     CSR rows through cursors + per-row sort, union-find components.  Replaces
//                     walk x asymmetricHamming, src/humid.cc:113-130.
//                     kernels_cluster.hip.h per component: the findClusters loop + src/cluster.cc,
//                     order-exact (k_cluster_trivial / _small / _components); ids = prefix sum over
//                     creators.  Replaces src/humid.cc:167-193.
//   C. map            kernels_map.hip.h     per read (cluster_id, keep); replaces
//                     trie.find()->leaf->cluster, src/humid.cc:223-231,276-277.  Also the
//                     multi-GPU result return.
// Sorts and scans are the library's own (prims.hip.h): no third-party device code is linked in, so
// every kernel of the code object carries the last-VGPR guard (tests/test_cabi_symbols.py reads the
// code object's kernel descriptors).  No CPU fallback lives here: every entry point either runs on
// the GPU or fails.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <cerrno>
#include <chrono>

#include "common.hip.h"
#include "prims.hip.h"
#include "kernels_count.hip.h"
#include "kernels_part.hip.h"
#include "kernels_part8.hip.h"
#include "kernels_graph.hip.h"
#include "kernels_cluster.hip.h"
#include "kernels_cgraph.hip.h"
#include "kernels_map.hip.h"
#include "kernels_xchg.hip.h"
#include "kernels_wide.hip.h"

// --------------------------------------------------------------------------------
// host side
// --------------------------------------------------------------------------------
// One slab of device memory a context may hold (humid_ctx_reserve): buffers are carved out of it
// with a bump pointer instead of one hipMalloc each -- a first run needs ~35 buffers and every
// hipMalloc costs about a millisecond, which is most of what the `humid` command line spends between
// "pass 1 done" and "device path done" on 10 M reads.  Nothing is returned to the slab; a buffer
// that outgrows its carving gets a new one (slab or hipMalloc).
struct Arena {
  char *base = nullptr;
  size_t size = 0, used = 0;
  void *take(size_t bytes) {
    const size_t at = (used + 255) & ~(size_t)255;
    if (!base || at + bytes > size) return nullptr;
    used = at + bytes;
    return base + at;
  }
};

struct DBuf {
  void *p = nullptr;
  size_t cap = 0;
  bool in_arena = false;
  hipError_t ensure(size_t bytes, Arena *arena = nullptr) {
    if (bytes <= cap) return hipSuccess;
    if (p && !in_arena) (void)hipFree(p);
    p = nullptr; cap = 0; in_arena = false;
    size_t want = bytes + bytes / 8 + 256;
    if (arena) {
      if (void *q = arena->take(bytes + 256)) { p = q; cap = bytes + 256; in_arena = true; return hipSuccess; }
    }
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) { p = nullptr; return e; }
    cap = want;
    return hipSuccess;
  }
  void release() { if (p && !in_arena) (void)hipFree(p); p = nullptr; cap = 0; in_arena = false; }
  template <class T> T *as() const { return (T *)p; }
};

struct humid_ctx {
  int device = 0;
  Arena arena;               // humid_ctx_reserve
  hipStream_t stream = nullptr;
  bool own_stream = false;
  std::string err;
  ull *d_ctr = nullptr;
  PsChain *ps_chain = nullptr;   // behind the counters
  u32 ps_epoch = 0;
  ull *h_ctr = nullptr;   // pinned mirror (CTR_N counters + the sequence word of read_counters)
  ull *h_ctr_dev = nullptr;   // the same memory as the device sees it
  ull ctr_seq = 0;
  const u32 *gf_valid = nullptr; // set by the last bucket order: device count of the words it holds (padded grouping), or null
  u32 *ucur_clean = nullptr;     // the un-permute's bin cursors at this address are all zero
  DBuf gf_cur;                    // cursors of the padded grouping (512 u32, kept at zero between uses)
  bool gf_padded = true;  // bucket orders of the compact graph stage through padded coarse bins (until one was full)
  bool no_chain = false;  // HUMID_NO_SCAN_CHAIN: scans without k_ps_scan_chain (experiment / cross-check)
  bool no_poll = false;   // HUMID_NO_POLL / a failed first try: blit copies + stream wait instead
  DBuf in_words, in_filt, in_bases, out_cid, out_keep;       // host entry point staging
  DBuf table, slot_out, slot_of_read, uniq_slot;             // table (cap+1) and per-read
  DBuf pk_keys, pk_vals, pbeg, ucount, pusable, ubase, pad_word, pad_cf, pslot;   // partitioned counts
  DBuf opos, own_packed, owner, owner_sorted, perm, small;                        // multi-GPU result return
  DBuf pc, poff, share_edges;                                                     // multi-GPU pair-search share
  DBuf own_words;                                                                 // multi-GPU dense count
  DBuf heads;                                                                     // big-component heads
  DBuf small_roots;         // k_comp_count: roots of the components of 3 .. 32 leaves (k_cluster_small works off this list)
  DBuf big_runs;            // k_big_runs: (start, length, first tile) of the buckets beyond k_pairs' walk, per combination
  DBuf had;                 // k_pairs: per combination and position, pairs found in the first phase (<< 24) | distance to the first one
  DBuf e_kx, e_vx, e_ky, e_vy, e_raw, e_sorted, e_edges, e_head, e_hpos;   // edit-distance neighbour search
  DBuf e_runlo, e_nch, e_choff, e_pc2, e_poff2;                            // ... its long runs in pieces
  bool edit = false;         // option "edit_distance": Levenshtein instead of Hamming neighbours (-e)
  DBuf xr_heads, xr_send, xr_zero;                             // the same for two-word words: heads, routed words, an all-usable flag array
  DBuf xr_hist, xr_recv, xr_eloc, xr_got, xr_eall, xr_ret;   // humid_dedup_run_exchange: histogram, received words, pair records, received items, results
  DBuf x_slot, x_slot_s, x_cnt, x_cnts, x_rec, x_ncnt, x_route, x_creator, x_base, x_mark, x_markcr, x_scan, x_lcid, x_lismax,
       x_items, x_w, x_id, x_ids, x_ends, x_ends_s, x_head, x_hpos, x_nodes, x_cedges;   // multi-GPU exchange mode
  DBuf w_sorted, w_head, w_hpos, w_start, w_heads;                                         // wide-word (sorted) counts
  // compact graph (kernels_cgraph.hip.h): pair regions + cursors, the two bitmaps with their rank blocks, per-node arrays
  DBuf cg_edges, cg_cur, cg_far, cg_bits, cg_nbits, cg_blk, cg_nblk, cg_nodes, cg_ncnt, cg_deg, cg_off, cg_idx, cg_parent, cg_csize,
       cg_curs, cg_cl_of, cg_maxleaf, cg_cl_size;
  u64 cg_ecap = 0;                  // room for pairs in the append regions (remembered from pass to pass; grown on demand)
  bool use_compact = true;          // option "compact_graph": 0 = the per-unique-word graph of rounds 1-2
  bool cg_valid = false;            // the last graph stage left its results in the cg_* arrays ...
  bool cg_expanded = false;         // ... and the per-unique-word view of them has been built (accessors)
  u32 cg_M = 0, cg_nblocks = 0;
  // owner-local clustering of the exchange pass (kernels_xchg.hip.h): records by destination, interior / crossing /
  // flagged-interior records, the forest over own leaves, crossing-creator bitmap and ids, own results
  DBuf xo_gw, xo_gc;                // the edit-distance road: unique words / counts of all ranks
  DBuf xo_regs, xo_inv;             // record regions of the pair search; routed position of every read
  u64 xr_ecap = 0;                  // room for pair records in the regions (remembered from pass to pass)
  DBuf xo_send, xo_int, xo_cross, xo_sel, xo_selall, xo_parent, xo_flag, xo_xroot, xo_xcbits, xo_xcblk, xo_xcid, xo_xcall, xo_ldeg, xo_cnt;
  DBuf pw_a, pw_ai, pw_b, pw_bi;    // two-word words: (word, read index) records of the two partition levels
  DBuf p8_a, p8_b, p8_cur, p8_status;               // 8-byte records of the count stage: level-1 output, level-2 output (kernels_part8.hip.h)
  bool use_rec8 = true;             // option "records8": 0 = always the 12-byte (key, read) pairs of kernels_part.hip.h
  bool last_rec8 = false;           // the last count ran on records: positions are (bucket << 9 | j), the un-permute reads p8_b
  const u32 *rec_cursor2 = nullptr; // reads per bucket of that count
  DBuf pt_work, unperm_rec, route_tiles;                                                     // LDS-staged partition / un-permute (kernels_part.hip.h)
  bool group_buckets = true;        // option "group_buckets": bucket order of stretch keys by two-level grouping instead of a library sort
  bool pt_padded = true;            // level 1 of the tile partition into padded coarse bins (no histogram pass); false after an overflow
  bool use_tile_partition = true;   // option "tile_partition": 0 = library radix passes + one-kernel un-permute (round 1)
  bool last_part_tiled = false;     // kev[39]..kev[40] bracket the second-level scatter of the last count
  int x_test_fail_after = -1, x_gathers = 0;   // option "test_fail_before_gather": this rank leaves the pass with an error in the compute phase before its k-th gather (tests)
  bool x_hist_done = false, x_peer_failed = false;   // humid_dedup_run_exchange: the pass's first gather is done; a peer's failure was seen
  bool route_checked = true;        // no humid_stage_route since the last humid_stage_route_check
  const u32 *route_bad = nullptr;   // device flag of the last humid_stage_route
  bool last_unperm_tiled = false;   // kev[36]..kev[41] bracket k_unperm_window of the last map
  // cached answer of prefix_fits_ordered for (reads, word length, key map): the sampled histogram and
  // its host wait run once per shape, not once per pass; an overflowing ordered run resets it
  bool oc_valid = false, oc_fits = false;
  u32 oc_n = 0, oc_nt = 0;
  u64 oc_lo = 0, oc_scale = 0;
  u32 n_parts = 0;           // buckets of the last LDS-partitioned count (0: none, e.g. the sorted wide count)
  bool stage_map_timed = false;                                                   // kev[37..38] bracket the last humid_stage_map_dense
  bool last_count_sorted = false;                                                 // last count was the wide-word sort
  u32 g_wpr = 1;                                                                  // uint64 per word of g_word
  bool dense_mode = false;   // last count ran on a compacted list of this rank's reads
  bool slots_done = false;   // slot_out already written by k_finalize_nodes (one-GPU fusion)
  int count_mode = 0;        // 0: hash-partitioned LDS tables (default), 1: one global HBM table
  u32 force_segments = 0;    // 0: automatic pigeonhole plan; else the number of segments s
  bool force_comm = false;   // humid_dedup_run_exchange: call the humid_comm callbacks even with one rank (transport tests)
  u32 walk_max = PT2_TILE;   // k_pairs compares a position with this many followers; longer buckets go to k_pairs_tiles (0: never)
  bool coop_big = true;      // big components: workgroup-cooperative kernel (directional method)
  bool last_count_lds = false;
  bool last_count_ordered = false;
  int count_order = -1;      // LDS buckets by word prefix: -1 automatic (uniform prefix), 0 never, 1 always
  DBuf uniq_word, s_word, s_slot, s_cnt, s_first;            // unique words (walk order)
  DBuf deg, nbr_off, nbr_idx, seg_k0, seg_v0, seg_ks, seg_vs, seg_ws, csize, cur;
  DBuf parent, mk0, mk1, cl_of, maxleaf, cl_size, flag, pos, cid, ismax, stk, tmp, scratch;
  hipEvent_t ev[6] = {};
  bool lean_events = false;  // set by run_device while the per-kernel timing is off: only ev[0], ev[4] and the count kernel's pair are recorded
                             // (an event record between two kernels is a marker the second one waits behind: ~4 us of idle GPU each, 8 per pass)
  bool kev_on = false;       // option "kernel_timing": events around the single kernels beyond the count kernel's kev[0..1] (13 more records per pass: 20-45 us)
  hipEvent_t kev[44] = {};   // per-kernel timing: [0,1] insert, [2,3] cluster, [4..19] pairs fill, [20..35] pairs count
  bool have_run = false;     // a full dedup run completed (all accessors valid)
  bool have_graph = false;   // stage B completed (leaf/adjacency/cluster accessors valid)
  bool graph_mode = false;   // last call was humid_cluster_graph
  const void *g_word = nullptr;  // arrays stage B ran on (u64 or W2 per word)
  const u32 *g_cnt = nullptr;
  u32 gU = 0;
  u32 cap_log2 = 0;
  u64 N = 0, U = 0, E = 0, M = 0, C = 0, usable = 0;
  u32 word_nt = 0, distance = 0, method = 0;
};

static std::string g_err;
// (the error text of calls without a context: also set from shm.cpp)
extern "C" __attribute__((visibility("hidden"))) void humid_set_global_error(const char *text) { g_err = text ? text : ""; }

static int fail(humid_ctx *c, int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (c) c->err = buf; else g_err = buf;
  return code;
}

#define HIPCHK(expr)                                                                        \
  do {                                                                                      \
    hipError_t _e = (expr);                                                                 \
    if (_e != hipSuccess)                                                                   \
      return fail(c, _e == hipErrorOutOfMemory ? HUMID_E_NOMEM : HUMID_E_HIP, "%s: %s (%s:%d)", \
                  #expr, hipGetErrorString(_e), __FILE__, __LINE__);                        \
  } while (0)

#define ENSURE(buf, bytes) HIPCHK((buf).ensure((bytes), &c->arena))

static inline u32 blocks_for(u64 n, u32 bs = 256) { return (u32)((n + bs - 1) / bs); }
static inline u32 grid_stride_blocks(u64 n, u32 bs = 256) {
  u64 b = (n + bs - 1) / bs;
  if (b > 256 * 8) b = 256 * 8;
  if (b == 0) b = 1;
  return (u32)b;
}
static inline u32 bits_for(u64 n) {  // bits needed to represent values < n
  u32 b = 0;
  while (b < 64 && ((u64)1 << b) < n) b++;
  return b ? b : 1;
}

// ---- sort / scan wrappers over prims.hip.h (temporary storage grown on demand) ----------
template <class K, class V, class KIn, class VIn>
static int sort_pairs_in(humid_ctx *c, KIn kin, K *kout, VIn vin, V *vout, u64 n, u32 b0, u32 b1) {
  if (n == 0) return HUMID_OK;
  if (n > 0xffffffffull) return fail(c, HUMID_E_OVERFLOW, "sort of more than 2^32-1 items");
  ENSURE(c->tmp, (rs_temp_bytes<K, V, true>(n)));
  HIPCHK((rs_sort<K, V, true>(c->tmp.p, kin, kout, vin, vout, n, b0, b1, c->stream)));
  return HUMID_OK;
}
template <class K, class V>
static int sort_pairs(humid_ctx *c, const K *kin, K *kout, const V *vin, V *vout, u64 n, u32 b0, u32 b1) {
  return sort_pairs_in<K, V>(c, PtrIn<K>{kin}, kout, PtrIn<V>{vin}, vout, n, b0, b1);
}
template <class K>
static int sort_keys(humid_ctx *c, const K *kin, K *kout, u64 n, u32 b0, u32 b1) {
  if (n == 0) return HUMID_OK;
  if (n > 0xffffffffull) return fail(c, HUMID_E_OVERFLOW, "sort of more than 2^32-1 items");
  ENSURE(c->tmp, (rs_temp_bytes<K, u32, false>(n)));
  HIPCHK((rs_sort<K, u32, false>(c->tmp.p, PtrIn<K>{kin}, kout, IotaIn{}, (u32 *)nullptr, n, b0, b1, c->stream)));
  return HUMID_OK;
}
template <class T, class In>
static int exscan_in(humid_ctx *c, In in, T *out, u64 n) {
  ENSURE(c->tmp, ps_scan_scratch_items(n) * sizeof(T) + 256);
  HIPCHK((ps_exscan<T>(in, out, n, (T *)c->tmp.p, c->stream, c->no_chain ? nullptr : c->ps_chain, &c->ps_epoch)));
  return HUMID_OK;
}
static int exscan_u32(humid_ctx *c, const u32 *in, u32 *out, u64 n) { return exscan_in<u32>(c, PtrIn<u32>{in}, out, n); }

// device counters -> pinned mirror, one stream sync.  extra32 (device u32, may be null) lands
// in h_ctr[CTR_N - 1].
// One tiny kernel stores the counters (and the extra value) straight into the page-locked mirror and then
// a sequence number; the host watches that word.  Two blit copies + hipStreamSynchronize cost ~30 us of idle
// GPU per host wait, this ~10 (three waits per single-GPU pass, eight in the multi-GPU pass).
__global__ void k_publish_counters(const ull *__restrict__ ctr, const u32 *__restrict__ extra32, const u32 *__restrict__ extra32b,
                                   volatile ull *host, ull seq, const u32 *__restrict__ extra32c = nullptr) {
  HUMID_GUARD_LAST_VGPR();
  if (threadIdx.x < CTR_N) {
    ull v = ctr[threadIdx.x];
    if (threadIdx.x == CTR_N - 1 && extra32) v = (v & ~0xffffffffull) | (ull)*extra32;
    if (threadIdx.x == CTR_N - 2 && extra32b) v = (ull)*extra32b;
    if (threadIdx.x == CTR_N - 3 && extra32c) v = (ull)*extra32c;
    host[threadIdx.x] = v;
  }
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) { host[CTR_N] = seq; __threadfence_system(); }
}
// extra32 -> h_ctr[CTR_N - 1] (low half), extra32b -> h_ctr[CTR_N - 2], extra32c -> h_ctr[CTR_N - 3]
static int read_counters(humid_ctx *c, const u32 *extra32 = nullptr, const u32 *extra32b = nullptr, const u32 *extra32c = nullptr) {
  if (c->h_ctr_dev && !c->no_poll) {
    const ull seq = ++c->ctr_seq;
    hipLaunchKernelGGL(k_publish_counters, dim3(1), dim3(64), 0, c->stream, (const ull *)c->d_ctr, extra32, extra32b,
                       (volatile ull *)c->h_ctr_dev, seq, extra32c);
    HIPCHK(hipGetLastError());
    volatile ull *flag = (volatile ull *)&c->h_ctr[CTR_N];
    const auto t0 = std::chrono::steady_clock::now();
    u32 spins = 0;
    while (*flag != seq) {
      if ((++spins & 0xfffu) == 0) {
        if (hipStreamQuery(c->stream) != hipErrorNotReady) break;              // drained (the stores are done or lost) or failed: settled below
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(30)) break;
      }
    }
    if (*flag == seq) { std::atomic_thread_fence(std::memory_order_acquire); return HUMID_OK; }
    HIPCHK(hipStreamSynchronize(c->stream));                                   // an error of an earlier kernel surfaces here
    if (*flag == seq) return HUMID_OK;
    c->no_poll = true;                                                          // mapped stores not visible on this system: copies from now on
  }
  HIPCHK(hipMemcpyAsync(c->h_ctr, c->d_ctr, CTR_N * sizeof(ull), hipMemcpyDeviceToHost, c->stream));
  if (extra32)
    HIPCHK(hipMemcpyAsync(&c->h_ctr[CTR_N - 1], extra32, 4, hipMemcpyDeviceToHost, c->stream));
  if (extra32b || extra32c) {
    HIPCHK(hipStreamSynchronize(c->stream));
    if (extra32b) { c->h_ctr[CTR_N - 2] = 0; HIPCHK(hipMemcpyAsync(&c->h_ctr[CTR_N - 2], extra32b, 4, hipMemcpyDeviceToHost, c->stream)); }
    if (extra32c) { c->h_ctr[CTR_N - 3] = 0; HIPCHK(hipMemcpyAsync(&c->h_ctr[CTR_N - 3], extra32c, 4, hipMemcpyDeviceToHost, c->stream)); }
  }
  HIPCHK(hipStreamSynchronize(c->stream));
  return HUMID_OK;
}

#define TRY(...) do { int _rc = (__VA_ARGS__); if (_rc != HUMID_OK) return _rc; } while (0)

// Plan of the generalised pigeonhole search (see ComboPlan).  s is chosen so that combo keys are
// long enough for buckets to be small at this U (>= ~log4(U) nucleotides) without exceeding
// MAX_COMBOS combinations; d >= n degenerates to one empty-mask combo (every pair compared).
static u64 n_choose_k(u32 n, u32 k) {
  if (k > n) return 0;
  u64 r = 1;
  for (u32 i = 1; i <= k; i++) r = r * (n - k + i) / i;
  return r;
}

// short_later: the keys of the combinations after the first (the ones whose bucket order has to be MADE;
// the first is a prefix of the sorted words) are cut to max(24, 2 * want) bits -- enough to tell U words
// apart, and at <= 24 bits the order comes from the two-level grouping instead of a library sort over
// every key bit (48 bits for two halves of a 48-nt word).  Only the one-GPU Hamming search asks for it:
// the shifted joins of the edit search and the exchange pass's routing keep whole segments.
static ComboPlan make_plan(u32 n, u32 d, u64 U, u32 force_segments, bool short_later = false) {
  ComboPlan p;
  memset(&p, 0, sizeof p);
  // d >= n: every pair is a neighbour pair.  d >= MAX_COMBOS: even the smallest plan, s = d + 1,
  // has d + 1 > MAX_COMBOS combinations (of ONE segment of at most n / (d + 1) <= 3 nucleotides at
  // n <= 64: buckets of a quarter of all words and more), so the search degenerates to the same
  // single combination with an empty mask: one bucket, every pair compared.
  if (d >= n || n_choose_k(d + 1, 1) > MAX_COMBOS) { p.ncombo = 1; p.key_bits = 0; p.mask[0] = W2{0, 0}; p.nfield[0] = 0; return p; }
  u32 want = 1;                                  // nucleotides of key wanted: 4^want >= U
  while (want < n && ((u64)1 << (2 * want)) < U) want++;
  u32 best_s = d + 1, best_len = 0;
  u64 best_c = ~0ull;
  for (u32 sgm = d + 1; sgm <= n && sgm <= d + MAX_FIELDS; sgm++) {
    const u64 combos = n_choose_k(sgm, sgm - d);
    if (combos > MAX_COMBOS) break;
    if (force_segments) {                         // test hook: take exactly this s if it is legal
      if (sgm == force_segments) { best_s = sgm; best_len = (sgm - d) * (n / sgm); best_c = combos; break; }
      continue;
    }
    const u32 len = (sgm - d) * (n / sgm);       // guaranteed key length (short segments)
    const bool better = (best_len < want) ? (len > best_len) : (len >= want && combos < best_c);
    if (best_len == 0 || better) { best_s = sgm; best_len = len; best_c = combos; }
    if (best_len >= want) break;                 // smallest s that reaches the wanted length
  }
  const u32 sgm = best_s, k = sgm - d;
  u32 seg_shift[64], seg_width[64];
  {
    u32 base = n / sgm, rem = n % sgm, pos = 0;
    for (u32 t = 0; t < sgm; t++) {
      u32 len = base + (t < rem ? 1 : 0);
      seg_shift[t] = 2 * (n - pos - len);
      seg_width[t] = 2 * len;
      pos += len;
    }
  }
  // combinations of k segments in lexicographic order: the first is {0..k-1}, a prefix
  u32 idx[64];
  for (u32 t = 0; t < k; t++) idx[t] = t;
  u32 c = 0, maxbits = 0;
  while (true) {
    // A combo key holds at most 64 bits (only wide words can exceed that): the last field is cut
    // to its top bits and later fields are dropped.  Two words within distance d still agree on the
    // shortened mask of some combo, so the search stays complete; it only compares a few more pairs.
    unsigned __int128 m = 0;
    u32 bits = 0, nf = 0;
    const u32 limit = (short_later && c > 0 && !force_segments) ? std::min<u32>(64u, std::max<u32>(24u, 2 * want)) : 64u;
    for (u32 t = 0; t < k && bits < limit; t++) {
      const u32 sg = idx[t];
      u32 wd = seg_width[sg], sh = seg_shift[sg];
      if (bits + wd > limit) { const u32 cut = bits + wd - limit; wd -= cut; sh += cut; }
      p.shift[c][nf] = (u8)sh;
      p.width[c][nf] = (u8)wd;
      m |= ((wd >= 64) ? (unsigned __int128)~0ull : (((unsigned __int128)1 << wd) - 1)) << sh;
      bits += wd;
      nf++;
    }
    p.mask[c] = W2{(u64)(m >> 64), (u64)m};
    p.nfield[c] = (u8)nf;
    if (bits > maxbits) maxbits = bits;
    c++;
    int t = (int)k - 1;
    while (t >= 0 && idx[t] == sgm - k + (u32)t) t--;
    if (t < 0) break;
    idx[t]++;
    for (u32 q = (u32)t + 1; q < k; q++) idx[q] = idx[q - 1] + 1;
    if (c >= MAX_COMBOS) {           // unreachable (C(best_s, k) <= MAX_COMBOS was checked above); never overrun
      memset(&p, 0, sizeof p);
      p.ncombo = 1;
      return p;
    }
  }
  p.ncombo = c;
  p.key_bits = maxbits;
  return p;
}

// ---- cluster stage shared by the full pipeline and the explicit-graph entry point ------
// The arrays a graph lives in: per unique word (legacy view: deg / nbr_off / ... of the context) or per
// COMPACT node (cg_* buffers, kernels_cgraph.hip.h).  n nodes, cnt[n] their counts.
struct GraphArrays {
  u32 *deg, *parent, *csize, *off, *idx, *cl_of, *maxleaf;
  u64 *cl_size;
};
static GraphArrays legacy_arrays(humid_ctx *c) {
  return GraphArrays{c->deg.as<u32>(), c->parent.as<u32>(), c->csize.as<u32>(), c->nbr_off.as<u32>(), c->nbr_idx.as<u32>(),
                     c->cl_of.as<u32>(), c->maxleaf.as<u32>(), c->cl_size.as<u64>()};
}
// needs: cnt[n], deg[n], off[n+1], idx, parent[n] + csize[n] (k_comp_stats done), M = nodes with deg > 0,
// Mbig = those in components larger than SMALL_COMP; small_roots listed by k_comp_count.
// Leaves cl_of (creator + 1) / maxleaf / cl_size (at the creators) in `g`.
// trivial_done: the components of one and two nodes are done and the roots listed (k_cg_trivial)
static int cluster_kernels(humid_ctx *c, const GraphArrays &g, const u32 *g_cnt, u32 U, u64 M, u64 Mbig, u32 method,
                           bool trivial_done = false) {
  hipStream_t st = c->stream;
  if (!trivial_done && c->kev_on) HIPCHK(hipEventRecord(c->kev[2], st));
  if (trivial_done) {
  } else if (method == HUMID_METHOD_MAXIMUM)
    hipLaunchKernelGGL(k_cluster_trivial<true>, dim3(blocks_for(U)), dim3(256), 0, st, g.deg, g.parent, g.csize, U, g_cnt, g.off,
                       g.idx, g.cl_of, g.maxleaf, g.cl_size);
  else
    hipLaunchKernelGGL(k_cluster_trivial<false>, dim3(blocks_for(U)), dim3(256), 0, st, g.deg, g.parent, g.csize, U, g_cnt, g.off,
                       g.idx, g.cl_of, g.maxleaf, g.cl_size);
  if (M > 0) {
    const u64 small_cap = M / 3 + 1;                  // listed roots: components of >= 3 of the M leaves with neighbours
    if (method == HUMID_METHOD_MAXIMUM)
      hipLaunchKernelGGL(k_cluster_small_lds<true>, dim3(blocks_for(small_cap, 64)), dim3(64), 0, st, c->small_roots.as<u32>(),
                         (const ull *)c->d_ctr, g.parent, g.csize, U, g_cnt, g.off, g.idx, g.cl_of, g.maxleaf, g.cl_size);
    else
      hipLaunchKernelGGL(k_cluster_small_lds<false>, dim3(blocks_for(small_cap, 64)), dim3(64), 0, st, c->small_roots.as<u32>(),
                         (const ull *)c->d_ctr, g.parent, g.csize, U, g_cnt, g.off, g.idx, g.cl_of, g.maxleaf, g.cl_size);
    if (Mbig > 0) {
      ENSURE(c->mk0, (size_t)Mbig * 8);
      ENSURE(c->mk1, (size_t)Mbig * 8);
      ENSURE(c->stk, (size_t)Mbig * 8);
      HIPCHK(hipMemsetAsync(&c->d_ctr[CTR_SPECIAL], 0, sizeof(ull), st));
      hipLaunchKernelGGL(k_member_keys, dim3(COMPACT_BLOCKS), dim3(256), 0, st, g.deg, g.parent, g.csize, U, c->mk0.as<u64>(), c->d_ctr);
      TRY(sort_keys<u64>(c, c->mk0.as<u64>(), c->mk1.as<u64>(), Mbig, 0, 32 + bits_for(U)));
      if (method == HUMID_METHOD_MAXIMUM) {
        // maxLeaf ties are broken by depth-first pre-order: one lane per component
        hipLaunchKernelGGL(k_cluster_components<true>, dim3(blocks_for(Mbig, 64)), dim3(64), 0, st,
                           c->mk1.as<u64>(), (u32)Mbig, g_cnt, g.off, g.idx, g.cl_of, g.maxleaf, g.cl_size, c->stk.as<u32>());
      } else if (c->coop_big) {
        // one workgroup per component, flood as a parallel BFS
        ENSURE(c->heads, (size_t)Mbig * 4);
        HIPCHK(hipMemsetAsync(&c->d_ctr[CTR_SPECIAL], 0, sizeof(ull), st));
        hipLaunchKernelGGL(k_comp_heads, dim3(COMPACT_BLOCKS), dim3(256), 0, st, c->mk1.as<u64>(), (u32)Mbig,
                           c->heads.as<u32>(), c->d_ctr);
        const u32 grid = (u32)(Mbig / (SMALL_COMP + 1) + 1 < 2048 ? Mbig / (SMALL_COMP + 1) + 1 : 2048);
        hipLaunchKernelGGL(k_cluster_big_coop, dim3(grid), dim3(256), 0, st, c->mk1.as<u64>(), (u32)Mbig,
                           c->heads.as<u32>(), c->d_ctr, g_cnt, g.off, g.idx, g.cl_of, g.maxleaf, g.cl_size, c->stk.as<u32>());
      } else {
        hipLaunchKernelGGL(k_cluster_components<false>, dim3(blocks_for(Mbig, 64)), dim3(64), 0, st,
                           c->mk1.as<u64>(), (u32)Mbig, g_cnt, g.off, g.idx, g.cl_of, g.maxleaf, g.cl_size, c->stk.as<u32>());
      }
    }
  }
  if (c->kev_on) HIPCHK(hipEventRecord(c->kev[3], st));
  HIPCHK(hipGetLastError());
  return HUMID_OK;
}

// the legacy view: clusters over all U unique words, then creator flags and their prefix sum
static int cluster_stage(humid_ctx *c, const u32 *g_cnt, u32 U, u64 M, u64 Mbig, u32 method) {
  hipStream_t st = c->stream;
  ENSURE(c->cl_of, (size_t)U * 4);
  ENSURE(c->maxleaf, (size_t)U * 4);
  ENSURE(c->cl_size, (size_t)U * 8);
  ENSURE(c->flag, (size_t)U * 4);
  ENSURE(c->pos, (size_t)(U + 1) * 4);
  ENSURE(c->cid, (size_t)U * 4);
  ENSURE(c->ismax, (size_t)U);
  TRY(cluster_kernels(c, legacy_arrays(c), g_cnt, U, M, Mbig, method));
  hipLaunchKernelGGL(k_creator_flags, dim3(blocks_for(U)), dim3(256), 0, st, c->cl_of.as<u32>(), U,
                     c->flag.as<u32>());
  TRY(exscan_u32(c, c->flag.as<u32>(), c->pos.as<u32>(), U));
  HIPCHK(hipGetLastError());
  return HUMID_OK;
}

static int n_clusters_from_scan(humid_ctx *c, u32 U, u64 *out) {
  // (the pass's last host wait: both values through the counters' mapped store)
  TRY(read_counters(c, c->pos.as<u32>() + (U - 1), c->flag.as<u32>() + (U - 1)));
  *out = (c->h_ctr[CTR_N - 1] & 0xffffffffull) + (c->h_ctr[CTR_N - 2] & 0xffffffffull);
  return HUMID_OK;
}

// ---- stage A: exact counts + walk order ------------------------------------------------
// Inserts the reads whose word lies in [range_lo, range_hi] (inclusive; the multi-GPU path
// gives every rank one range, a single GPU takes everything), compacts the table and sorts
// the unique words.  Leaves table/slot_of_read/s_word/s_slot/s_cnt/s_first in the context.
static int stage_count_global(humid_ctx *c, const u64 *d_words, const u8 *d_filt, u32 N, u32 word_nt,
                              u64 range_lo, u64 range_hi, u64 expected_reads, humid_summary &s) {
  hipStream_t st = c->stream;
  c->last_count_lds = false;
  c->last_count_sorted = false;
  c->last_rec8 = false;
  if (expected_reads == 0 || expected_reads > N) expected_reads = N;
  u32 cap_log2 = 10;
  while (((u64)1 << cap_log2) < expected_reads + expected_reads / 2) cap_log2++;
  const u64 cap = (u64)1 << cap_log2;
  c->cap_log2 = cap_log2;
  ENSURE(c->table, (cap + 1) * sizeof(Slot));
  ENSURE(c->slot_out, (cap + 1) * 8);
  ENSURE(c->slot_of_read, (size_t)N * 4);
  ENSURE(c->uniq_slot, (size_t)expected_reads * 4 + 4);
  ENSURE(c->uniq_word, (size_t)expected_reads * 8 + 8);
  HIPCHK(hipEventRecord(c->ev[0], st));
  HIPCHK(hipMemsetAsync(c->d_ctr, 0, CTR_N * sizeof(ull), st));
  HIPCHK(hipMemsetAsync(c->table.p, 0xff, (cap + 1) * sizeof(Slot), st));
  HIPCHK(hipEventRecord(c->kev[0], st));
  hipLaunchKernelGGL(k_hash_insert, dim3(grid_stride_blocks(N)), dim3(256), 0, st, d_words, d_filt, N,
                     c->table.as<Slot>(), cap_log2, c->slot_of_read.as<u32>(), range_lo, range_hi,
                     (u32)(cap - cap / 8), c->d_ctr);
  HIPCHK(hipEventRecord(c->kev[1], st));
  hipLaunchKernelGGL(k_compact_table, dim3(COMPACT_BLOCKS), dim3(256), 0, st,
                     c->table.as<Slot>(), (u32)(cap + 1), c->uniq_word.as<u64>(), c->uniq_slot.as<u32>(),
                     (u32)expected_reads, c->d_ctr);
  HIPCHK(hipGetLastError());
  TRY(read_counters(c));
  if (c->h_ctr[CTR_OVERFULL])
    return fail(c, HUMID_E_INVALID, "hash table over-full: more reads fell into this range than expected_reads");
  const u32 U = (u32)c->h_ctr[CTR_UNIQUE];
  s.usable = c->usable = c->h_ctr[CTR_USABLE];
  s.unique = c->U = U;
  if (U == 0) { if (!c->lean_events) HIPCHK(hipEventRecord(c->ev[1], st)); return HUMID_OK; }
  ENSURE(c->s_word, (size_t)U * 8);
  ENSURE(c->s_slot, (size_t)U * 4);
  ENSURE(c->s_cnt, (size_t)U * 4);
  ENSURE(c->s_first, (size_t)U * 4);
  TRY(sort_pairs<u64, u32>(c, c->uniq_word.as<u64>(), c->s_word.as<u64>(), c->uniq_slot.as<u32>(),
                           c->s_slot.as<u32>(), U, 0, 2 * word_nt));
  hipLaunchKernelGGL(k_post_sort, dim3(blocks_for(U)), dim3(256), 0, st, c->s_slot.as<u32>(),
                     c->table.as<Slot>(), U, c->s_cnt.as<u32>(), c->s_first.as<u32>());
  if (!c->lean_events) HIPCHK(hipEventRecord(c->ev[1], st));
  HIPCHK(hipGetLastError());
  return HUMID_OK;
}

// bucket bits of the partitioned count: 2^pb buckets of about PART_TARGET reads
static inline u32 part_bits(u32 N) {
  static const u64 target = getenv("HUMID_PART_TARGET") ? (u64)std::max(32, atoi(getenv("HUMID_PART_TARGET"))) : (u64)PART_TARGET;   // experiments
  u32 pb = 1;
  while (pb < 26 && (target << pb) < (u64)N) pb++;
  return pb;
}

// Ordered partition key = (word - lo) * scale: the value range the reads lie in, stretched over the
// whole 64-bit key space (see PartKeyOp).  shift < 64 iff scale == 2^shift.
struct KeyMap {
  u64 lo, scale;
  u32 shift;
};
static KeyMap key_map(u32 word_nt, u64 lo, u64 hi, bool within) {
  const u64 top = word_nt >= 32 ? ~0ull : (((u64)1 << (2 * word_nt)) - 1);
  if (!within) { lo = 0; hi = top; }
  if (hi > top) hi = top;
  if (lo > hi) { lo = 0; hi = top; }
  KeyMap m;
  m.lo = lo;
  const u64 span = hi - lo;
  if (span == ~0ull) { m.scale = 1; m.shift = 0; return m; }
  const u64 cnt = span + 1;
  if ((cnt & (cnt - 1)) == 0 && cnt > 1) {
    const u32 k = (u32)__builtin_ctzll(cnt);
    m.shift = 64 - k;
    m.scale = (u64)1 << m.shift;
  } else {
    m.shift = 64;
    m.scale = ~0ull / cnt;
  }
  return m;
}

// Partitioned variant of stage A (see section 1b of the kernels).  Returns HUMID_OK with
// *overflowed = true when a bucket held more unique words than its LDS table (the caller then
// runs the global-table variant; results are never taken from an overflowed run).
// The count stage's two-level tile partition of the reads `src` yields (kernels_part.hip.h) into pk_keys / pk_vals,
// bucket bounds in pbeg.  SRC: ReadsSrc (one-word words) or WideReadsSrc (the heads of two-word words).
template <class SRC>
static int count_partition(humid_ctx *c, const SRC &src, u32 N, u32 pb, bool *used_padded) {
  hipStream_t st = c->stream;
  const u32 n_parts = 1u << pb;
  // hand-written partition (kernels_part.hip.h): two levels of LDS-staged scatter; excluded reads
  // (filtered, or outside this rank's value range) never enter it
  const u32 d1 = (pb + 1) / 2, d2 = pb - d1;
  const u32 nb1 = 1u << d1;
  // pt_work, in u32: [hist1 512 | cursor1 512 | hist_fine n_parts + 1 | cursor2 n_parts] zeroed, then
  // [cbase 513 | tprefix 513]
  const size_t zero_words = 1024 + (size_t)n_parts + 1 + n_parts;
  ENSURE(c->pt_work, (zero_words + 1026) * 4);
  u32 *hist1 = c->pt_work.as<u32>(), *cursor1 = hist1 + 512, *hist_fine = cursor1 + 512,
      *cursor2 = hist_fine + n_parts + 1, *cbase = cursor2 + n_parts, *tprefix = cbase + 513;
  HIPCHK(hipMemsetAsync(c->pt_work.p, 0, zero_words * 4, st));
  const u32 tiles1 = (N + PT_TILE - 1) / PT_TILE, tiles2 = tiles1 + nb1;
  // Two levels and keys that spread evenly over the coarse bins (hashed keys always do, word-ordered
  // keys were only chosen because their prefix does): level 1 scatters into PADDED coarse bins of a
  // fixed room (mean + 25 % + 1024) and needs no histogram pass over the reads in front; the bins'
  // counts are the cursors it leaves behind.  A bin that outgrows its room (heavily duplicated words:
  // all reads of a word share a bin) is reported, the run discarded, and this context goes back to the
  // histogram form (pt_padded = false).
  static const u32 pad_div = getenv("HUMID_PAD_DIV") ? (u32)std::max(1, atoi(getenv("HUMID_PAD_DIV"))) : 4u;   // head room = mean / pad_div
  const bool padded = d2 > 0 && c->pt_padded;
  const u32 cap1 = padded ? (u32)std::min<u64>(0xffffffffull / nb1, (u64)N / nb1 + (u64)N / nb1 / pad_div + 1024) : 0u;
  const size_t room1 = padded ? (size_t)nb1 * cap1 : (size_t)N;
  *used_padded = padded;
  if (padded) {
    ENSURE(c->pad_word, room1 * 8);
    ENSURE(c->pslot, room1 * 4);
  }
  // level-1 output: the final arrays when there is no second level, else scratch that is dead until
  // k_dedup_lds writes it (pad_word, pslot)
  u64 *k1 = d2 ? c->pad_word.as<u64>() : c->pk_keys.as<u64>();
  u32 *v1 = d2 ? c->pslot.as<u32>() : c->pk_vals.as<u32>();
  if (!padded) {
    hipLaunchKernelGGL(k_pt_hist1<SRC>, dim3(tiles1 < 512 ? tiles1 : 512), dim3(1024), 0, st, src, N, d1, hist1);
    hipLaunchKernelGGL(k_pt_scan1, dim3(1), dim3(1024), 0, st, hist1, d1, d2, cbase, tprefix, c->pbeg.as<u32>(),
                       c->ucount.as<u32>() + n_parts, 0u);
  }
  hipLaunchKernelGGL((k_pt_scatter<1, SRC>), dim3(tiles1), dim3(1024), 0, st, src, N, (const u64 *)nullptr,
                     (const u32 *)nullptr, (const u32 *)nullptr, (const u32 *)nullptr, d1, d2, cbase, cursor1, k1, v1,
                     (u32 *)nullptr, cap1, &c->d_ctr[CTR_SPECIAL]);
  if (padded)
    hipLaunchKernelGGL(k_pt_scan1, dim3(1), dim3(1024), 0, st, (const u32 *)cursor1, d1, d2, cbase, tprefix, c->pbeg.as<u32>(),
                       c->ucount.as<u32>() + n_parts, cap1);
  if (c->kev_on) HIPCHK(hipEventRecord(c->kev[39], st));
  if (d2) {
    hipLaunchKernelGGL(k_pt_hist2<SRC>, dim3(tiles2), dim3(1024), 0, st, src, k1, tprefix, cbase, d1, d2, hist_fine, cap1);
    hipLaunchKernelGGL((k_pt_scatter<2, SRC>), dim3(tiles2), dim3(1024), 0, st, src, N, k1, v1, tprefix, cbase, d1, d2,
                       hist_fine, cursor2, c->pk_keys.as<u64>(), c->pk_vals.as<u32>(), c->pbeg.as<u32>(), cap1, &c->d_ctr[CTR_SPECIAL]);
  }
  if (c->kev_on) HIPCHK(hipEventRecord(c->kev[40], st));
  return HUMID_OK;
}

// wide != null (33 <= word_nt <= 64, `ordered` and the tile partition only; d_words unused): buckets are cut
// by the words' heads (WideReadsSrc) and counted by k_dedup_lds_wide (kernels_wide.hip.h).
static int stage_count_lds(humid_ctx *c, const u64 *d_words, const u8 *d_filt, u32 N, u32 word_nt,
                           u64 range_lo, u64 range_hi, const KeyMap &km, bool ordered, humid_summary &s,
                           bool *overflowed, const W2 *wide = nullptr) {
  hipStream_t st = c->stream;
  if (wide && !(ordered && c->use_tile_partition && part_bits(N) <= 18))
    return fail(c, HUMID_E_INVALID, "wide words are counted in word-ordered buckets of the tile partition only");
  const size_t wsize = wide ? sizeof(W2) : 8;
  *overflowed = false;
  c->last_count_lds = true;
  c->last_count_sorted = false;
  c->last_rec8 = false;
  c->last_count_ordered = ordered;
  const u32 pb = part_bits(N);
  const u32 n_parts = 1u << pb;
  c->n_parts = n_parts;
  ENSURE(c->pk_keys, (size_t)N * 8);
  ENSURE(c->pk_vals, (size_t)N * 4);
  ENSURE(c->pbeg, (size_t)(n_parts + 1) * 4);
  ENSURE(c->ucount, (size_t)(n_parts + 1) * 4);
  ENSURE(c->pusable, (size_t)(n_parts + 1) * 4);
  ENSURE(c->ubase, (size_t)(n_parts + 1) * 4);
  // pad_word / pslot double as the output of the padded first partition level (below): sized for that at
  // once, so that they are carved from the context's slab a single time
  const size_t room_early = (size_t)N + (size_t)N / 4 + ((size_t)1024 << ((pb + 1) / 2));
  ENSURE(c->pad_word, std::max(room_early * 8, (size_t)N * wsize));
  ENSURE(c->pad_cf, (size_t)N * 8);
  ENSURE(c->pslot, room_early * 4);
  ENSURE(c->slot_out, ((size_t)N + 1) * 8);
  ENSURE(c->uniq_slot, (size_t)N * 4 + 4);
  ENSURE(c->uniq_word, (size_t)N * 8 + 8);
  HIPCHK(hipEventRecord(c->ev[0], st));
  HIPCHK(hipMemsetAsync(c->d_ctr, 0, CTR_N * sizeof(ull), st));
  const bool check_range = !(range_lo == 0 && range_hi == ~0ull);
  c->last_part_tiled = c->use_tile_partition && pb <= 2 * 9;
  bool used_padded = false;
  if (c->last_part_tiled) {
    PtInput in;
    in.words = d_words; in.filtered = d_filt; in.rlo = range_lo; in.rhi = range_hi;
    in.check_range = check_range ? 1u : 0u;
    in.key = PartKeyOp{ordered ? 1u : 0u, km.lo, km.scale};
    if (wide) TRY(count_partition(c, WideReadsSrc{wide, d_filt, 2 * (word_nt - 32), in.key}, N, pb, &used_padded));
    else TRY(count_partition(c, ReadsSrc{in}, N, pb, &used_padded));
  } else {
    // beyond 2^18 buckets (> ~90 M reads): radix passes over the top pb key bits (prims.hip.h)
    ComposeIn<PartKeyOp, PtrIn<u64>> kin{PartKeyOp{ordered ? 1u : 0u, km.lo, km.scale}, PtrIn<u64>{d_words}};
    ComposeIn<ReadTagOp, IotaIn> vin{ReadTagOp{check_range ? d_words : nullptr, d_filt, range_lo, range_hi}, IotaIn{}};
    TRY((sort_pairs_in<u64, u32>(c, kin, c->pk_keys.as<u64>(), vin, c->pk_vals.as<u32>(), N, 64 - pb, 64)));
    hipLaunchKernelGGL(k_part_bounds, dim3(blocks_for(n_parts + 1)), dim3(256), 0, st, c->pk_keys.as<u64>(), N,
                       pb, n_parts, c->pbeg.as<u32>(), c->ucount.as<u32>());
  }
  HIPCHK(hipEventRecord(c->kev[0], st));
  if (wide) {
    hipLaunchKernelGGL((k_dedup_lds_wide<9, 512, 0, WL_SMALL_LEN>), dim3(n_parts), dim3(256), 0, st, c->pk_keys.as<u64>(), c->pk_vals.as<u32>(),
                       c->pbeg.as<u32>(), wide, 2 * (word_nt - 32), N, pb, c->pad_word.as<W2>(), c->pad_cf.as<uint2>(),
                       c->ucount.as<u32>(), c->pusable.as<u32>(), c->pslot.as<u32>(), c->d_ctr);
    if (N > WL_SMALL_LEN)
      hipLaunchKernelGGL((k_dedup_lds_wide<10, 1024, WL_SMALL_LEN, WL_STAGE>), dim3(n_parts), dim3(256), 0, st, c->pk_keys.as<u64>(), c->pk_vals.as<u32>(),
                         c->pbeg.as<u32>(), wide, 2 * (word_nt - 32), N, pb, c->pad_word.as<W2>(), c->pad_cf.as<uint2>(),
                         c->ucount.as<u32>(), c->pusable.as<u32>(), c->pslot.as<u32>(), c->d_ctr);
  } else if (ordered)
    hipLaunchKernelGGL(k_dedup_lds<true>, dim3(n_parts), dim3(256), 0, st, c->pk_keys.as<u64>(), c->pk_vals.as<u32>(),
                       c->pbeg.as<u32>(), N, pb, km.lo, km.scale, km.shift, c->pad_word.as<u64>(), c->pad_cf.as<uint2>(),
                       c->ucount.as<u32>(), c->pusable.as<u32>(), c->pslot.as<u32>(), c->d_ctr);
  else
    hipLaunchKernelGGL(k_dedup_lds<false>, dim3(n_parts), dim3(256), 0, st, c->pk_keys.as<u64>(), c->pk_vals.as<u32>(),
                       c->pbeg.as<u32>(), N, pb, km.lo, km.scale, km.shift, c->pad_word.as<u64>(), c->pad_cf.as<uint2>(),
                       c->ucount.as<u32>(), c->pusable.as<u32>(), c->pslot.as<u32>(), c->d_ctr);
  HIPCHK(hipEventRecord(c->kev[1], st));
  hipLaunchKernelGGL(k_part_totals, dim3(n_parts >= 16384 ? 64 : 4), dim3(256), 0, st, c->ucount.as<u32>(),
                     c->pusable.as<u32>(), n_parts, c->d_ctr);
  TRY(exscan_u32(c, c->ucount.as<u32>(), c->ubase.as<u32>(), (u64)n_parts + 1));
  HIPCHK(hipGetLastError());
  TRY(read_counters(c));
  if (used_padded && c->h_ctr[CTR_SPECIAL]) {          // a coarse bin outgrew its padded room: once more, with the histogram pass
    c->pt_padded = false;
    return stage_count_lds(c, d_words, d_filt, N, word_nt, range_lo, range_hi, km, ordered, s, overflowed, wide);
  }
  if (c->h_ctr[CTR_OVERFULL]) { *overflowed = true; return HUMID_OK; }
  const u32 U = (u32)c->h_ctr[CTR_UNIQUE];
  s.usable = c->usable = c->h_ctr[CTR_USABLE];
  s.unique = c->U = U;
  if (U == 0) { if (!c->lean_events) HIPCHK(hipEventRecord(c->ev[1], st)); return HUMID_OK; }
  ENSURE(c->s_word, (size_t)(U + 1) * wsize);
  ENSURE(c->s_slot, (size_t)(U + 1) * 4);
  ENSURE(c->s_cnt, (size_t)(U + 1) * 4);
  ENSURE(c->s_first, (size_t)(U + 1) * 4);
  if (wide) {
    hipLaunchKernelGGL(k_compact_padded_wide, dim3(blocks_for((u64)n_parts * 64)), dim3(256), 0, st,
                       c->pad_word.as<W2>(), c->pad_cf.as<uint2>(), c->pbeg.as<u32>(), c->ucount.as<u32>(),
                       c->ubase.as<u32>(), n_parts, c->s_word.as<W2>(), c->s_slot.as<u32>(),
                       c->s_cnt.as<u32>(), c->s_first.as<u32>());
  } else if (ordered) {
    // buckets are runs of the word order and sorted inside: squeezing out the holes IS the sort
    hipLaunchKernelGGL(k_compact_padded<true>, dim3(blocks_for((u64)n_parts * 64)), dim3(256), 0, st,
                       c->pad_word.as<u64>(), c->pad_cf.as<uint2>(), c->pbeg.as<u32>(), c->ucount.as<u32>(),
                       c->ubase.as<u32>(), n_parts, c->s_word.as<u64>(), c->s_slot.as<u32>(),
                       c->s_cnt.as<u32>(), c->s_first.as<u32>());
  } else {
    hipLaunchKernelGGL(k_compact_padded<false>, dim3(blocks_for((u64)n_parts * 64)), dim3(256), 0, st,
                       c->pad_word.as<u64>(), c->pad_cf.as<uint2>(), c->pbeg.as<u32>(), c->ucount.as<u32>(),
                       c->ubase.as<u32>(), n_parts, c->uniq_word.as<u64>(), c->uniq_slot.as<u32>(),
                       (u32 *)nullptr, (u32 *)nullptr);
    TRY(sort_pairs<u64, u32>(c, c->uniq_word.as<u64>(), c->s_word.as<u64>(), c->uniq_slot.as<u32>(),
                             c->s_slot.as<u32>(), U, 0, 2 * word_nt));
    hipLaunchKernelGGL(k_post_sort_padded, dim3(blocks_for(U)), dim3(256), 0, st, c->s_slot.as<u32>(),
                       c->pad_cf.as<uint2>(), U, c->s_cnt.as<u32>(), c->s_first.as<u32>());
  }
  if (!c->lean_events) HIPCHK(hipEventRecord(c->ev[1], st));
  HIPCHK(hipGetLastError());
  return HUMID_OK;
}

// Stage A on 8-byte records (kernels_part8.hip.h): word-ordered buckets only, both partition levels padded.
// *done = false: not this shape (the record would not fit 64 bits, too few / too many buckets, the read
// set too large for the tiled un-permute) or a bin outgrew its room -- the caller takes stage_count_lds.
static int stage_count_rec(humid_ctx *c, const u64 *d_words, const u8 *d_filt, u32 N, u32 word_nt, u64 range_lo, u64 range_hi,
                           const KeyMap &km, humid_summary &s, bool *done) {
  hipStream_t st = c->stream;
  *done = false;
  if (!c->use_rec8 || !c->use_tile_partition || !c->pt_padded) return HUMID_OK;
  const u32 pb = part_bits(N);
  if (pb < 6 || pb > 18) return HUMID_OK;
  if ((((u64)N + (1u << UW_MAXSHIFT) - 1) >> UW_MAXSHIFT) > UW_MAXBINS) return HUMID_OK;
  const u32 d1 = (pb + 1) / 2, d2 = pb - d1, nb1 = 1u << d1, n_parts = 1u << pb;
  RecKey rk;
  rk.lo = km.lo; rk.scale = km.scale;
  rk.pow2 = km.shift < 64 ? 1u : 0u;
  rk.z = rk.pow2 ? km.shift : 63u - (u32)__builtin_clzll(km.scale);
  rk.kbits = 64 - rk.z;
  const u32 ibits = bits_for(N);
  if (rk.kbits < pb + 1 || rk.kbits - d1 + ibits > 64) return HUMID_OK;
  static const u32 pad_div = getenv("HUMID_PAD_DIV") ? (u32)std::max(1, atoi(getenv("HUMID_PAD_DIV"))) : 4u;
  const u32 cap1 = (u32)std::min<u64>(0xffffffffull / nb1, (u64)N / nb1 + (u64)N / nb1 / pad_div + 1024);
  const size_t room1 = (size_t)nb1 * cap1, room2 = (size_t)n_parts << P8_CAP2_LOG;
  ENSURE(c->p8_a, room1 * 8);
  ENSURE(c->p8_b, room2 * 8);
  ENSURE(c->pad_word, room2 * 8);
  ENSURE(c->pad_cf, room2 * 8);
  ENSURE(c->slot_out, (room2 + 1) * 8);
  ENSURE(c->pbeg, (size_t)(n_parts + 1) * 4);
  ENSURE(c->ucount, (size_t)(n_parts + 1) * 4);
  // per bucket (reads << 32 | unique words) and its exclusive scan; entry n_parts = the scan's sentinel -> the totals
  ENSURE(c->p8_status, ((size_t)n_parts + 1) * 16);
  u64 *agg = c->p8_status.as<u64>(), *abase = agg + n_parts + 1;
  // p8_cur, in u32: [cursor1 512 | cursor2 n_parts] zeroed, then [cbase 513 | tprefix 513].  (Not pt_work: the
  // reads per bucket, cursor2, are read again by the un-permute at the end of the pass, and the graph
  // stage's grouping uses pt_work in between.)
  ENSURE(c->p8_cur, ((size_t)512 + n_parts + 1026) * 4);
  u32 *cursor1 = c->p8_cur.as<u32>(), *cursor2 = cursor1 + 512, *cbase = cursor2 + n_parts, *tprefix = cbase + 513;
  HIPCHK(hipEventRecord(c->ev[0], st));
  {
    ZeroList z;
    memset(&z, 0, sizeof z);
    z.p[0] = cursor1; z.n[0] = 512 + n_parts;
    z.p[1] = (u32 *)c->d_ctr; z.n[1] = 2 * CTR_N;
    z.p[2] = (u32 *)(agg + n_parts); z.n[2] = 2;
    hipLaunchKernelGGL(k_zero_many, dim3(32), dim3(256), 0, st, z);
  }
  const bool check_range = !(range_lo == 0 && range_hi == ~0ull);
  const Reads8 src{d_words, d_filt, range_lo, range_hi, check_range ? 1u : 0u, rk};
  const u32 tiles1 = (N + PT_TILE - 1) / PT_TILE, tiles2 = tiles1 + nb1;
  static const bool s1_small = getenv("HUMID_S1_THREADS") ? atoi(getenv("HUMID_S1_THREADS")) == 512 : false;  // (experiments: 512 is 20 us slower)
  if (s1_small)
    hipLaunchKernelGGL((k_p8_scatter1<Reads8, 512>), dim3((N + 4095) / 4096), dim3(512), 0, st, src, N, rk.kbits, d1, ibits, cap1, cursor1,
                       c->p8_a.as<u64>(), c->d_ctr);
  else
    hipLaunchKernelGGL((k_p8_scatter1<Reads8, 1024>), dim3(tiles1), dim3(1024), 0, st, src, N, rk.kbits, d1, ibits, cap1, cursor1,
                       c->p8_a.as<u64>(), c->d_ctr);
  hipLaunchKernelGGL(k_pt_scan1, dim3(1), dim3(1024), 0, st, (const u32 *)cursor1, d1, d2, cbase, tprefix, c->pbeg.as<u32>(),
                     c->ucount.as<u32>() + n_parts, cap1);
  if (c->kev_on) HIPCHK(hipEventRecord(c->kev[39], st));
  hipLaunchKernelGGL(k_p8_scatter2, dim3(tiles2), dim3(1024), 0, st, (const u64 *)c->p8_a.as<u64>(), (const u32 *)tprefix,
                     (const u32 *)cbase, rk.kbits, d1, d2, ibits, cap1, cursor2, c->p8_b.as<u64>(), c->d_ctr);
  if (c->kev_on) HIPCHK(hipEventRecord(c->kev[40], st));
  HIPCHK(hipEventRecord(c->kev[0], st));
  hipLaunchKernelGGL(k_dedup_rec, dim3(n_parts), dim3(256), 0, st, c->p8_b.as<u64>(), (const u32 *)cursor2, N, pb, d1, ibits, rk,
                     c->pad_word.as<u64>(), c->pad_cf.as<uint2>(), agg, c->d_ctr);
  HIPCHK(hipEventRecord(c->kev[1], st));
  TRY(exscan_in<u64>(c, PtrIn<u64>{agg}, abase, (u64)n_parts + 1));
  // the walk-order arrays are squeezed out of the padded ones BEFORE the host knows how many unique words there are
  // (at most N: a bucket never reports more words than records it holds): the host's wait for the counters -- it
  // needs U to shape the graph stage -- then runs beside this kernel instead of an idle GPU
  ENSURE(c->s_word, (size_t)(N + 1) * 8);
  ENSURE(c->s_slot, (size_t)(N + 1) * 4);
  ENSURE(c->s_cnt, (size_t)(N + 1) * 4);
  ENSURE(c->s_first, (size_t)(N + 1) * 4);
  hipLaunchKernelGGL(k_compact_padded8, dim3(blocks_for((u64)n_parts * 64)), dim3(256), 0, st, c->pad_word.as<u64>(),
                     c->pad_cf.as<uint2>(), (const u64 *)agg, (const u64 *)abase, n_parts, c->s_word.as<u64>(),
                     c->s_slot.as<u32>(), c->s_cnt.as<u32>(), c->s_first.as<u32>());
  if (!c->lean_events) HIPCHK(hipEventRecord(c->ev[1], st));
  HIPCHK(hipGetLastError());
  TRY(read_counters(c, (const u32 *)(abase + n_parts), (const u32 *)(abase + n_parts) + 1));   // U, usable
  if (getenv("HUMID_TRACE_COUNT"))
    fprintf(stderr, "[rec count] N %u pb %u kbits %u ibits %u cap1 %u special %llu overfull %llu unique %llu usable %llu\n", N, pb, rk.kbits,
            ibits, cap1, (ull)c->h_ctr[CTR_SPECIAL], (ull)c->h_ctr[CTR_OVERFULL], (ull)(c->h_ctr[CTR_N - 1] & 0xffffffffull), (ull)(c->h_ctr[CTR_N - 2] & 0xffffffffull));
  if (c->h_ctr[CTR_SPECIAL]) { c->pt_padded = false; return HUMID_OK; }     // a bin outgrew its room: the exact kernels from now on
  if (c->h_ctr[CTR_OVERFULL]) return HUMID_OK;
  c->last_count_lds = true;
  c->last_count_sorted = false;
  c->last_count_ordered = true;
  c->last_part_tiled = true;
  c->last_rec8 = true;
  c->rec_cursor2 = cursor2;
  c->n_parts = n_parts;
  const u32 U = (u32)(c->h_ctr[CTR_N - 1] & 0xffffffffull);
  s.usable = c->usable = c->h_ctr[CTR_N - 2] & 0xffffffffull;
  s.unique = c->U = U;
  *done = true;
  return HUMID_OK;
}

// Stage A for two-word words on records (kernels_part8.hip.h, second half): the 16-byte word + its read index travel
// through both padded partition levels, k_dedup_wide_rec reads its bucket as two contiguous streams.  *done = false:
// not this shape, or a bin outgrew its room -- the caller takes stage_count_lds (keys + gather) or the sort.
static int stage_count_rec_wide(humid_ctx *c, const W2 *d_words, const u8 *d_filt, u32 N, u32 word_nt, const KeyMap &km,
                                humid_summary &s, bool *done) {
  hipStream_t st = c->stream;
  *done = false;
  if (!c->use_rec8 || !c->use_tile_partition || !c->pt_padded) return HUMID_OK;
  const u32 pb = part_bits(N);
  if (pb < 6 || pb > 18) return HUMID_OK;
  if ((((u64)N + (1u << UW_MAXSHIFT) - 1) >> UW_MAXSHIFT) > UW_MAXBINS) return HUMID_OK;
  const u32 d1 = (pb + 1) / 2, d2 = pb - d1, nb1 = 1u << d1, n_parts = 1u << pb;
  RecKey rk;
  rk.lo = km.lo; rk.scale = km.scale;
  rk.pow2 = km.shift < 64 ? 1u : 0u;
  rk.z = rk.pow2 ? km.shift : 63u - (u32)__builtin_clzll(km.scale);
  rk.kbits = 64 - rk.z;
  if (rk.kbits < pb + 1) return HUMID_OK;
  const u32 hbits = 2 * (word_nt - 32);
  static const u32 pad_div = getenv("HUMID_PAD_DIV") ? (u32)std::max(1, atoi(getenv("HUMID_PAD_DIV"))) : 4u;
  const u32 cap1 = (u32)std::min<u64>(0xffffffffull / nb1, (u64)N / nb1 + (u64)N / nb1 / pad_div + 1024);
  const size_t room1 = (size_t)nb1 * cap1, room2 = (size_t)n_parts << P8_CAP2_LOG;
  ENSURE(c->pw_a, room1 * 16);
  ENSURE(c->pw_ai, room1 * 4);
  ENSURE(c->pw_b, room2 * 16);
  ENSURE(c->pw_bi, room2 * 4);
  ENSURE(c->p8_b, room2 * 8);
  ENSURE(c->pad_word, room2 * 16);
  ENSURE(c->pad_cf, room2 * 8);
  ENSURE(c->slot_out, (room2 + 1) * 8);
  ENSURE(c->pbeg, (size_t)(n_parts + 1) * 4);
  ENSURE(c->ucount, (size_t)(n_parts + 1) * 4);
  ENSURE(c->p8_status, ((size_t)n_parts + 1) * 16);
  u64 *agg = c->p8_status.as<u64>(), *abase = agg + n_parts + 1;
  ENSURE(c->p8_cur, ((size_t)512 + n_parts + 1026) * 4);
  u32 *cursor1 = c->p8_cur.as<u32>(), *cursor2 = cursor1 + 512, *cbase = cursor2 + n_parts, *tprefix = cbase + 513;
  HIPCHK(hipEventRecord(c->ev[0], st));
  {
    ZeroList z;
    memset(&z, 0, sizeof z);
    z.p[0] = cursor1; z.n[0] = 512 + n_parts;
    z.p[1] = (u32 *)c->d_ctr; z.n[1] = 2 * CTR_N;
    z.p[2] = (u32 *)(agg + n_parts); z.n[2] = 2;
    hipLaunchKernelGGL(k_zero_many, dim3(32), dim3(256), 0, st, z);
  }
  const u32 tiles1 = (N + PT_TILE - 1) / PT_TILE, tiles2 = tiles1 + nb1;
  hipLaunchKernelGGL(k_pw_scatter<1>, dim3(tiles1), dim3(1024), 0, st, d_words, d_filt, (const u32 *)nullptr, N, hbits, rk,
                     (const u32 *)nullptr, (const u32 *)nullptr, d1, d2, cap1, cursor1, c->pw_a.as<W2>(), c->pw_ai.as<u32>(), c->d_ctr);
  hipLaunchKernelGGL(k_pt_scan1, dim3(1), dim3(1024), 0, st, (const u32 *)cursor1, d1, d2, cbase, tprefix, c->pbeg.as<u32>(),
                     c->ucount.as<u32>() + n_parts, cap1);
  if (c->kev_on) HIPCHK(hipEventRecord(c->kev[39], st));
  hipLaunchKernelGGL(k_pw_scatter<2>, dim3(tiles2), dim3(1024), 0, st, (const W2 *)c->pw_a.as<W2>(), (const u8 *)nullptr,
                     (const u32 *)c->pw_ai.as<u32>(), N, hbits, rk, (const u32 *)tprefix, (const u32 *)cbase, d1, d2, cap1, cursor2,
                     c->pw_b.as<W2>(), c->pw_bi.as<u32>(), c->d_ctr);
  if (c->kev_on) HIPCHK(hipEventRecord(c->kev[40], st));
  HIPCHK(hipEventRecord(c->kev[0], st));
  hipLaunchKernelGGL((k_dedup_wide_rec<9, 512, 0, WL_SMALL_LEN>), dim3(n_parts), dim3(256), 0, st, (const W2 *)c->pw_b.as<W2>(),
                     (const u32 *)c->pw_bi.as<u32>(), (const u32 *)cursor2, hbits, rk, N, pb, c->pad_word.as<W2>(), c->pad_cf.as<uint2>(),
                     agg, c->p8_b.as<u64>(), c->d_ctr);
  if (N > WL_SMALL_LEN)
    hipLaunchKernelGGL((k_dedup_wide_rec<10, 1024, WL_SMALL_LEN, WL_STAGE>), dim3(n_parts), dim3(256), 0, st,
                       (const W2 *)c->pw_b.as<W2>(), (const u32 *)c->pw_bi.as<u32>(), (const u32 *)cursor2, hbits, rk, N, pb,
                       c->pad_word.as<W2>(), c->pad_cf.as<uint2>(), agg, c->p8_b.as<u64>(), c->d_ctr);
  HIPCHK(hipEventRecord(c->kev[1], st));
  TRY(exscan_in<u64>(c, PtrIn<u64>{agg}, abase, (u64)n_parts + 1));
  // (as in stage_count_rec: squeezed out beside the host's wait for the counters; at most N unique words)
  ENSURE(c->s_word, (size_t)(N + 1) * 16);
  ENSURE(c->s_slot, (size_t)(N + 1) * 4);
  ENSURE(c->s_cnt, (size_t)(N + 1) * 4);
  ENSURE(c->s_first, (size_t)(N + 1) * 4);
  hipLaunchKernelGGL(k_compact_padded8_wide, dim3(blocks_for((u64)n_parts * 64)), dim3(256), 0, st, (const W2 *)c->pad_word.as<W2>(),
                     (const uint2 *)c->pad_cf.as<uint2>(), (const u64 *)agg, (const u64 *)abase, n_parts, c->s_word.as<W2>(),
                     c->s_slot.as<u32>(), c->s_cnt.as<u32>(), c->s_first.as<u32>());
  if (!c->lean_events) HIPCHK(hipEventRecord(c->ev[1], st));
  HIPCHK(hipGetLastError());
  TRY(read_counters(c, (const u32 *)(abase + n_parts), (const u32 *)(abase + n_parts) + 1));   // U, usable
  if (getenv("HUMID_TRACE_COUNT"))
    fprintf(stderr, "[rec count, wide] N %u pb %u kbits %u cap1 %u special %llu overfull %llu unique %llu usable %llu\n", N, pb, rk.kbits, cap1,
            (ull)c->h_ctr[CTR_SPECIAL], (ull)c->h_ctr[CTR_OVERFULL], (ull)(c->h_ctr[CTR_N - 1] & 0xffffffffull), (ull)(c->h_ctr[CTR_N - 2] & 0xffffffffull));
  if (c->h_ctr[CTR_SPECIAL] || c->h_ctr[CTR_OVERFULL]) return HUMID_OK;       // (the key + gather road decides by itself what to do next)
  c->last_count_lds = true;
  c->last_count_sorted = false;
  c->last_count_ordered = true;
  c->last_part_tiled = true;
  c->last_rec8 = true;
  c->rec_cursor2 = cursor2;
  c->n_parts = n_parts;
  const u32 U = (u32)(c->h_ctr[CTR_N - 1] & 0xffffffffull);
  s.usable = c->usable = c->h_ctr[CTR_N - 2] & 0xffffffffull;
  s.unique = c->U = U;
  *done = true;
  return HUMID_OK;
}

// Would word-ordered buckets fit their LDS tables?  Histogram of the top (up to 12) word bits over
// a sample of the reads, folded / scaled to the 2^pb buckets the partition will use: the fullest
// bucket, with a 1.5x margin, must stay below the table's fill limit (a bucket's unique words
// cannot exceed its reads).  UMI-first layouts pass; read-prefix-first amplicon or low-complexity
// data does not and keeps the hashed buckets.  A wrong "yes" only costs the overflow fallback.
static int prefix_fits_ordered(humid_ctx *c, const u64 *d_words, const u8 *d_filt, u32 N, u32 word_nt,
                               const KeyMap &km, bool *fits) {
  *fits = false;
  // the answer for this shape is remembered: the sample and its host wait are paid once, not per
  // pass (a wrong "yes" on other data of the same shape costs the overflow fallback and resets it)
  if (c->oc_valid && c->oc_n == N && c->oc_nt == word_nt && c->oc_lo == km.lo && c->oc_scale == km.scale) {
    *fits = c->oc_fits;
    return HUMID_OK;
  }
  const u32 bits = 2 * word_nt < 12 ? 2 * word_nt : 12;
  const u32 n_bins = 1u << bits;
  if (N < 65536) return HUMID_OK;                              // small inputs: not worth a decision
  // a sample is enough: the first 512 K reads (FastQ order is unrelated to the word value)
  const u32 n_sample = N < (1u << 19) ? N : (1u << 19);
  ENSURE(c->small, (size_t)n_bins * 4);
  HIPCHK(hipMemsetAsync(c->small.p, 0, (size_t)n_bins * 4, c->stream));
  hipLaunchKernelGGL(k_top_hist, dim3(128), dim3(1024), n_bins * 4, c->stream, d_words, d_filt, n_sample,
                     km.lo, km.scale, bits, c->small.as<u32>());
  std::vector<u32> h(n_bins);
  HIPCHK(hipMemcpyAsync(h.data(), c->small.p, (size_t)n_bins * 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(c->stream));
  const u32 pb = part_bits(N);
  double worst = 0;
  if (pb >= bits) {                    // several buckets per bin: assume the bin splits evenly
    u32 mx = 0;
    for (u32 v : h) if (v > mx) mx = v;
    worst = (double)mx / (double)(1u << (pb - bits));
  } else {                             // several bins per bucket: fold
    const u32 per = 1u << (bits - pb);
    for (u32 b = 0; b < n_bins; b += per) {
      u64 t = 0;
      for (u32 k = 0; k < per; k++) t += h[b + k];
      if ((double)t > worst) worst = (double)t;
    }
  }
  worst *= (double)N / (double)n_sample;
  *fits = worst * 1.5 <= (double)LDS_FILL_LIMIT;
  c->oc_valid = true; c->oc_fits = *fits;
  c->oc_n = N; c->oc_nt = word_nt; c->oc_lo = km.lo; c->oc_scale = km.scale;
  return HUMID_OK;
}

// stage A dispatcher: partitioned LDS tables when every usable read is counted here (one GPU; a
// multi-GPU rank in exchange mode, `within`: all reads lie in [range_lo, range_hi]), the global
// table for a partial range of a larger array or after a bucket overflow.
static int stage_count(humid_ctx *c, const u64 *d_words, const u8 *d_filt, u32 N, u32 word_nt,
                       u64 range_lo, u64 range_hi, u64 expected_reads, humid_summary &s, bool within = false) {
  const bool full_range = (range_lo == 0 && range_hi == ~0ull);
  if (c->count_mode == 0 && (full_range || within)) {
    const KeyMap km = key_map(word_nt, range_lo, range_hi, within);
    bool overflowed = false;
    bool ordered = c->count_order == 1;
    if (c->count_order < 0) TRY(prefix_fits_ordered(c, d_words, d_filt, N, word_nt, km, &ordered));
    c->last_rec8 = false;
    if (ordered) {
      bool done = false;
      TRY(stage_count_rec(c, d_words, d_filt, N, word_nt, range_lo, range_hi, km, s, &done));
      if (done) return HUMID_OK;
      TRY(stage_count_lds(c, d_words, d_filt, N, word_nt, range_lo, range_hi, km, true, s, &overflowed));
      if (!overflowed) return HUMID_OK;
      // these words do not fit word-ordered buckets after all: remember that for this shape
      c->oc_valid = true; c->oc_fits = false;
      c->oc_n = N; c->oc_nt = word_nt; c->oc_lo = km.lo; c->oc_scale = km.scale;
    }
    TRY(stage_count_lds(c, d_words, d_filt, N, word_nt, range_lo, range_hi, km, false, s, &overflowed));
    if (!overflowed) return HUMID_OK;
  }
  return stage_count_global(c, d_words, d_filt, N, word_nt, range_lo, range_hi, expected_reads, s);
}

// Stage A for wide words (two uint64 per read): counts by sorting, see kernels_wide.hip.h.
// Leaves s_word (W2)/s_cnt/s_first/s_slot and, for stage C, the partition-order arrays
// pk_vals/pslot in the context.
// head_lo / head_hi (within): every word's head lies in that range (a rank's value range in the exchange pass).
static int stage_count_wide(humid_ctx *c, const W2 *d_words, const u8 *d_filt, u32 N, u32 word_nt, humid_summary &s,
                            u64 head_lo = 0, u64 head_hi = ~0ull, bool within = false) {
  hipStream_t st = c->stream;
  // LDS tables over head-ordered buckets when the heads spread evenly (count_mode 0, as for one-word
  // words; count_order 0 keeps the sort); the sort below otherwise and after an overflow
  if (c->count_mode == 0 && c->count_order != 0 && c->use_tile_partition && (N >= 65536 || c->count_order == 1) &&
      part_bits(N) <= 18) {
    const KeyMap km = key_map(24, head_lo >> WIDE_KEY_DROP, head_hi >> WIDE_KEY_DROP, within);   // (the keys: 48-bit numbers, see WideReadsSrc)
    bool ordered = c->count_order == 1;
    if (c->count_order < 0) {
      // the decision samples the first 512 K reads (and is remembered for the shape): heads of those only;
      // the partition itself computes a word's head as it reads the word (WideReadsSrc)
      const u32 n_sample = N < (1u << 19) ? N : (1u << 19);
      ENSURE(c->w_heads, (size_t)n_sample * 8);
      hipLaunchKernelGGL(k_wide_head64, dim3(blocks_for(n_sample)), dim3(256), 0, st, d_words, n_sample, 2 * (word_nt - 32), c->w_heads.as<u64>(),
                         WIDE_KEY_DROP);
      TRY(prefix_fits_ordered(c, c->w_heads.as<u64>(), d_filt, N, word_nt, km, &ordered));
    }
    if (getenv("HUMID_TRACE_COUNT")) fprintf(stderr, "[wide count] N %u order %d fits %d lo %llx scale %llx shift %u\n", N, c->count_order, (int)ordered, (ull)km.lo, (ull)km.scale, km.shift);
    c->last_rec8 = false;
    if (ordered) {
      bool done8 = false;
      TRY(stage_count_rec_wide(c, d_words, d_filt, N, word_nt, km, s, &done8));
      if (done8) return HUMID_OK;
      bool overflowed = false;
      TRY(stage_count_lds(c, nullptr, d_filt, N, word_nt, 0ull, ~0ull, km, true, s, &overflowed, d_words));
      if (getenv("HUMID_TRACE_COUNT")) fprintf(stderr, "[wide count] overflowed %d special %llu overfull %llu\n", (int)overflowed, (ull)c->h_ctr[CTR_SPECIAL], (ull)c->h_ctr[CTR_OVERFULL]);
      if (!overflowed) return HUMID_OK;
      c->oc_valid = true; c->oc_fits = false;
      c->oc_n = N; c->oc_nt = word_nt; c->oc_lo = km.lo; c->oc_scale = km.scale;
    }
  }
  c->last_count_lds = true;          // stage C walks pk_vals/pslot (k_read_map_part)
  c->last_count_ordered = false;
  c->last_count_sorted = true;
  c->last_rec8 = false;
  c->n_parts = 0;
  const u32 hbits = 2 * (word_nt - 32);
  const u32 grid = grid_stride_blocks(N);
  ENSURE(c->pk_keys, (size_t)N * 8);
  ENSURE(c->pad_word, (size_t)N * 8);
  ENSURE(c->pk_vals, (size_t)N * 4);
  ENSURE(c->uniq_slot, (size_t)N * 4 + 4);
  ENSURE(c->pslot, (size_t)N * 4);
  ENSURE(c->w_sorted, (size_t)N * sizeof(W2));
  ENSURE(c->w_head, ((size_t)N + 1) * 4);
  ENSURE(c->w_hpos, ((size_t)N + 1) * 4);
  HIPCHK(hipEventRecord(c->ev[0], st));
  HIPCHK(hipMemsetAsync(c->d_ctr, 0, CTR_N * sizeof(ull), st));
  HIPCHK(hipEventRecord(c->kev[0], st));
  u64 *k0 = c->pk_keys.as<u64>(), *k1 = c->pad_word.as<u64>();
  u32 *va = c->uniq_slot.as<u32>(), *vb = c->pk_vals.as<u32>();
  hipLaunchKernelGGL(k_wide_keys_lo, dim3(grid), dim3(256), 0, st, d_words, d_filt, N, k0, va, c->d_ctr);
  TRY(sort_pairs<u64, u32>(c, k0, k1, va, vb, N, 0, 64));                       // by lo
  hipLaunchKernelGGL(k_wide_keys_hi, dim3(grid), dim3(256), 0, st, d_words, d_filt, vb, N, hbits, k0);
  TRY(sort_pairs<u64, u32>(c, k0, k1, vb, va, N, 0, hbits < 64 ? hbits + 1 : 64));   // by (filtered,) hi
  u32 *v = va;
  if (hbits == 64) {                                                            // n = 64: no spare key bit
    hipLaunchKernelGGL(k_wide_keys_flag, dim3(grid), dim3(256), 0, st, d_filt, va, N, (u32 *)k0);
    TRY(sort_pairs<u32, u32>(c, (u32 *)k0, (u32 *)k1, va, vb, N, 0, 1));
    v = vb;
  }
  HIPCHK(hipEventRecord(c->kev[1], st));
  hipLaunchKernelGGL(k_wide_gather, dim3(grid), dim3(256), 0, st, d_words, v, N, hbits, c->w_sorted.as<W2>());
  hipLaunchKernelGGL(k_wide_heads, dim3(grid), dim3(256), 0, st, c->w_sorted.as<W2>(), N, c->d_ctr,
                     c->w_head.as<u32>());
  TRY(exscan_u32(c, c->w_head.as<u32>(), c->w_hpos.as<u32>(), (u64)N + 1));
  HIPCHK(hipGetLastError());
  TRY(read_counters(c, c->w_hpos.as<u32>() + N));                               // h_ctr[CTR_N-1] = U
  const u32 U = (u32)(c->h_ctr[CTR_N - 1] & 0xffffffffull);
  s.usable = c->usable = c->h_ctr[CTR_USABLE];
  s.unique = c->U = U;
  ENSURE(c->s_word, (size_t)(U + 1) * sizeof(W2));
  ENSURE(c->s_slot, (size_t)(U + 1) * 4);
  ENSURE(c->s_cnt, (size_t)(U + 1) * 4);
  ENSURE(c->s_first, (size_t)(U + 1) * 4);
  ENSURE(c->w_start, (size_t)(U + 2) * 4);
  ENSURE(c->slot_out, (size_t)(U + 1) * 8);
  hipLaunchKernelGGL(k_wide_unique, dim3(grid), dim3(256), 0, st, c->w_sorted.as<W2>(), v, c->w_head.as<u32>(),
                     c->w_hpos.as<u32>(), N, c->d_ctr, c->s_word.as<W2>(), c->s_first.as<u32>(),
                     c->w_start.as<u32>(), c->pslot.as<u32>(), c->pk_vals.as<u32>());
  if (U)
    hipLaunchKernelGGL(k_wide_counts, dim3(blocks_for(U)), dim3(256), 0, st, c->w_start.as<u32>(), U,
                       c->s_cnt.as<u32>(), c->s_slot.as<u32>());
  if (!c->lean_events) HIPCHK(hipEventRecord(c->ev[1], st));
  HIPCHK(hipGetLastError());
  return HUMID_OK;
}

// buckets longer than k_pairs' bounded walk, over the walked order W[0, n) of one combination: device list at
// c->big_runs + slot * cap (start, length, first tile), host copy in `runs` with the total as a last entry
template <class WT>
// cap_n (0: n): the array length the slots of the device list are sized by -- one value for all combinations of a
// caller that keeps several lists at once
static int find_big_runs(humid_ctx *c, const WT *W, u32 n, WT mask, u32 walk_max, u32 slot, std::vector<BigRun> &runs,
                         const BigRun **d_runs_out, u32 cap_n = 0) {
  hipStream_t st = c->stream;
  const u32 cap = (cap_n ? cap_n : n) / (walk_max + 2) + 1;       // runs are disjoint and longer than walk_max + 1
  ENSURE(c->big_runs, (size_t)MAX_COMBOS * cap * sizeof(BigRun) + 16);
  u32 *d_n = (u32 *)((char *)c->big_runs.p + (size_t)MAX_COMBOS * cap * sizeof(BigRun));
  BigRun *d_runs = c->big_runs.as<BigRun>() + (size_t)slot * cap;
  HIPCHK(hipMemsetAsync(d_n, 0, 4, st));
  hipLaunchKernelGGL(k_big_runs<WT>, dim3(blocks_for(n)), dim3(256), 0, st, W, n, mask, walk_max, d_runs, cap, d_n);
  u32 n_runs = 0;
  HIPCHK(hipMemcpyAsync(&n_runs, d_n, 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  if (n_runs > cap) return fail(c, HUMID_E_INVALID, "more large buckets (%u) than fit the input (%u)", n_runs, cap);
  runs.resize(n_runs);
  if (n_runs) HIPCHK(hipMemcpy(runs.data(), d_runs, (size_t)n_runs * sizeof(BigRun), hipMemcpyDeviceToHost));
  std::sort(runs.begin(), runs.end(), [](const BigRun &x, const BigRun &y) { return x.start < y.start; });
  ull tiles = 0;
  for (BigRun &x : runs) {
    const ull nt = ((ull)x.len + PT2_TILE - 1) / PT2_TILE;
    x.tile0 = tiles;
    tiles += nt * (nt + 1) / 2;
  }
  runs.push_back(BigRun{0u, 0u, tiles});                          // sentinel: the total
  if (n_runs) HIPCHK(hipMemcpy(d_runs, runs.data(), (size_t)n_runs * sizeof(BigRun), hipMemcpyHostToDevice));
  *d_runs_out = d_runs;
  return HUMID_OK;
}

// Keys of at most 24 bits (any combination of one-word words): bucket order by GROUPING in two hand-written levels
// (kernels_part.hip.h: the tile partition by the top d1 <= 9 key bits, k_group_fine by the rest) instead
// of a library sort; only that equal keys end up next to each other matters.  *done = false: not this
// shape (the caller sorts).
static ComboFields plan_fields(const ComboPlan &plan, u32 cb);
template <class SRC, class WT>
static int group_words_by_stretch(humid_ctx *c, const ComboPlan &plan, u32 cb, const WT *W, u32 n, u64 *ws, u32 *vs, bool *done,
                                  bool may_pad = false) {
  hipStream_t st = c->stream;
  u32 bit_n = 0;
  for (u32 f = 0; f < plan.nfield[cb]; f++) bit_n += plan.width[cb][f];
  *done = c->group_buckets && plan.nfield[cb] >= 1 && bit_n >= 2 && bit_n <= 24 && n >= 4096;
  c->gf_valid = nullptr;
  if (!*done) return HUMID_OK;
  const u32 d1 = bit_n >= 18 ? 9u : (bit_n + 1) / 2, d2 = bit_n - d1;          // d2 <= 15: 2^15 LDS counters at most
  const u32 nb1 = 1u << d1;
  // may_pad (the caller reads CTR_GOVER at its host wait and comes back without it when a bin was full): level 1
  // scatters into PADDED coarse bins (mean + 25 % + 1024, as the count stage's first level) -- no histogram pass
  // over the words in front, the bins' counts are the cursors left behind; and no bin can hold more than its
  // room, so the launches for bin sizes beyond it are left out
  const bool padded = may_pad && c->gf_padded;
  const u32 cap1 = padded ? n / nb1 + n / nb1 / 4 + 1024 : 0u;
  const size_t room = padded ? (size_t)nb1 * cap1 : (size_t)n;
  // scratch: [hist1 512 | cursor1 512] zeroed, then [cbase 513 | tprefix 513 | pbeg dummy 514]
  ENSURE(c->pt_work, (size_t)(1024 + 513 + 513 + 516) * 4);
  u32 *hist1 = c->pt_work.as<u32>(), *cursor1 = hist1 + 512, *cbase = cursor1 + 512, *tprefix = cbase + 513, *dummy = tprefix + 513;
  if (padded) {                                              // its own cursors, cleared by the scan that reads them
    if (!c->gf_cur.p) {
      ENSURE(c->gf_cur, 512 * 4);
      HIPCHK(hipMemsetAsync(c->gf_cur.p, 0, 512 * 4, st));
    }
    cursor1 = c->gf_cur.as<u32>();
  } else HIPCHK(hipMemsetAsync(c->pt_work.p, 0, 1024 * 4, st));
  ENSURE(c->seg_k0, room * 8);
  ENSURE(c->seg_v0, room * 4);
  const SRC src{W, plan_fields(plan, cb), bit_n};
  const u32 tiles1 = (n + PT_TILE - 1) / PT_TILE;
  if (!padded) {
    hipLaunchKernelGGL(k_pt_hist1<SRC>, dim3(tiles1 < 512 ? tiles1 : 512), dim3(1024), 0, st, src, n, d1, hist1);
    hipLaunchKernelGGL(k_pt_scan1, dim3(1), dim3(1024), 0, st, (const u32 *)hist1, d1, 0u, cbase, tprefix, dummy, dummy + 513, 0u);
  }
  static const bool gs_small = getenv("HUMID_GS_THREADS") ? atoi(getenv("HUMID_GS_THREADS")) == 512 : false;  // (experiments: 512-thread tiles are 7 us slower -- shorter runs per bin)
  if (gs_small)
    hipLaunchKernelGGL((k_pt_scatter<1, SRC, 512>), dim3((n + 4095) / 4096), dim3(512), 0, st, src, n, (const u64 *)nullptr,
                       (const u32 *)nullptr, (const u32 *)nullptr, (const u32 *)nullptr, d1, d2, (const u32 *)cbase, cursor1,
                       c->seg_k0.as<u64>(), c->seg_v0.as<u32>(), (u32 *)nullptr, cap1, &c->d_ctr[CTR_GOVER]);
  else
    hipLaunchKernelGGL((k_pt_scatter<1, SRC>), dim3(tiles1), dim3(1024), 0, st, src, n, (const u64 *)nullptr,
                       (const u32 *)nullptr, (const u32 *)nullptr, (const u32 *)nullptr, d1, d2, (const u32 *)cbase, cursor1,
                       c->seg_k0.as<u64>(), c->seg_v0.as<u32>(), (u32 *)nullptr, cap1, &c->d_ctr[CTR_GOVER]);
  if (padded)
    hipLaunchKernelGGL(k_pt_scan1, dim3(1), dim3(1024), 0, st, (const u32 *)cursor1, d1, 0u, cbase, tprefix, dummy, dummy + 513, cap1, cursor1);
  // up to three launches over the coarse bins, by bin size; which sizes cannot occur is known from n alone only
  // roughly (the bins of a skewed key can be any size) unless the bins are padded, so only the impossible
  // ones are left out
  c->gf_valid = padded ? (const u32 *)(cbase + nb1) : (const u32 *)nullptr;    // words the order holds (< n: a bin was full)
  const u32 largest = padded ? cap1 : n;
  hipLaunchKernelGGL((k_group_fine<SRC, 0>), dim3(nb1), dim3(GF_THREADS), 0, st, src,
                     (const u64 *)c->seg_k0.as<u64>(), (const u32 *)c->seg_v0.as<u32>(), (const u32 *)cbase, d1, d2, ws, vs, cap1);
  if (largest > GF_SMALL)
    hipLaunchKernelGGL((k_group_fine<SRC, 1>), dim3(nb1), dim3(GF_THREADS), 0, st, src,
                       (const u64 *)c->seg_k0.as<u64>(), (const u32 *)c->seg_v0.as<u32>(), (const u32 *)cbase, d1, d2, ws, vs, cap1);
  if (largest > GF_MID)
    hipLaunchKernelGGL((k_group_fine<SRC, 2>), dim3(nb1), dim3(GF_THREADS), 0, st, src,
                       (const u64 *)c->seg_k0.as<u64>(), (const u32 *)c->seg_v0.as<u32>(), (const u32 *)cbase, d1, d2, ws, vs, cap1);
  HIPCHK(hipGetLastError());
  return HUMID_OK;
}

// A combination whose key is ONE stretch of the word (a single segment, or neighbouring segments):
// the words themselves are the sort keys over that bit range and come out in bucket order (ws), the
// positions ride along as values (vs) -- no key array, and no gather of the words afterwards (44 us
// and 320 MB of traffic at 10 M reads; the 8-byte keys cost the sort 19 us more: tools/sort_probe.hip).
// *done = false: the key is not one stretch, nothing was queued.
static int sort_words_by_stretch(humid_ctx *c, const ComboPlan &plan, u32 cb, const u64 *W, u32 n, u64 *ws, u32 *vs, bool *done) {
  u32 bit_lo = 0, bit_n = 0;
  bool stretch = plan.nfield[cb] >= 1;
  for (u32 f = 0; stretch && f < plan.nfield[cb]; f++) {
    if (f + 1 < plan.nfield[cb] && plan.shift[cb][f] != plan.shift[cb][f + 1] + plan.width[cb][f + 1]) stretch = false;
    bit_n += plan.width[cb][f];
    bit_lo = plan.shift[cb][f];
  }
  *done = stretch && bit_n >= 1 && bit_lo + bit_n <= 64;
  if (!*done) return HUMID_OK;
  return sort_pairs_in<u64, u32>(c, PtrIn<u64>{W}, ws, IotaIn{}, vs, n, bit_lo, bit_lo + bit_n);
}

// words of the unique array in bucket order of combination `seg` (> 0): ws[i] = the word walked at
// position i, vs[i] = its walk index.  Keys of <= 24 bits: two-level grouping; one stretch of the word:
// the words themselves as sort keys; else keys + sort + gather.  Scratch: seg_k0 / seg_v0 / seg_ks.
template <class WT>
static int bucket_order(humid_ctx *c, const ComboPlan &plan, u32 seg, const WT *g_word, u32 U, WT *ws, u32 *vs, bool may_pad = false) {
  hipStream_t st = c->stream;
  u32 kb = 0;                                            // key bits of THIS combination
  for (u32 f = 0; f < plan.nfield[seg]; f++) kb += plan.width[seg][f];
  if (kb == 0) kb = 1;
  bool stretch = false;
  if (std::is_same<WT, u64>::value) {
    TRY((group_words_by_stretch<FieldsSrc, u64>(c, plan, seg, (const u64 *)g_word, U, (u64 *)ws, vs, &stretch, may_pad)));
    if (!stretch) TRY(sort_words_by_stretch(c, plan, seg, (const u64 *)g_word, U, (u64 *)ws, vs, &stretch));
  } else {
    // two-word words: the keys are grouped (scratch), the words follow through the grouped positions
    TRY((group_words_by_stretch<FieldsSrcW2, W2>(c, plan, seg, (const W2 *)g_word, U, c->seg_ks.as<u64>(), vs, &stretch, may_pad)));
    if (stretch) hipLaunchKernelGGL(k_gather_bucket_words<WT>, dim3(blocks_for(U)), dim3(256), 0, st, g_word, vs, U, ws, c->gf_valid);
  }
  if (stretch) return HUMID_OK;
  const ComboFields cf = plan_fields(plan, seg);
  if (kb <= 32) {
    hipLaunchKernelGGL((k_combo_keys<u32, WT>), dim3(blocks_for(U)), dim3(256), 0, st, g_word, U, cf,
                       c->seg_k0.as<u32>(), c->seg_v0.as<u32>());
    TRY(sort_pairs<u32, u32>(c, c->seg_k0.as<u32>(), c->seg_ks.as<u32>(), c->seg_v0.as<u32>(), vs, U, 0, kb));
  } else {
    hipLaunchKernelGGL((k_combo_keys<u64, WT>), dim3(blocks_for(U)), dim3(256), 0, st, g_word, U, cf,
                       c->seg_k0.as<u64>(), c->seg_v0.as<u32>());
    TRY(sort_pairs<u64, u32>(c, c->seg_k0.as<u64>(), c->seg_ks.as<u64>(), c->seg_v0.as<u32>(), vs, U, 0, kb));
  }
  hipLaunchKernelGGL(k_gather_bucket_words<WT>, dim3(blocks_for(U)), dim3(256), 0, st, g_word, vs, U, ws);
  HIPCHK(hipGetLastError());
  return HUMID_OK;
}

// ---- stage B: neighbours + clusters over a sorted unique array ---------------------------
// g_word[U] ascending, g_cnt[U] (device; the context's own arrays on one GPU, the gathered
// arrays of all ranks on several).  Leaves deg/nbr_off/nbr_idx/cl_of/maxleaf/cl_size/flag/
// pos/cid/ismax in the context.
// ext_edges != nullptr: the neighbour pairs are GIVEN (multi-GPU: every rank searched its share,
// humid_stage_pairs, and the shares were all-gathered); otherwise they are searched here.
// WT: u64 (n <= 32) or W2 (33 <= n <= 64, two uint64 per word).
template <class WT>
static int stage_graph(humid_ctx *c, const WT *g_word, const u32 *g_cnt, u32 U, u32 word_nt,
                       u32 distance, u32 method, humid_summary &s, u32 &n_pair_segs_out,
                       const u64 *ext_edges = nullptr, u64 n_ext_edges = 0) {
  hipStream_t st = c->stream;
  c->g_word = g_word;
  c->g_wpr = (u32)(sizeof(WT) / 8);
  c->g_cnt = g_cnt;
  c->gU = U;
  c->cg_valid = false;
  // ---------------- 3. neighbours -----------------
  // deg has U+1 entries (last stays 0) so that one exclusive scan yields nbr_off[U] = 2E
  ENSURE(c->deg, (size_t)(U + 1) * 4);
  ENSURE(c->nbr_off, (size_t)(U + 1) * 4);
  ENSURE(c->parent, (size_t)U * 4);
  ENSURE(c->csize, (size_t)U * 4);
  ENSURE(c->cur, (size_t)U * 4);
  ENSURE(c->small_roots, ((size_t)U / 3 + 2) * 4);
  HIPCHK(hipMemsetAsync(&c->d_ctr[CTR_EDGES], 0, (CTR_SMALLROOTS - CTR_EDGES + 1) * sizeof(ull), st));
  hipLaunchKernelGGL(k_graph_init, dim3(blocks_for((u64)U + 1)), dim3(256), 0, st, c->parent.as<u32>(),
                     c->deg.as<u32>(), c->csize.as<u32>(), c->cur.as<u32>(), U);
  u64 E = 0, M = 0, Mbig = 0;
  u32 n_pair_segs = 0;
  const ComboPlan plan = make_plan(word_nt, distance, U, c->force_segments, true);
  EarlierMasksT<WT> d_masks;                         // masks of all combos, for the first-combo rule
  for (u32 t = 0; t < MAX_COMBOS; t++) d_masks.m[t] = w_from<WT>(plan.mask[t]);
  auto fields_of = [&](u32 cb) {
    ComboFields cf;
    cf.nf = plan.nfield[cb];
    for (u32 f = 0; f < MAX_FIELDS; f++) { cf.shift[f] = plan.shift[cb][f]; cf.width[f] = plan.width[cb][f]; }
    return cf;
  };
  const bool given = ext_edges != nullptr;
  const bool search = !given && distance > 0 && U > 1;
  // directional method: only neighbour pairs a climb or a flood can cross join two components
  // (joins_for_clustering); maximum method: all of them
  const u32 *join_cnt = (method & 1) ? nullptr : g_cnt;

  // one bucket holding every word (d >= n, or d too large for any pigeonhole plan): U^2 / 2
  // comparisons and, at such distances, nearly as many pairs -- beyond a few 10^5 words the pair
  // list cannot fit 32-bit CSR offsets anyway; refuse before spending minutes to find that out
  if (search && plan.ncombo == 1 && plan.key_bits == 0 && U > (1u << 18))
    return fail(c, HUMID_E_OVERFLOW, "distance %u over %u-nt words compares all pairs of %u unique words: too many neighbour pairs",
                distance, word_nt, U);
  if (given && n_ext_edges) {
    hipLaunchKernelGGL(k_edges_apply<false>, dim3(grid_stride_blocks(n_ext_edges)), dim3(256), 0, st, ext_edges,
                       n_ext_edges, U, c->deg.as<u32>(), c->parent.as<u32>(), (const u32 *)nullptr,
                       (u32 *)nullptr, (u32 *)nullptr, c->d_ctr, join_cnt);
    hipLaunchKernelGGL(k_comp_stats, dim3(blocks_for(U)), dim3(256), 0, st, c->deg.as<u32>(),
                       c->parent.as<u32>(), U, c->csize.as<u32>());
    hipLaunchKernelGGL(k_comp_count, dim3(512), dim3(256), 0, st, c->deg.as<u32>(), c->parent.as<u32>(),
                       c->csize.as<u32>(), U, c->d_ctr, c->small_roots.as<u32>());
  }
  // buckets beyond k_pairs' bounded walk (c->walk_max words; 0 = walk to the end of the bucket)
  const u32 walk_max = c->walk_max;
  u64 big_mask = 0;
  std::vector<BigRun> h_runs[MAX_COMBOS];
  auto walked = [&](u32 seg, const WT *&W, const u32 *&V) {
    W = seg ? c->seg_ws.as<WT>() + (size_t)(seg - 1) * U : g_word;
    V = seg ? c->seg_vs.as<u32>() + (size_t)(seg - 1) * U : nullptr;
  };
  auto big_find = [&](u32 seg) -> int {
    const WT *W; const u32 *V;
    walked(seg, W, V);
    const BigRun *d_runs = nullptr;
    return find_big_runs<WT>(c, W, U, w_from<WT>(plan.mask[seg]), walk_max, seg, h_runs[seg], &d_runs);
  };
  auto big_tiles = [&](u32 seg, int mode) -> int {
    const std::vector<BigRun> &r = h_runs[seg];
    if (r.size() < 2) return HUMID_OK;
    const WT *W; const u32 *V;
    walked(seg, W, V);
    const u32 cap = U / (walk_max + 2) + 1;
    const ull tiles = r.back().tile0;
    const u32 grid = (u32)std::min<ull>(tiles, 1u << 20);
#define BIG_TILES(P0, M)                                                                                          \
  hipLaunchKernelGGL((k_pairs_tiles<P0, M, WT>), dim3(grid), dim3(PT2_THREADS), 0, st, W, V,                       \
                     c->big_runs.as<BigRun>() + (size_t)seg * cap, (u32)r.size() - 1, tiles, d_masks, seg, distance, \
                     walk_max, c->deg.as<u32>(), c->parent.as<u32>(), c->nbr_off.as<u32>(), c->cur.as<u32>(),      \
                     c->nbr_idx.as<u32>(), join_cnt)
    if (seg == 0 && mode == PM_COUNT) BIG_TILES(true, PM_COUNT);
    else if (seg == 0) BIG_TILES(true, PM_FILL);
    else if (mode == PM_COUNT) BIG_TILES(false, PM_COUNT);
    else BIG_TILES(false, PM_FILL);
#undef BIG_TILES
    HIPCHK(hipGetLastError());
    return HUMID_OK;
  };
  if (search) {
    const u32 nseg = plan.ncombo;
    n_pair_segs = nseg < 8 ? nseg : 8;
    ENSURE(c->had, (size_t)nseg * U * 4);                   // per combination and position: pairs found, distance to the first
    if (nseg > 1) {
      ENSURE(c->seg_k0, (size_t)U * 8);
      ENSURE(c->seg_v0, (size_t)U * 4);
      ENSURE(c->seg_ks, (size_t)U * 8);                     // sorted keys: scratch, not kept
      ENSURE(c->seg_vs, (size_t)(nseg - 1) * U * 4);        // ranks in bucket order, per combo
      ENSURE(c->seg_ws, (size_t)(nseg - 1) * U * sizeof(WT));   // words in bucket order, per combo
    }
    // phase A: bucket order per combo; degrees and component forest
    for (u32 seg = 0; seg < nseg; seg++) {
      if (seg == 0) {
        if (c->kev_on) HIPCHK(hipEventRecord(c->kev[20], st));
        hipLaunchKernelGGL((k_pairs<true, PM_COUNT, WT>), dim3(blocks_for(U)), dim3(256), 0, st, g_word,
                           (const u32 *)nullptr, U, 0u, U, w_from<WT>(plan.mask[seg]), d_masks, seg, distance, c->deg.as<u32>(),
                           c->parent.as<u32>(), (const u32 *)nullptr, (u32 *)nullptr, (u32 *)nullptr,
                           (u32 *)nullptr, (const u32 *)nullptr, (u64 *)nullptr, c->had.as<u32>(), walk_max,
                           &c->d_ctr[CTR_BIGMASK], join_cnt);
      } else {
        u32 *vs = c->seg_vs.as<u32>() + (size_t)(seg - 1) * U;
        WT *ws = c->seg_ws.as<WT>() + (size_t)(seg - 1) * U;
        TRY(bucket_order<WT>(c, plan, seg, g_word, U, ws, vs));
        if (seg < 8) if (c->kev_on) HIPCHK(hipEventRecord(c->kev[20 + 2 * seg], st));
        hipLaunchKernelGGL((k_pairs<false, PM_COUNT, WT>), dim3(blocks_for(U)), dim3(256), 0, st, ws,
                           vs, U, 0u, U, w_from<WT>(plan.mask[seg]), d_masks, seg, distance, c->deg.as<u32>(), c->parent.as<u32>(),
                           (const u32 *)nullptr, (u32 *)nullptr, (u32 *)nullptr, (u32 *)nullptr,
                           (const u32 *)nullptr, (u64 *)nullptr, c->had.as<u32>() + (size_t)seg * U, walk_max,
                           &c->d_ctr[CTR_BIGMASK], join_cnt);
      }
      if (seg < 8) if (c->kev_on) HIPCHK(hipEventRecord(c->kev[21 + 2 * seg], st));
    }
    hipLaunchKernelGGL(k_comp_stats, dim3(blocks_for(U)), dim3(256), 0, st, c->deg.as<u32>(),
                       c->parent.as<u32>(), U, c->csize.as<u32>());
    hipLaunchKernelGGL(k_comp_count, dim3(512), dim3(256), 0, st, c->deg.as<u32>(), c->parent.as<u32>(),
                       c->csize.as<u32>(), U, c->d_ctr, c->small_roots.as<u32>());
  }
  TRY(exscan_u32(c, c->deg.as<u32>(), c->nbr_off.as<u32>(), (u64)U + 1));
  if (search || (given && n_ext_edges)) {
    HIPCHK(hipGetLastError());
    TRY(read_counters(c, c->nbr_off.as<u32>() + U));   // h_ctr[CTR_N-1] = 2E
    if (c->h_ctr[CTR_OVERFULL]) return fail(c, HUMID_E_INVALID, "malformed edge list (node index out of range)");
    // the degrees summed in 64 bits (k_comp_count): the 32-bit scan below it may have wrapped
    if (c->h_ctr[CTR_EDGES] > 0xffffffffull)
      return fail(c, HUMID_E_OVERFLOW, "%llu neighbour pairs exceed the 32-bit adjacency offsets", (ull)(c->h_ctr[CTR_EDGES] / 2));
    big_mask = search ? c->h_ctr[CTR_BIGMASK] : 0;
    if (big_mask) {
      // some bucket is longer than k_pairs walks: find those runs, count their remaining pairs as
      // tiles, and take the component statistics and the offsets again
      for (u32 seg = 0; seg < plan.ncombo; seg++)
        if (big_mask >> seg & 1) {
          TRY(big_find(seg));
          TRY(big_tiles(seg, PM_COUNT));
        }
      HIPCHK(hipMemsetAsync(&c->d_ctr[CTR_EDGES], 0, 3 * sizeof(ull), st));
      HIPCHK(hipMemsetAsync(&c->d_ctr[CTR_SMALLROOTS], 0, sizeof(ull), st));
      HIPCHK(hipMemsetAsync(c->csize.p, 0, (size_t)U * 4, st));
      hipLaunchKernelGGL(k_comp_stats, dim3(blocks_for(U)), dim3(256), 0, st, c->deg.as<u32>(),
                         c->parent.as<u32>(), U, c->csize.as<u32>());
      hipLaunchKernelGGL(k_comp_count, dim3(512), dim3(256), 0, st, c->deg.as<u32>(), c->parent.as<u32>(),
                         c->csize.as<u32>(), U, c->d_ctr, c->small_roots.as<u32>());
      TRY(exscan_u32(c, c->deg.as<u32>(), c->nbr_off.as<u32>(), (u64)U + 1));
      HIPCHK(hipGetLastError());
      TRY(read_counters(c, c->nbr_off.as<u32>() + U));
      if (c->h_ctr[CTR_EDGES] > 0xffffffffull)
        return fail(c, HUMID_E_OVERFLOW, "%llu neighbour pairs exceed the 32-bit adjacency offsets", (ull)(c->h_ctr[CTR_EDGES] / 2));
    }
    const u64 twoE = c->h_ctr[CTR_N - 1] & 0xffffffffull;
    E = twoE / 2;
    M = c->h_ctr[CTR_NONSINGLE];
    Mbig = c->h_ctr[CTR_MEMBERS];
  }
  s.edges = c->E = E;
  s.nonsingle = c->M = M;
  ENSURE(c->nbr_idx, (size_t)(2 * E + 1) * 4);
  if (E > 0) {
    if (given)
      hipLaunchKernelGGL(k_edges_apply<true>, dim3(grid_stride_blocks(n_ext_edges)), dim3(256), 0, st, ext_edges,
                         n_ext_edges, U, (u32 *)nullptr, (u32 *)nullptr, c->nbr_off.as<u32>(),
                         c->cur.as<u32>(), c->nbr_idx.as<u32>(), c->d_ctr);
    // phase B: same loops, now writing the CSR rows
    for (u32 seg = 0; !given && seg < plan.ncombo; seg++) {
      if (seg < 8) if (c->kev_on) HIPCHK(hipEventRecord(c->kev[4 + 2 * seg], st));
      if (seg == 0) {
        hipLaunchKernelGGL((k_pairs<true, PM_FILL, WT>), dim3(blocks_for(U)), dim3(256), 0, st, g_word,
                           (const u32 *)nullptr, U, 0u, U, w_from<WT>(plan.mask[seg]), d_masks, seg, distance, (u32 *)nullptr,
                           (u32 *)nullptr, c->nbr_off.as<u32>(), c->cur.as<u32>(), c->nbr_idx.as<u32>(),
                           (u32 *)nullptr, (const u32 *)nullptr, (u64 *)nullptr, c->had.as<u32>(), walk_max);
        if (big_mask & 1) TRY(big_tiles(0, PM_FILL));
      } else {
        const u32 *vs = c->seg_vs.as<u32>() + (size_t)(seg - 1) * U;
        const WT *ws = c->seg_ws.as<WT>() + (size_t)(seg - 1) * U;
        hipLaunchKernelGGL((k_pairs<false, PM_FILL, WT>), dim3(blocks_for(U)), dim3(256), 0, st, ws,
                           vs, U, 0u, U, w_from<WT>(plan.mask[seg]), d_masks, seg, distance, (u32 *)nullptr, (u32 *)nullptr,
                           c->nbr_off.as<u32>(), c->cur.as<u32>(), c->nbr_idx.as<u32>(),
                           (u32 *)nullptr, (const u32 *)nullptr, (u64 *)nullptr, c->had.as<u32>() + (size_t)seg * U,
                           walk_max);
        if (big_mask >> seg & 1) TRY(big_tiles(seg, PM_FILL));
      }
      if (seg < 8) if (c->kev_on) HIPCHK(hipEventRecord(c->kev[5 + 2 * seg], st));
    }
    hipLaunchKernelGGL(k_sort_lists, dim3(blocks_for(U)), dim3(256), 0, st, c->nbr_off.as<u32>(), U,
                       c->nbr_idx.as<u32>());
  }
  if (!c->lean_events) HIPCHK(hipEventRecord(c->ev[2], st));

  // clusters
  TRY(cluster_stage(c, g_cnt, U, M, Mbig, method));
  // one GPU: the graph is over this context's own unique words, so the per-slot result words can
  // be written in the same pass (stage C then skips k_slot_results)
  const bool own = ((const void *)g_word == c->s_word.p) && U == (u32)c->U && !given;
  hipLaunchKernelGGL(k_finalize_nodes, dim3(blocks_for(U)), dim3(256), 0, st, c->cl_of.as<u32>(),
                     c->pos.as<u32>(), c->maxleaf.as<u32>(), U, c->cid.as<u32>(), c->ismax.as<u8>(),
                     own ? c->s_first.as<u32>() : (const u32 *)nullptr,
                     own ? c->s_slot.as<u32>() : (const u32 *)nullptr,
                     own ? c->slot_out.as<u64>() : (u64 *)nullptr);
  c->slots_done = own;
  HIPCHK(hipGetLastError());
  n_pair_segs_out = n_pair_segs;
  return HUMID_OK;
}


// ---- the compact graph from marked ids + pairs (kernels_cgraph.hip.h): shared by the single-GPU search
// (pairs appended into regions, counts by id) and the multi-GPU pass (pair records in global ids) ----
struct CgSource {
  EdgeRegs er;                 // pairs in id space: regions + far list (recs == null); set to the compact pairs on return
  const ulonglong2 *recs;      // or: pair records {a << 32 | b, count a | count b << 32}
  u32 n_recs;
  const RecSegs *segs = nullptr;   // or: several record arrays (the multi-GPU pass: interior, crossing, flagged-interior of the others)
  const u32 *cnt_by_id;        // counts by id (with plain pairs)
  u32 n_ids;                   // id space = bits of the bitmap c->cg_bits (zeroed, then marked, by the caller)
  u64 pairs_bound;             // no more pairs than this can be in the source
};
struct CgStatus {
  bool overflow = false;       // an append region was full: `wanted` says how much room the search wants in all
  bool group_over = false;     // a padded coarse bin of a bucket order was full (CTR_GOVER)
  u64 wanted = 0, big_mask = 0, E = 0, M = 0, Mbig = 0;
};
static GraphArrays cg_arrays(humid_ctx *c) {
  return GraphArrays{c->cg_deg.as<u32>(), c->cg_parent.as<u32>(), c->cg_csize.as<u32>(), c->cg_off.as<u32>(), c->cg_idx.as<u32>(),
                     c->cg_cl_of.as<u32>(), c->cg_maxleaf.as<u32>(), c->cg_cl_size.as<u64>()};
}
// rank structure, nodes, compact pairs, degrees, forest, CSR rows (ascending), component sizes, the trivial
// components; ONE host wait at the end (every launch before it is sized by bounds: nodes <= 2 x pairs).
static int cg_build(humid_ctx *c, CgSource &src, u32 method, CgStatus &out) {
  hipStream_t st = c->stream;
  const u32 n_words = (((src.n_ids + 31) / 32) + 7) & ~7u, n_blk = n_words / 8;
  const u64 pb = std::max<u64>(src.pairs_bound, 1);
  if (2 * pb + 2 > 0xffffffffull) return fail(c, HUMID_E_OVERFLOW, "too many neighbour pairs");
  const u32 Mb = (u32)std::min<u64>(src.n_ids, 2 * pb);
  ENSURE(c->cg_blk, ((size_t)n_blk + 1) * 4);
  ENSURE(c->cg_nodes, ((size_t)Mb + 1) * 4);
  ENSURE(c->cg_ncnt, ((size_t)Mb + 1) * 4);
  ENSURE(c->cg_deg, ((size_t)Mb + 2) * 4);
  ENSURE(c->cg_off, ((size_t)Mb + 2) * 4);
  ENSURE(c->cg_parent, ((size_t)Mb + 1) * 4);
  ENSURE(c->cg_csize, ((size_t)Mb + 1) * 4);
  ENSURE(c->cg_curs, ((size_t)Mb + 1) * 4);
  ENSURE(c->cg_idx, (size_t)(2 * pb + 1) * 4);
  ENSURE(c->cg_cl_of, ((size_t)Mb + 1) * 4);
  ENSURE(c->cg_maxleaf, ((size_t)Mb + 1) * 4);
  ENSURE(c->cg_cl_size, ((size_t)Mb + 1) * 8);
  ENSURE(c->small_roots, ((size_t)Mb / 3 + 2) * 4);
  ENSURE(c->small, 64);
  TRY(exscan_in<u32>(c, BitsBlockIn{c->cg_bits.as<u32>(), n_blk}, c->cg_blk.as<u32>(), (u64)n_blk + 1));
  const BitRank br{c->cg_bits.as<u32>(), c->cg_blk.as<u32>()};
  const u32 *m_dev = c->cg_blk.as<u32>() + n_blk;
  const GraphArrays g = cg_arrays(c);
  hipLaunchKernelGGL(k_nodes_init, dim3(blocks_for(n_words)), dim3(256), 0, st, br, n_words, src.cnt_by_id, c->cg_nodes.as<u32>(),
                     c->cg_ncnt.as<u32>(), g.deg, g.parent, g.csize, c->cg_curs.as<u32>(), n_blk);
  const bool by_count = (method & 1) == 0;
  if (src.segs) {
    const u32 n_all = src.segs->first[REC_SEGS];
    u32 n_max = 1;
    for (u32 q = 0; q < REC_SEGS; q++) n_max = std::max(n_max, src.segs->n[q]);
    ENSURE(c->cg_far, ((size_t)n_all + 1) * 8);
    hipLaunchKernelGGL(k_segs_relabel, dim3(std::min<u32>(blocks_for(n_max), 4096), REC_SEGS), dim3(256), 0, st, *src.segs, src.n_ids, br,
                       c->cg_far.as<u64>(), c->cg_ncnt.as<u32>(), g.deg, g.parent, by_count);
    src.er.far = c->cg_far.as<u64>();
    src.er.n_far = n_all;
  } else if (src.recs) {
    ENSURE(c->cg_far, ((size_t)src.n_recs + 1) * 8);
    if (src.n_recs)
      hipLaunchKernelGGL(k_records_relabel, dim3(blocks_for(src.n_recs)), dim3(256), 0, st, src.recs, src.n_recs, src.n_ids, br,
                         c->cg_far.as<u64>(), c->cg_ncnt.as<u32>(), g.deg, g.parent, by_count);
    src.er.far = c->cg_far.as<u64>();
    src.er.n_far = src.n_recs;
  } else {
    const u32 gx = (u32)std::min<u64>(std::max<u64>(blocks_for(std::max<u64>(src.er.cap_r, src.er.n_far)), 1), 4096);
    hipLaunchKernelGGL(k_pairs_relabel, dim3(gx, ER_REGIONS + 1), dim3(256), 0, st, src.er, br, (const u32 *)c->cg_ncnt.as<u32>(),
                       g.deg, g.parent, by_count);
  }
  TRY(exscan_in<u32>(c, DegIn{g.deg, m_dev}, g.off, (u64)Mb + 1));
  {
    // CSR rows and component sizes side by side (one launch: kernels_cgraph.hip.h)
    const u32 gx = (u32)std::min<u64>(std::max<u64>(blocks_for(std::max<u64>(src.er.cap_r, src.er.n_far)), 1), 4096);
    hipLaunchKernelGGL(k_fill_and_stats, dim3(gx * (ER_REGIONS + 1) + blocks_for(Mb)), dim3(256), 0, st, src.er, (const u32 *)g.off,
                       c->cg_curs.as<u32>(), g.idx, c->small.as<u32>(), gx, (const u32 *)g.deg, g.parent, Mb, g.csize, m_dev);
  }
  hipLaunchKernelGGL(k_sort_lists, dim3(blocks_for(Mb)), dim3(256), 0, st, (const u32 *)g.off, Mb, g.idx);
  if (c->kev_on) HIPCHK(hipEventRecord(c->kev[2], st));
  const u32 tg = (u32)std::min<u64>(std::max<u64>(blocks_for(Mb), 1), 512);
  if (method == HUMID_METHOD_MAXIMUM)
    hipLaunchKernelGGL(k_cg_trivial<true>, dim3(tg), dim3(256), 0, st, (const u32 *)g.parent, (const u32 *)g.csize, m_dev,
                       (const u32 *)c->cg_ncnt.as<u32>(), (const u32 *)g.off, (const u32 *)g.idx, g.cl_of, g.maxleaf, g.cl_size, c->d_ctr,
                       c->small_roots.as<u32>());
  else
    hipLaunchKernelGGL(k_cg_trivial<false>, dim3(tg), dim3(256), 0, st, (const u32 *)g.parent, (const u32 *)g.csize, m_dev,
                       (const u32 *)c->cg_ncnt.as<u32>(), (const u32 *)g.off, (const u32 *)g.idx, g.cl_of, g.maxleaf, g.cl_size, c->d_ctr,
                       c->small_roots.as<u32>());
  HIPCHK(hipGetLastError());
  TRY(read_counters(c, g.off + Mb, c->small.as<u32>(), m_dev));              // 2E, pairs the fullest region wanted, M
  out.wanted = (c->h_ctr[CTR_N - 2] & 0xffffffffull) * ER_REGIONS;            // (as a total: every region has the same room)
  out.overflow = (c->h_ctr[CTR_EOVER] & 0xffffffffull) != 0;
  out.group_over = c->h_ctr[CTR_GOVER] != 0;
  out.big_mask = c->h_ctr[CTR_BIGMASK];
  out.E = (c->h_ctr[CTR_N - 1] & 0xffffffffull) / 2;
  out.M = c->h_ctr[CTR_N - 3] & 0xffffffffull;
  out.Mbig = c->h_ctr[CTR_MEMBERS];
  return HUMID_OK;
}
// the components of 3 and more nodes, then the nodes that created no cluster as a bitmap over the ids with
// its rank structure (c->cg_nbits zeroed by the caller; cg_nblk)
static int cg_cluster_rest(humid_ctx *c, u32 n_ids, u64 M, u64 Mbig, u32 method) {
  hipStream_t st = c->stream;
  const u32 n_words = (((n_ids + 31) / 32) + 7) & ~7u, n_blk = n_words / 8;
  const GraphArrays g = cg_arrays(c);
  ENSURE(c->cg_nblk, ((size_t)n_blk + 1) * 4);
  if (M > 0) TRY(cluster_kernels(c, g, c->cg_ncnt.as<u32>(), (u32)M, M, Mbig, method, true));
  else if (c->kev_on) HIPCHK(hipEventRecord(c->kev[3], st));
  if (M > 0)
    hipLaunchKernelGGL(k_noncreator_bits, dim3(blocks_for(M)), dim3(256), 0, st, (const u32 *)g.cl_of, c->cg_nodes.as<u32>(), (u32)M,
                       c->cg_nbits.as<u32>());
  TRY(exscan_in<u32>(c, BitsBlockIn{c->cg_nbits.as<u32>(), n_blk}, c->cg_nblk.as<u32>(), (u64)n_blk + 1));
  HIPCHK(hipGetLastError());
  return HUMID_OK;
}

// ---- stage B on the COMPACT graph (kernels_cgraph.hip.h): the single-GPU pipeline's form ---------
// Same contract as stage_graph (neighbours + clusters of the ascending unique array g_word / g_cnt),
// but every graph and cluster array lives on the M leaves that have neighbours; the per-unique-word
// view (deg / nbr_off / nbr_idx / cl_of / ... of the context) is only built when an accessor asks for
// it (expand_compact).  Leaves: slot_out (own = the graph is over this context's unique words) or
// cid / ismax (own = false), c->cg_* and the cluster count on the device (n_clusters_compact).
// ext_edges != nullptr: the pairs are GIVEN as (smaller << 32 | larger) walk indices (edit distance).
template <class WT>
static int stage_graph_compact(humid_ctx *c, const WT *g_word, const u32 *g_cnt, u32 U, u32 word_nt,
                               u32 distance, u32 method, humid_summary &s, u32 &n_pair_segs_out,
                               const u64 *ext_edges = nullptr, u64 n_ext_edges = 0) {
  hipStream_t st = c->stream;
  c->g_word = g_word;
  c->g_wpr = (u32)(sizeof(WT) / 8);
  c->g_cnt = g_cnt;
  c->gU = U;
  c->cg_valid = false;
  c->cg_expanded = false;
  const ComboPlan plan = make_plan(word_nt, distance, U, c->force_segments, true);
  EarlierMasksT<WT> d_masks;
  for (u32 t = 0; t < MAX_COMBOS; t++) d_masks.m[t] = w_from<WT>(plan.mask[t]);
  const bool given = ext_edges != nullptr;
  const bool search = !given && distance > 0 && U > 1;
  if (search && plan.ncombo == 1 && plan.key_bits == 0 && U > (1u << 18))
    return fail(c, HUMID_E_OVERFLOW, "distance %u over %u-nt words compares all pairs of %u unique words: too many neighbour pairs",
                distance, word_nt, U);
  if (given && n_ext_edges > 0x7fffffffull) return fail(c, HUMID_E_OVERFLOW, "too many neighbour pairs");
  const u32 walk_max = c->walk_max;
  const u32 nseg = search ? plan.ncombo : 0;
  const u32 n_words = (((U + 31) / 32) + 7) & ~7u, n_blk = n_words / 8;
  c->cg_nblocks = n_blk;
  ENSURE(c->cg_bits, (size_t)n_words * 4);
  ENSURE(c->cg_nbits, (size_t)n_words * 4);
  ENSURE(c->cg_blk, ((size_t)n_blk + 1) * 4);
  ENSURE(c->cg_nblk, ((size_t)n_blk + 1) * 4);
  ENSURE(c->cg_cur, (size_t)(ER_REGIONS * ER_STRIDE + 8) * 4);
  ENSURE(c->small_roots, ((size_t)U / 3 + 2) * 4);
  ENSURE(c->small, 64);
  if (nseg > 1) {
    ENSURE(c->seg_k0, (size_t)U * 8);
    ENSURE(c->seg_v0, (size_t)U * 4);
    ENSURE(c->seg_ks, (size_t)U * 8);
    ENSURE(c->seg_vs, (size_t)(nseg - 1) * U * 4);
    ENSURE(c->seg_ws, (size_t)(nseg - 1) * U * sizeof(WT));
  }
  u32 *bad = c->cg_cur.as<u32>() + ER_REGIONS * ER_STRIDE;       // malformed given pair
  const u64 *far = given ? ext_edges : nullptr;
  u64 n_far = given ? n_ext_edges : 0;
  u64 E = 0, M = 0, Mbig = 0;
  CgStatus cgs;
  EdgeRegs er;
  // the bucket order of a combination is made ONCE: the grouping places equal keys with atomics, so a
  // second run may order a bucket differently -- and the near / far split of a large bucket (walk
  // distance) must be the same in the search that follows the tiles as in the one before them
  bool ordered_seg[MAX_COMBOS] = {false};
  const u32 *seg_valid[MAX_COMBOS] = {nullptr};
  for (int attempt = 0;; attempt++) {
    if (attempt > 5) return fail(c, HUMID_E_INVALID, "internal: the pair list does not settle");
    if (search && c->cg_ecap == 0) c->cg_ecap = std::max<u64>((u64)U / 4, 4096);
    const u64 ecap = search ? c->cg_ecap : ER_REGIONS;
    if (ecap / ER_REGIONS + 1 > 0xfffffff0ull) return fail(c, HUMID_E_OVERFLOW, "too many neighbour pairs");
    er.cap_r = (u32)((ecap + ER_REGIONS - 1) / ER_REGIONS);
    ENSURE(c->cg_edges, (size_t)ER_REGIONS * er.cap_r * 8);
    er.e = c->cg_edges.as<u64>();
    er.cur = c->cg_cur.as<u32>();
    er.far = far;
    er.n_far = (u32)n_far;
    {
      ZeroList z;
      memset(&z, 0, sizeof z);
      z.p[0] = c->cg_bits.as<u32>(); z.n[0] = n_words;
      z.p[1] = c->cg_nbits.as<u32>(); z.n[1] = n_words;
      z.p[2] = c->cg_cur.as<u32>(); z.n[2] = ER_REGIONS * ER_STRIDE + 8;
      z.p[3] = (u32 *)&c->d_ctr[CTR_EDGES]; z.n[3] = 2 * (CTR_GOVER - CTR_EDGES + 1);
      hipLaunchKernelGGL(k_zero_many, dim3(64), dim3(256), 0, st, z);
    }
    for (u32 seg = 0; seg < nseg; seg++) {
      if (seg == 0) {
        if (c->kev_on) HIPCHK(hipEventRecord(c->kev[20], st));
        hipLaunchKernelGGL((k_pairs_append<true, WT>), dim3(blocks_for(U, PA_PPT * 256)), dim3(256), 0, st, g_word, (const u32 *)nullptr, U,
                           w_from<WT>(plan.mask[0]), d_masks, 0u, distance, walk_max, er, c->cg_bits.as<u32>(),
                           &c->d_ctr[CTR_BIGMASK], (u32 *)&c->d_ctr[CTR_EOVER]);
      } else {
        u32 *vs = c->seg_vs.as<u32>() + (size_t)(seg - 1) * U;
        WT *ws = c->seg_ws.as<WT>() + (size_t)(seg - 1) * U;
        // the count of words a padded grouping holds (pt_work: the next grouping overwrites it only behind this search,
        // in stream order); an order kept from an earlier attempt is complete, or that attempt would have been discarded
        seg_valid[seg] = nullptr;
        if (!ordered_seg[seg]) {
          TRY(bucket_order<WT>(c, plan, seg, g_word, U, ws, vs, true));
          seg_valid[seg] = c->gf_valid;
        }
        ordered_seg[seg] = true;
        if (seg < 8 && c->kev_on) HIPCHK(hipEventRecord(c->kev[20 + 2 * seg], st));
        hipLaunchKernelGGL((k_pairs_append<false, WT>), dim3(blocks_for(U, PA_PPT * 256)), dim3(256), 0, st, (const WT *)ws, (const u32 *)vs, U,
                           w_from<WT>(plan.mask[seg]), d_masks, seg, distance, walk_max, er, c->cg_bits.as<u32>(),
                           &c->d_ctr[CTR_BIGMASK], (u32 *)&c->d_ctr[CTR_EOVER], seg_valid[seg]);
      }
      if (seg < 8 && c->kev_on) HIPCHK(hipEventRecord(c->kev[21 + 2 * seg], st));
    }
    if (n_far)
      hipLaunchKernelGGL(k_mark_pairs, dim3(blocks_for(n_far)), dim3(256), 0, st, far, (u32)n_far, U, c->cg_bits.as<u32>(), bad);
    CgSource src;
    src.er = er; src.recs = nullptr; src.n_recs = 0; src.cnt_by_id = g_cnt; src.n_ids = U;
    src.pairs_bound = ecap + n_far;
    TRY(cg_build(c, src, method, cgs));
    er = src.er;
    if (cgs.group_over) {                                // a padded coarse bin of a grouping was full: words are missing
      c->gf_padded = false;                              // from a bucket order -- all of it again with exact bins
      for (u32 q = 0; q < MAX_COMBOS; q++) ordered_seg[q] = false;
      far = given ? ext_edges : nullptr;
      n_far = given ? n_ext_edges : 0;
      continue;
    }
    if (cgs.overflow) {                                  // a region was full: more room, once more
      c->cg_ecap = cgs.wanted + cgs.wanted / 2 + ER_REGIONS * 64;
      continue;
    }
    if (given) {
      u32 h_bad = 0;
      HIPCHK(hipMemcpyAsync(&h_bad, bad, 4, hipMemcpyDeviceToHost, st));
      HIPCHK(hipStreamSynchronize(st));
      if (h_bad) return fail(c, HUMID_E_INVALID, "malformed edge list (node index out of range)");
    }
    if (search && cgs.big_mask && !far) {
      // some bucket is longer than k_pairs_append walks: its remaining pairs (further apart than the
      // walk) come from the tiles, as one more region; then the search is taken again with them in place
      const u64 big_mask = cgs.big_mask;
      u64 got = 0;
      for (int phase = 0; phase < 2; phase++) {
        u64 at = 0;
        for (u32 seg = 0; seg < nseg; seg++) {
          if (!(big_mask >> seg & 1)) continue;
          const WT *W = seg ? c->seg_ws.as<WT>() + (size_t)(seg - 1) * U : g_word;
          const u32 *V = seg ? c->seg_vs.as<u32>() + (size_t)(seg - 1) * U : nullptr;
          std::vector<BigRun> runs;
          const BigRun *d_runs = nullptr;
          TRY(find_big_runs<WT>(c, W, U, w_from<WT>(plan.mask[seg]), walk_max, seg, runs, &d_runs));
          const ull tiles = runs.back().tile0;
          if (!tiles) continue;
          const u32 tgrid = (u32)std::min<ull>(tiles, 1u << 20);
          const ull start = phase ? at : 0;
          HIPCHK(hipMemcpyAsync(&c->d_ctr[CTR_SPECIAL], &start, sizeof(ull), hipMemcpyHostToDevice, st));
          HIPCHK(hipStreamSynchronize(st));
#define CG_TILES(P0, MD)                                                                                              \
  hipLaunchKernelGGL((k_pairs_tiles<P0, MD, WT>), dim3(tgrid), dim3(PT2_THREADS), 0, st, W, V, d_runs, (u32)runs.size() - 1, \
                     tiles, d_masks, seg, distance, walk_max, (u32 *)nullptr, (u32 *)nullptr, (const u32 *)nullptr,       \
                     (u32 *)nullptr, (u32 *)nullptr, (const u32 *)nullptr, c->cg_far.as<u64>(), &c->d_ctr[CTR_SPECIAL])
          if (phase == 0) { if (V) CG_TILES(false, PM_EMIT_COUNT); else CG_TILES(true, PM_EMIT_COUNT); }
          else { if (V) CG_TILES(false, PM_EMIT_FILL); else CG_TILES(true, PM_EMIT_FILL); }
#undef CG_TILES
          HIPCHK(hipGetLastError());
          TRY(read_counters(c));
          if (phase == 0) got += c->h_ctr[CTR_SPECIAL]; else at = c->h_ctr[CTR_SPECIAL];
        }
        if (phase == 0) {
          if (got > 0x7fffffffull) return fail(c, HUMID_E_OVERFLOW, "%llu neighbour pairs exceed the 32-bit adjacency offsets", (ull)got);
          ENSURE(c->cg_far, (size_t)(got + 1) * 8);
        }
      }
      far = c->cg_far.as<u64>();
      n_far = got;
      if (n_far) continue;                              // (nothing beyond the walk after all: the list stands)
    }
    E = cgs.E; M = cgs.M; Mbig = cgs.Mbig;
    // far too roomy for this input: the next pass gets what this one wanted + a quarter (launches are sized by it)
    if (search && 2 * (cgs.wanted + cgs.wanted / 4 + ER_REGIONS * 64) < c->cg_ecap) c->cg_ecap = cgs.wanted + cgs.wanted / 4 + ER_REGIONS * 64;
    break;
  }
  s.edges = c->E = E;
  s.nonsingle = c->M = M;
  c->cg_M = (u32)M;
  if (!c->lean_events) HIPCHK(hipEventRecord(c->ev[2], st));
  TRY(cg_cluster_rest(c, U, M, Mbig, method));
  const GraphArrays g = cg_arrays(c);
  const bool own = ((const void *)g_word == c->s_word.p) && U == (u32)c->U;
  if (!own) { ENSURE(c->cid, (size_t)U * 4); ENSURE(c->ismax, (size_t)U); }
  hipLaunchKernelGGL(k_finalize_leaves, dim3(blocks_for(U)), dim3(256), 0, st, BitRank{c->cg_bits.as<u32>(), c->cg_blk.as<u32>()},
                     BitRank{c->cg_nbits.as<u32>(), c->cg_nblk.as<u32>()}, c->cg_nodes.as<u32>(), (const u32 *)g.cl_of,
                     (const u32 *)g.maxleaf, U, own ? c->s_first.as<u32>() : (const u32 *)nullptr,
                     own ? c->s_slot.as<u32>() : (const u32 *)nullptr, own ? c->slot_out.as<u64>() : (u64 *)nullptr,
                     own ? (u32 *)nullptr : c->cid.as<u32>(), own ? (u8 *)nullptr : c->ismax.as<u8>());
  c->slots_done = own;
  c->cg_valid = true;
  HIPCHK(hipGetLastError());
  n_pair_segs_out = nseg < 8 ? nseg : 8;
  return HUMID_OK;
}

// clusters = unique words - graph nodes that created no cluster (the pass's last host wait)
static int n_clusters_compact(humid_ctx *c, u32 U, u64 *out) {
  TRY(read_counters(c, c->cg_nblk.as<u32>() + c->cg_nblocks));
  *out = (u64)U - (c->h_ctr[CTR_N - 1] & 0xffffffffull);
  return HUMID_OK;
}

// The per-unique-word view of a compact graph stage, for the accessors (humid_get_leaves / _adjacency /
// _clusters / _histogram): degrees, CSR rows in walk indices, cluster arrays, creator prefix sum, ids.
static int expand_compact(humid_ctx *c) {
  if (!c->cg_valid || c->cg_expanded) return HUMID_OK;
  hipStream_t st = c->stream;
  const u32 U = c->gU, M = c->cg_M;
  const u64 E = c->E;
  if (U == 0) { c->cg_expanded = true; return HUMID_OK; }
  ENSURE(c->deg, ((size_t)U + 1) * 4);
  ENSURE(c->nbr_off, ((size_t)U + 1) * 4);
  ENSURE(c->nbr_idx, (size_t)(2 * E + 1) * 4);
  ENSURE(c->cl_of, (size_t)U * 4);
  ENSURE(c->maxleaf, (size_t)U * 4);
  ENSURE(c->cl_size, (size_t)U * 8);
  ENSURE(c->flag, (size_t)U * 4);
  ENSURE(c->pos, ((size_t)U + 1) * 4);
  ENSURE(c->cid, (size_t)U * 4);
  ENSURE(c->ismax, (size_t)U);
  const BitRank br{c->cg_bits.as<u32>(), c->cg_blk.as<u32>()};
  hipLaunchKernelGGL(k_expand_leaves, dim3(blocks_for((u64)U + 1)), dim3(256), 0, st, br, c->cg_nodes.as<u32>(), c->cg_deg.as<u32>(),
                     c->cg_cl_of.as<u32>(), c->cg_maxleaf.as<u32>(), c->cg_cl_size.as<u64>(), c->g_cnt, U, c->deg.as<u32>(),
                     c->cl_of.as<u32>(), c->maxleaf.as<u32>(), c->cl_size.as<u64>());
  TRY(exscan_u32(c, c->deg.as<u32>(), c->nbr_off.as<u32>(), (u64)U + 1));
  if (M)
    hipLaunchKernelGGL(k_expand_rows, dim3(blocks_for(M)), dim3(256), 0, st, c->cg_nodes.as<u32>(), c->cg_off.as<u32>(),
                       c->cg_idx.as<u32>(), M, c->nbr_off.as<u32>(), c->nbr_idx.as<u32>());
  hipLaunchKernelGGL(k_creator_flags, dim3(blocks_for(U)), dim3(256), 0, st, c->cl_of.as<u32>(), U, c->flag.as<u32>());
  TRY(exscan_u32(c, c->flag.as<u32>(), c->pos.as<u32>(), U));
  hipLaunchKernelGGL(k_finalize_nodes, dim3(blocks_for(U)), dim3(256), 0, st, c->cl_of.as<u32>(), c->pos.as<u32>(),
                     c->maxleaf.as<u32>(), U, c->cid.as<u32>(), c->ismax.as<u8>(), (const u32 *)nullptr, (const u32 *)nullptr,
                     (u64 *)nullptr);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(st));
  c->cg_expanded = true;
  return HUMID_OK;
}

static ComboFields plan_fields(const ComboPlan &plan, u32 cb);
static int unique_edges(humid_ctx *c, const u64 *d_edges, u64 raw, u32 U, u64 *n_edges_out);

// ---- edit distance (-e): neighbour pairs under Levenshtein distance 2 .. 5 --------------------------
// (distance <= 1 is the Hamming search: equal lengths leave no room for a lone insertion.)
// Pigeonhole with shifts.  The plan of the Hamming search cuts the word into s segments and looks at
// every combination of k = s - d of them: d edits damage at most d segments, so some combination is
// untouched.  Untouched does not mean unmoved: between a deletion and an insertion the text is
// shifted by one position.  With at most one such pair (d <= 3) the untouched segments of a
// combination are, in order, unshifted / shifted by one / unshifted again; taking as "X" the word
// whose text reappears one position LATER in the other (the other has the insertion first), the
// shift is +1.  So for every combination and every pattern (a, b) --
// members [a, b) of the combination shifted, a == b: none -- the words' own segments (X) are joined
// with the segments read at the shifted positions (Y); candidates are verified by the dynamic
// programme (lev_band1).  Every unordered pair is found from one of its two sides;
// duplicates go away in a final sort + unique.
// d = 4, 5 allow TWO insertion/deletion pairs: every untouched member t of a combination then sits at
// an offset o_t in {-2 .. +2} of its own, and the offsets form a walk that starts and ends at 0 (equal
// lengths) with one unit step per insertion or deletion: |o_0| + sum |o_t - o_{t-1}| + |o_last| <= 4.
// All such offset vectors are joined (up to a global sign: the mirrored vector finds the same pairs
// from their other side); d <= 3 is the special case 0..0 1..1 0..0.  Verification: lev_band2.  Result: c->e_edges (ascending), *n_edges_out.
// part_rank / part_world: this caller's share of the joins (multi-GPU: every rank holds the whole
// unique array and runs every part_world-th join; the shares are gathered and made unique by
// humid_stage_unique_edges).  make_unique = false leaves the raw list in c->e_raw.
template <class WT>
static int edit_edges(humid_ctx *c, const WT *g_word, u32 U, u32 word_nt, u32 distance, u64 *n_edges_out,
                      u32 part_rank = 0, u32 part_world = 1, bool make_unique = true) {
  hipStream_t st = c->stream;
  *n_edges_out = 0;
  if (U < 2) return HUMID_OK;
  const ComboPlan plan = make_plan(word_nt, distance, U, c->force_segments);
  const u32 kb = plan.key_bits ? plan.key_bits : 1;
  const bool k32 = kb <= 32;
  ENSURE(c->e_kx, (size_t)U * 8);
  ENSURE(c->e_vx, (size_t)U * 4);
  ENSURE(c->e_ky, (size_t)U * 8);
  ENSURE(c->e_vy, (size_t)U * 4);
  ENSURE(c->seg_k0, (size_t)U * 8);
  ENSURE(c->seg_v0, (size_t)U * 4);
  ENSURE(c->pc, ((size_t)U + 1) * 4);
  ENSURE(c->poff, ((size_t)U + 1) * 4);
  u64 raw = 0;                                   // pairs collected so far (with duplicates)
  u32 join_no = 0;
  auto sort_keys_of = [&](const ComboFields &cf, DBuf &kout, DBuf &vout) -> int {
    if (k32) {
      hipLaunchKernelGGL((k_combo_keys<u32, WT>), dim3(blocks_for(U)), dim3(256), 0, st, g_word, U, cf,
                         c->seg_k0.as<u32>(), c->seg_v0.as<u32>());
      TRY(sort_pairs<u32, u32>(c, c->seg_k0.as<u32>(), kout.as<u32>(), c->seg_v0.as<u32>(), vout.as<u32>(), U, 0, kb));
    } else {
      hipLaunchKernelGGL((k_combo_keys<u64, WT>), dim3(blocks_for(U)), dim3(256), 0, st, g_word, U, cf,
                         c->seg_k0.as<u64>(), c->seg_v0.as<u32>());
      TRY(sort_pairs<u64, u32>(c, c->seg_k0.as<u64>(), kout.as<u64>(), c->seg_v0.as<u32>(), vout.as<u32>(), U, 0, kb));
    }
    return HUMID_OK;
  };
  for (u32 cb = 0; cb < plan.ncombo; cb++) {
    const ComboFields cfx = plan_fields(plan, cb);
    TRY(sort_keys_of(cfx, c->e_kx, c->e_vx));
    const u32 k = cfx.nf;
    // offset vectors o[0 .. k) in [-D, D], D = d / 2, walk cost <= 2 D, first non-zero entry positive
    const int D = (int)(distance / 2);
    std::vector<std::vector<int>> patterns;
    {
      std::vector<int> o(k, 0);
      std::function<void(u32, int, bool)> rec = [&](u32 t, int cost, bool signed_yet) {
        if (t == k) {
          const int total = cost + (k ? (o[k - 1] < 0 ? -o[k - 1] : o[k - 1]) : 0);
          if (total <= 2 * D) patterns.push_back(o);
          return;
        }
        for (int v = -D; v <= D; v++) {
          if (!signed_yet && v < 0) continue;                 // canonical sign
          const int prev = t ? o[t - 1] : 0;
          const int step = v > prev ? v - prev : prev - v;
          if (cost + step > 2 * D) continue;
          o[t] = v;
          rec(t + 1, cost + step, signed_yet || v != 0);
        }
      };
      rec(0, 0, false);
    }
    for (const std::vector<int> &o : patterns) {
      {
        ComboFields cfy = cfx;
        bool valid = true, shifted = false;
        for (u32 t = 0; t < k; t++) {
          // offset +1 = one nucleotide towards the end of the word = a field shift lower by 2 bits
          const int sh = (int)cfy.shift[t] - 2 * o[t];
          if (sh < 0 || sh + (int)cfy.width[t] > (int)(2 * word_nt)) { valid = false; break; }   // off the word
          cfy.shift[t] = (u8)sh;
          shifted = shifted || o[t] != 0;
        }
        if (!valid) continue;
        const u32 a = 0, b = shifted ? 1u : 0u;                // (a != b: Y keys differ from X keys)
        if (join_no++ % part_world != part_rank) continue;     // another rank's join
        const void *ky = c->e_kx.p;
        const u32 *vy = c->e_vx.as<u32>();
        if (a != b) {
          TRY(sort_keys_of(cfy, c->e_ky, c->e_vy));
          ky = c->e_ky.p;
          vy = c->e_vy.as<u32>();
        }
        HIPCHK(hipMemsetAsync(c->pc.as<u32>() + U, 0, 4, st));
        HIPCHK(hipMemsetAsync(&c->d_ctr[CTR_BIGMASK], 0, sizeof(ull), st));
        const u32 jwalk = c->walk_max;                         // candidates one lane verifies for one entry (0: all)
#define EDIT_JOIN(FILL, KT, BAND, PC, POFF, OUT)                                                              \
  hipLaunchKernelGGL((k_edit_join<FILL, KT, WT, BAND>), dim3(blocks_for(U)), dim3(256), 0, st, c->e_kx.as<KT>(), \
                     c->e_vx.as<u32>(), (const KT *)ky, vy, U, g_word, word_nt, distance, PC, POFF, OUT, jwalk, &c->d_ctr[CTR_BIGMASK])
#define EDIT_CHUNKS(FILL, KT, BAND, NP, PC, POFF, OUT)                                                                    \
  hipLaunchKernelGGL((k_edit_join_chunks<FILL, KT, WT, BAND>), dim3(blocks_for(NP)), dim3(256), 0, st, c->e_kx.as<KT>(),  \
                     c->e_vx.as<u32>(), (const KT *)ky, vy, U, (const u32 *)c->e_runlo.as<u32>(), (const u32 *)c->e_choff.as<u32>(), \
                     (u32)(NP), jwalk, g_word, word_nt, distance, PC, POFF, OUT)
        if (k32) { if (D <= 1) EDIT_JOIN(false, u32, 1, c->pc.as<u32>(), (const u32 *)nullptr, (u64 *)nullptr);
                   else EDIT_JOIN(false, u32, 2, c->pc.as<u32>(), (const u32 *)nullptr, (u64 *)nullptr); }
        else { if (D <= 1) EDIT_JOIN(false, u64, 1, c->pc.as<u32>(), (const u32 *)nullptr, (u64 *)nullptr);
               else EDIT_JOIN(false, u64, 2, c->pc.as<u32>(), (const u32 *)nullptr, (u64 *)nullptr); }
        TRY(exscan_u32(c, c->pc.as<u32>(), c->poff.as<u32>(), (u64)U + 1));
        HIPCHK(hipGetLastError());
        TRY(read_counters(c, c->poff.as<u32>() + U));
        u64 found = c->h_ctr[CTR_N - 1] & 0xffffffffull;
        u64 n_pieces = 0;                                       // > 0: this join goes through the pieces
        if (c->h_ctr[CTR_BIGMASK]) {
          // some run of equal keys is longer than one lane walks: every run in pieces of jwalk candidates
          ENSURE(c->e_runlo, ((size_t)U + 1) * 4);
          ENSURE(c->e_nch, ((size_t)U + 1) * 4);
          ENSURE(c->e_choff, ((size_t)U + 1) * 4);
          if (k32) hipLaunchKernelGGL(k_edit_chunks<u32>, dim3(blocks_for((u64)U + 1)), dim3(256), 0, st, c->e_kx.as<u32>(), (const u32 *)ky, U,
                                      jwalk, c->e_runlo.as<u32>(), c->e_nch.as<u32>());
          else hipLaunchKernelGGL(k_edit_chunks<u64>, dim3(blocks_for((u64)U + 1)), dim3(256), 0, st, c->e_kx.as<u64>(), (const u64 *)ky, U,
                                  jwalk, c->e_runlo.as<u32>(), c->e_nch.as<u32>());
          TRY(exscan_u32(c, c->e_nch.as<u32>(), c->e_choff.as<u32>(), (u64)U + 1));
          HIPCHK(hipGetLastError());
          TRY(read_counters(c, c->e_choff.as<u32>() + U));
          n_pieces = c->h_ctr[CTR_N - 1] & 0xffffffffull;
          if (n_pieces >= 0xfffffff0ull) return fail(c, HUMID_E_OVERFLOW, "too many candidate pieces in the edit-distance search");
          ENSURE(c->e_pc2, ((size_t)n_pieces + 1) * 4);
          ENSURE(c->e_poff2, ((size_t)n_pieces + 1) * 4);
          HIPCHK(hipMemsetAsync(c->e_pc2.as<u32>() + n_pieces, 0, 4, st));
          if (k32) { if (D <= 1) EDIT_CHUNKS(false, u32, 1, n_pieces, c->e_pc2.as<u32>(), (const u32 *)nullptr, (u64 *)nullptr);
                     else EDIT_CHUNKS(false, u32, 2, n_pieces, c->e_pc2.as<u32>(), (const u32 *)nullptr, (u64 *)nullptr); }
          else { if (D <= 1) EDIT_CHUNKS(false, u64, 1, n_pieces, c->e_pc2.as<u32>(), (const u32 *)nullptr, (u64 *)nullptr);
                 else EDIT_CHUNKS(false, u64, 2, n_pieces, c->e_pc2.as<u32>(), (const u32 *)nullptr, (u64 *)nullptr); }
          TRY(exscan_u32(c, c->e_pc2.as<u32>(), c->e_poff2.as<u32>(), n_pieces + 1));
          HIPCHK(hipGetLastError());
          TRY(read_counters(c, c->e_poff2.as<u32>() + n_pieces));
          found = c->h_ctr[CTR_N - 1] & 0xffffffffull;
        }
        if (found == 0) continue;
        if (raw + found >= 0x7fffffffull) return fail(c, HUMID_E_OVERFLOW, "too many candidate pairs in the edit-distance search");
        if ((raw + found) * 8 > c->e_raw.cap) {               // grow, keeping what is there
          DBuf bigger;
          HIPCHK(bigger.ensure((size_t)((raw + found) * 8 * 2)));
          if (raw) HIPCHK(hipMemcpyAsync(bigger.p, c->e_raw.p, (size_t)raw * 8, hipMemcpyDeviceToDevice, st));
          HIPCHK(hipStreamSynchronize(st));
          c->e_raw.release();
          c->e_raw = bigger;
        }
        if (n_pieces) {
          if (k32) { if (D <= 1) EDIT_CHUNKS(true, u32, 1, n_pieces, (u32 *)nullptr, (const u32 *)c->e_poff2.as<u32>(), c->e_raw.as<u64>() + raw);
                     else EDIT_CHUNKS(true, u32, 2, n_pieces, (u32 *)nullptr, (const u32 *)c->e_poff2.as<u32>(), c->e_raw.as<u64>() + raw); }
          else { if (D <= 1) EDIT_CHUNKS(true, u64, 1, n_pieces, (u32 *)nullptr, (const u32 *)c->e_poff2.as<u32>(), c->e_raw.as<u64>() + raw);
                 else EDIT_CHUNKS(true, u64, 2, n_pieces, (u32 *)nullptr, (const u32 *)c->e_poff2.as<u32>(), c->e_raw.as<u64>() + raw); }
        } else
        if (k32) { if (D <= 1) EDIT_JOIN(true, u32, 1, (u32 *)nullptr, c->poff.as<u32>(), c->e_raw.as<u64>() + raw);
                   else EDIT_JOIN(true, u32, 2, (u32 *)nullptr, c->poff.as<u32>(), c->e_raw.as<u64>() + raw); }
        else { if (D <= 1) EDIT_JOIN(true, u64, 1, (u32 *)nullptr, c->poff.as<u32>(), c->e_raw.as<u64>() + raw);
               else EDIT_JOIN(true, u64, 2, (u32 *)nullptr, c->poff.as<u32>(), c->e_raw.as<u64>() + raw); }
#undef EDIT_CHUNKS
#undef EDIT_JOIN
        raw += found;
      }
    }
  }
  HIPCHK(hipGetLastError());
  if (raw == 0) return HUMID_OK;
  if (!make_unique) { *n_edges_out = raw; return HUMID_OK; }
  TRY(unique_edges(c, c->e_raw.as<u64>(), raw, U, n_edges_out));
  return HUMID_OK;
}

// sorted, duplicate-free copy of an edge list (smaller << 32 | larger) -> c->e_edges
static int unique_edges(humid_ctx *c, const u64 *d_edges, u64 raw, u32 U, u64 *n_edges_out) {
  hipStream_t st = c->stream;
  *n_edges_out = 0;
  if (raw == 0) return HUMID_OK;
  if (raw >= 0x7fffffffull) return fail(c, HUMID_E_OVERFLOW, "too many edges");
  // ---- sort + unique ----
  const u32 R = (u32)raw;
  ENSURE(c->e_sorted, (size_t)R * 8);
  ENSURE(c->e_head, ((size_t)R + 1) * 4);
  ENSURE(c->e_hpos, ((size_t)R + 1) * 4);
  TRY(sort_keys<u64>(c, d_edges, c->e_sorted.as<u64>(), R, 0, 32 + bits_for(U)));
  hipLaunchKernelGGL(k_heads_u64, dim3(blocks_for((u64)R + 1)), dim3(256), 0, st, c->e_sorted.as<u64>(), R,
                     c->e_head.as<u32>());
  TRY(exscan_u32(c, c->e_head.as<u32>(), c->e_hpos.as<u32>(), (u64)R + 1));
  HIPCHK(hipGetLastError());
  TRY(read_counters(c, c->e_hpos.as<u32>() + R));
  const u64 E = c->h_ctr[CTR_N - 1] & 0xffffffffull;
  ENSURE(c->e_edges, (size_t)(E + 1) * 8);
  hipLaunchKernelGGL(k_compact_heads_u64, dim3(blocks_for(R)), dim3(256), 0, st, c->e_sorted.as<u64>(),
                     c->e_head.as<u32>(), c->e_hpos.as<u32>(), R, c->e_edges.as<u64>());
  HIPCHK(hipGetLastError());
  *n_edges_out = E;
  return HUMID_OK;
}

// ---- multi-GPU: this rank's share of the neighbour search ----------------------------------
// Every rank holds the whole ascending unique array.  Rank r of P looks for the pairs whose
// first element lies in its slice: for the prefix combo the r-th P-th of the positions, for a
// sorted combo the words whose combo key falls into the r-th P-th of the key space (a bucket is
// never split).  The union over ranks is every pair exactly once; pairs come out as
// (smaller rank << 32 | larger rank), unordered.
static int stage_pairs_share(humid_ctx *c, const u64 *g_word, u32 U, u32 word_nt, u32 distance,
                             u32 part_rank, u32 part_world, u64 *n_edges_out) {
  hipStream_t st = c->stream;
  *n_edges_out = 0;
  if (distance == 0 || U < 2) return HUMID_OK;
  const ComboPlan plan = make_plan(word_nt, distance, U, c->force_segments);
  EarlierMasksT<u64> d_masks;
  for (u32 t = 0; t < MAX_COMBOS; t++) d_masks.m[t] = plan.mask[t].lo;
  auto fields_of = [&](u32 cb) {
    ComboFields cf;
    cf.nf = plan.nfield[cb];
    for (u32 f = 0; f < MAX_FIELDS; f++) { cf.shift[f] = plan.shift[cb][f]; cf.width[f] = plan.width[cb][f]; }
    return cf;
  };
  const u32 nseg = plan.ncombo;
  const u32 kb = plan.key_bits ? plan.key_bits : 1;
  // share of the prefix combo: an equal slice of the positions
  const u32 p_lo = (u32)((u64)U * part_rank / part_world), p_hi = (u32)((u64)U * (part_rank + 1) / part_world);
  std::vector<u32> n_sel(nseg, 0);
  n_sel[0] = p_hi - p_lo;
  if (nseg > 1) {
    ENSURE(c->seg_k0, (size_t)U * 8);
    ENSURE(c->seg_v0, (size_t)U * 4);
    ENSURE(c->seg_ks, (size_t)U * 8);
    ENSURE(c->seg_vs, (size_t)(nseg - 1) * U * 4);
    ENSURE(c->seg_ws, (size_t)(nseg - 1) * U * 8);
  }
  // key range of this rank: [floor(r 2^kb / P), floor((r+1) 2^kb / P) - 1]
  const unsigned __int128 span = (unsigned __int128)1 << kb;
  const u64 klo = (u64)(span * part_rank / part_world);
  const u64 khi = (u64)(span * (part_rank + 1) / part_world - 1);
  for (u32 seg = 1; seg < nseg; seg++) {
    u32 *vs = c->seg_vs.as<u32>() + (size_t)(seg - 1) * U;
    HIPCHK(hipMemsetAsync(&c->d_ctr[CTR_SPECIAL], 0, sizeof(ull), st));
    if (kb <= 32)
      hipLaunchKernelGGL(k_select_keyrange<u32>, dim3(COMPACT_BLOCKS), dim3(256), 0, st, g_word, U, fields_of(seg),
                         klo, khi, c->seg_k0.as<u32>(), c->seg_v0.as<u32>(), c->d_ctr);
    else
      hipLaunchKernelGGL(k_select_keyrange<u64>, dim3(COMPACT_BLOCKS), dim3(256), 0, st, g_word, U, fields_of(seg),
                         klo, khi, c->seg_k0.as<u64>(), c->seg_v0.as<u32>(), c->d_ctr);
    HIPCHK(hipGetLastError());
    TRY(read_counters(c));
    n_sel[seg] = (u32)c->h_ctr[CTR_SPECIAL];
    if (n_sel[seg] > 1) {
      if (kb <= 32) TRY(sort_pairs<u32, u32>(c, c->seg_k0.as<u32>(), c->seg_ks.as<u32>(), c->seg_v0.as<u32>(), vs, n_sel[seg], 0, kb));
      else TRY(sort_pairs<u64, u32>(c, c->seg_k0.as<u64>(), c->seg_ks.as<u64>(), c->seg_v0.as<u32>(), vs, n_sel[seg], 0, kb));
    } else if (n_sel[seg] == 1) {
      HIPCHK(hipMemcpyAsync(vs, c->seg_v0.p, 4, hipMemcpyDeviceToDevice, st));
    }
    if (n_sel[seg])
      hipLaunchKernelGGL(k_gather_bucket_words<u64>, dim3(blocks_for(n_sel[seg])), dim3(256), 0, st, g_word, vs,
                         n_sel[seg], c->seg_ws.as<u64>() + (size_t)(seg - 1) * U);
  }
  u64 T = 0;
  std::vector<u64> base(nseg, 0);
  for (u32 seg = 0; seg < nseg; seg++) { base[seg] = T; T += n_sel[seg]; }
  if (T == 0) return HUMID_OK;
  if (T + 1 >= 0xffffffffull) return fail(c, HUMID_E_OVERFLOW, "too many positions in one share");
  ENSURE(c->pc, (size_t)(T + 1) * 4);
  ENSURE(c->poff, (size_t)(T + 1) * 4);
  HIPCHK(hipMemsetAsync(c->pc.as<u32>() + T, 0, 4, st));
  HIPCHK(hipMemsetAsync(&c->d_ctr[CTR_BIGMASK], 0, sizeof(ull), st));
  // the walk of a position is bounded as on one GPU (round 3: a bucket of 10^5 words made a lane walk it all);
  // what lies beyond it inside large buckets is finished by the tiles below
  const u32 walk_max = c->walk_max;
  u64 E_near = 0, E_far = 0;
  std::vector<std::vector<BigRun>> runs(nseg);
  std::vector<const BigRun *> d_runs(nseg, nullptr);
  u64 big_mask = 0;
  // what the tiles of combination `seg` walk: the whole array, first positions in this rank's slice (prefix
  // combination), or this rank's selected words (the others)
  auto tile_launch = [&](u32 seg, int mode) -> int {
    const ull tiles = runs[seg].back().tile0;
    if (!tiles) return HUMID_OK;
    const u32 tgrid = (u32)std::min<ull>(tiles, 1u << 20);
    const u32 *vs = seg ? c->seg_vs.as<u32>() + (size_t)(seg - 1) * U : nullptr;
    const u64 *ws = seg ? c->seg_ws.as<u64>() + (size_t)(seg - 1) * U : g_word;
    const u32 lo = seg ? 0u : p_lo, hi = seg ? 0xffffffffu : p_hi;
#define SHARE_TILES(P0, MD)                                                                                                     \
  hipLaunchKernelGGL((k_pairs_tiles<P0, MD, u64>), dim3(tgrid), dim3(PT2_THREADS), 0, st, ws, vs, d_runs[seg],                    \
                     (u32)runs[seg].size() - 1, tiles, d_masks, seg, distance, walk_max, (u32 *)nullptr, (u32 *)nullptr,          \
                     (const u32 *)nullptr, (u32 *)nullptr, (u32 *)nullptr, (const u32 *)nullptr, c->share_edges.as<u64>(),       \
                     &c->d_ctr[CTR_SPECIAL], lo, hi)
    if (seg == 0 && mode == PM_EMIT_COUNT) SHARE_TILES(true, PM_EMIT_COUNT);
    else if (seg == 0) SHARE_TILES(true, PM_EMIT_FILL);
    else if (mode == PM_EMIT_COUNT) SHARE_TILES(false, PM_EMIT_COUNT);
    else SHARE_TILES(false, PM_EMIT_FILL);
#undef SHARE_TILES
    HIPCHK(hipGetLastError());
    return HUMID_OK;
  };
  for (int phase = 0; phase < 2; phase++) {
    for (u32 seg = 0; seg < nseg; seg++) {
      if (n_sel[seg] == 0) continue;
      u32 *pcs = c->pc.as<u32>() + base[seg];
      const u32 *pos = c->poff.as<u32>() + base[seg];
      const u32 *vs = seg ? c->seg_vs.as<u32>() + (size_t)(seg - 1) * U : nullptr;
      const u64 *ws = seg ? c->seg_ws.as<u64>() + (size_t)(seg - 1) * U : g_word;
      u64 *ed = c->share_edges.as<u64>();
      const dim3 grid(blocks_for(n_sel[seg])), blk(256);
      if (seg == 0 && phase == 0)
        hipLaunchKernelGGL((k_pairs<true, PM_EMIT_COUNT, u64>), grid, blk, 0, st, g_word, vs, U, p_lo, n_sel[0], plan.mask[0].lo,
                           d_masks, 0u, distance, (u32 *)nullptr, (u32 *)nullptr, (const u32 *)nullptr,
                           (u32 *)nullptr, (u32 *)nullptr, pcs, pos, ed, (u32 *)nullptr, walk_max, &c->d_ctr[CTR_BIGMASK]);
      else if (seg == 0)
        hipLaunchKernelGGL((k_pairs<true, PM_EMIT_FILL, u64>), grid, blk, 0, st, g_word, vs, U, p_lo, n_sel[0], plan.mask[0].lo,
                           d_masks, 0u, distance, (u32 *)nullptr, (u32 *)nullptr, (const u32 *)nullptr,
                           (u32 *)nullptr, (u32 *)nullptr, pcs, pos, ed, (u32 *)nullptr, walk_max);
      else if (phase == 0)
        hipLaunchKernelGGL((k_pairs<false, PM_EMIT_COUNT, u64>), grid, blk, 0, st, ws, vs, n_sel[seg], 0u, n_sel[seg],
                           plan.mask[seg].lo, d_masks, seg, distance, (u32 *)nullptr, (u32 *)nullptr,
                           (const u32 *)nullptr, (u32 *)nullptr, (u32 *)nullptr, pcs, pos, ed, (u32 *)nullptr, walk_max,
                           &c->d_ctr[CTR_BIGMASK]);
      else
        hipLaunchKernelGGL((k_pairs<false, PM_EMIT_FILL, u64>), grid, blk, 0, st, ws, vs, n_sel[seg], 0u, n_sel[seg],
                           plan.mask[seg].lo, d_masks, seg, distance, (u32 *)nullptr, (u32 *)nullptr,
                           (const u32 *)nullptr, (u32 *)nullptr, (u32 *)nullptr, pcs, pos, ed, (u32 *)nullptr, walk_max);
    }
    if (phase == 0) {
      TRY(exscan_u32(c, c->pc.as<u32>(), c->poff.as<u32>(), T + 1));
      HIPCHK(hipGetLastError());
      TRY(read_counters(c, c->poff.as<u32>() + T));
      E_near = c->h_ctr[CTR_N - 1] & 0xffffffffull;
      big_mask = c->h_ctr[CTR_BIGMASK];
      if (big_mask) {
        HIPCHK(hipMemsetAsync(&c->d_ctr[CTR_SPECIAL], 0, sizeof(ull), st));
        for (u32 seg = 0; seg < nseg; seg++) {
          if (!(big_mask >> seg & 1) || n_sel[seg] == 0) continue;
          const u64 *ws = seg ? c->seg_ws.as<u64>() + (size_t)(seg - 1) * U : g_word;
          TRY(find_big_runs<u64>(c, ws, seg ? n_sel[seg] : U, plan.mask[seg].lo, walk_max, seg, runs[seg], &d_runs[seg], U));
          TRY(tile_launch(seg, PM_EMIT_COUNT));
        }
        TRY(read_counters(c));
        E_far = c->h_ctr[CTR_SPECIAL];
      }
      if (E_near + E_far > 0xffffffffull) return fail(c, HUMID_E_OVERFLOW, "%llu neighbour pairs in one share", (ull)(E_near + E_far));
      *n_edges_out = E_near + E_far;
      if (E_near + E_far == 0) return HUMID_OK;
      ENSURE(c->share_edges, (size_t)(E_near + E_far) * 8);
      if (E_near == 0) break;                          // (only far pairs: no fill launches of k_pairs)
    }
  }
  if (E_far) {
    const ull at = E_near;                             // the tiles append behind k_pairs' pairs
    HIPCHK(hipMemcpyAsync(&c->d_ctr[CTR_SPECIAL], &at, sizeof(ull), hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));                  // (`at` is a host temporary)
    for (u32 seg = 0; seg < nseg; seg++)
      if ((big_mask >> seg & 1) && n_sel[seg] && runs[seg].size() > 1) TRY(tile_launch(seg, PM_EMIT_FILL));
  }
  HIPCHK(hipGetLastError());
  return HUMID_OK;
}

// The un-permute in two coalesced passes (kernels_part.hip.h): position i of the partition order
// (pk_vals = read, pslot = padded slot of its word) -> cluster_id / keep in read order, or, packed,
// cluster id | keep << 31 per read.  *done = false: the read set is too large for the bin table
// (more than 2048 windows of 32 K reads) or the option is off; the caller takes the one-kernel form.
// ev_mid is recorded between the two kernels.
static int unpermute_tiled(humid_ctx *c, u32 N, bool packed, u32 *d_cid, u8 *d_keep, hipEvent_t ev_mid, bool *done) {
  hipStream_t st = c->stream;
  *done = false;
  if (!c->use_tile_partition || N == 0) return HUMID_OK;
  static const u32 uw_pref = getenv("HUMID_UW_SHIFT") ? (u32)atoi(getenv("HUMID_UW_SHIFT")) : 14u;   // (experiments)
  u32 wshift = uw_pref == 15 ? 15u : 14u;
  if (((u64)N + (1u << wshift) - 1) >> wshift > UW_MAXBINS) wshift = UW_MAXSHIFT;
  const u32 n_bins = (u32)(((u64)N + (1u << wshift) - 1) >> wshift);
  if (n_bins > UW_MAXBINS) return HUMID_OK;
  ENSURE(c->unperm_rec, ((size_t)n_bins << wshift) * 8 + (size_t)UW_MAXBINS * 4);
  u64 *rec = c->unperm_rec.as<u64>();
  u32 *ucur = (u32 *)(rec + ((size_t)n_bins << wshift));
  // the bins' cursors: k_unperm_window leaves every cursor it read at zero, so only a new place needs a clear
  if (c->ucur_clean != ucur) HIPCHK(hipMemsetAsync(ucur, 0, (size_t)UW_MAXBINS * 4, st));
  c->ucur_clean = nullptr;
  // positions in use: all N for the sorted (wide-word) count, else up to pbeg[n_parts] (on the device)
  const bool bucketed = c->n_parts && !c->last_count_sorted;
  const u32 *n_pos_dev = bucketed ? c->pbeg.as<u32>() + c->n_parts : (const u32 *)nullptr;
  if (c->last_rec8) {
    // buckets per workgroup: about 7/8 of a tile's worth of records (reads per bucket: usable / buckets)
    const u64 mean = std::max<u64>(1, c->usable / c->n_parts);
    const u32 B = (u32)std::min<u64>(64, std::max<u64>(1, (PT_TILE - PT_TILE / 8) / mean));
    const u32 grid = (c->n_parts + B - 1) / B;
    if (n_bins <= 1024)
      hipLaunchKernelGGL(k_unperm_bins8<1024>, dim3(grid), dim3(1024), 0, st, (const u64 *)c->p8_b.as<u64>(), c->rec_cursor2,
                         (const u64 *)c->slot_out.as<u64>(), c->n_parts, B, N, wshift, n_bins, ucur, rec);
    else
      hipLaunchKernelGGL(k_unperm_bins8<2048>, dim3(grid), dim3(1024), 0, st, (const u64 *)c->p8_b.as<u64>(), c->rec_cursor2,
                         (const u64 *)c->slot_out.as<u64>(), c->n_parts, B, N, wshift, n_bins, ucur, rec);
  }
  else
  hipLaunchKernelGGL(k_unperm_bins, dim3((N + PT_TILE - 1) / PT_TILE), dim3(1024), 0, st, c->pk_vals.as<u32>(),
                     c->pslot.as<u32>(), c->slot_out.as<u64>(), n_pos_dev, N, N, wshift, n_bins, ucur, rec);
  if (!c->lean_events) HIPCHK(hipEventRecord(ev_mid, st));
  if (packed) {
    if (wshift == 14) hipLaunchKernelGGL((k_unperm_window<true, 14>), dim3(n_bins), dim3(UW_THREADS), 0, st, rec, ucur, N, d_cid, d_keep);
    else hipLaunchKernelGGL((k_unperm_window<true, 15>), dim3(n_bins), dim3(UW_THREADS), 0, st, rec, ucur, N, d_cid, d_keep);
  } else {
    if (wshift == 14) hipLaunchKernelGGL((k_unperm_window<false, 14>), dim3(n_bins), dim3(UW_THREADS), 0, st, rec, ucur, N, d_cid, d_keep);
    else hipLaunchKernelGGL((k_unperm_window<false, 15>), dim3(n_bins), dim3(UW_THREADS), 0, st, rec, ucur, N, d_cid, d_keep);
  }
  if (c->kev_on) HIPCHK(hipEventRecord(c->kev[41], st));
  HIPCHK(hipGetLastError());
  c->ucur_clean = ucur;
  *done = true;
  return HUMID_OK;
}

// ---- stage C: per-read outputs -------------------------------------------------------------
// l_cid/l_ismax: cluster id and maxLeaf flag of THIS context's unique words in local walk order
// (on one GPU the arrays stage B left behind; on several, this rank's slice of them).
static int stage_map(humid_ctx *c, const u32 *l_cid, const u8 *l_ismax, u32 N, u32 *d_cid, u8 *d_keep) {
  hipStream_t st = c->stream;
  const u32 U = (u32)c->U;
  const bool fused = c->slots_done && l_cid == c->cid.as<u32>() && l_ismax == c->ismax.as<u8>();
  c->slots_done = false;
  if (U > 0 && !fused)
    hipLaunchKernelGGL(k_slot_results, dim3(blocks_for(U)), dim3(256), 0, st, l_cid, l_ismax,
                       c->s_first.as<u32>(), c->s_slot.as<u32>(), U, c->slot_out.as<u64>());
  if (!c->lean_events) HIPCHK(hipEventRecord(c->ev[3], st));
  if (c->last_count_lds) {
    bool tiled = false;
    TRY(unpermute_tiled(c, N, false, d_cid, d_keep, c->kev[36], &tiled));
    c->last_unperm_tiled = tiled;
    if (!tiled) {
      // round-1 form: one scattered 4-byte store per read, then a coalesced split.  pk_keys (the
      // partitioned keys) is dead by now: reuse it for the packed per-read results; reads that were
      // excluded from the partition are not in it, so the array starts as zeros
      u32 *packed = c->pk_keys.as<u32>();
      HIPCHK(hipMemsetAsync(packed, 0, (size_t)N * 4, st));
      if (c->n_parts && !c->last_count_sorted)
        hipLaunchKernelGGL(k_read_map_bucket, dim3(c->n_parts), dim3(256), 0, st, c->pk_vals.as<u32>(),
                           c->pslot.as<u32>(), c->slot_out.as<u64>(), c->pbeg.as<u32>(), c->ucount.as<u32>(), N, packed);
      else
        hipLaunchKernelGGL(k_read_map_part, dim3(grid_stride_blocks(N)), dim3(256), 0, st, c->pk_vals.as<u32>(),
                           c->pslot.as<u32>(), c->slot_out.as<u64>(), N, packed);
      if (c->kev_on) HIPCHK(hipEventRecord(c->kev[36], st));
      hipLaunchKernelGGL(k_split_out, dim3(grid_stride_blocks(N)), dim3(256), 0, st, packed, N, d_cid, d_keep);
    }
  } else
    hipLaunchKernelGGL(k_read_map, dim3(grid_stride_blocks(N)), dim3(256), 0, st, c->slot_of_read.as<u32>(),
                       c->slot_out.as<u64>(), N, d_cid, d_keep);
  HIPCHK(hipEventRecord(c->ev[4], st));
  HIPCHK(hipGetLastError());
  return HUMID_OK;
}

static int check_run_args(humid_ctx *c, u64 n_reads, u32 word_nt, u32 method, u32 max_nt = 32) {
  if (word_nt == 0) return fail(c, HUMID_E_INVALID, "word_nt must be >= 1");
  if (word_nt > max_nt) return fail(c, HUMID_E_UNSUPPORTED, "word_nt %u > %u is not supported by this entry point", word_nt, max_nt);
  if (method > 1) return fail(c, HUMID_E_INVALID, "method must be 0 (directional) or 1 (maximum)");
  if (n_reads > 0x7fffffffull) return fail(c, HUMID_E_OVERFLOW, "n_reads %llu exceeds 2^31-1", (ull)n_reads);
  return HUMID_OK;
}

// ---- the full pipeline on device buffers (one GPU) -------------------------------------------
// WT = u64: word_nt <= 32, one uint64 per read.  WT = W2: 33 <= word_nt <= 64, two per read.
template <class WT>
static int run_device(humid_ctx *c, const WT *d_words, const u8 *d_filt, u64 n_reads, u32 word_nt,
                      u32 distance, u32 method, u32 *d_cid, u8 *d_keep, humid_summary *sum) {
  if (!c) return HUMID_E_INVALID;
  constexpr bool WIDE = sizeof(WT) == 16;
  c->have_run = false;
  c->graph_mode = false;
  c->have_graph = false;
  c->dense_mode = false;
  TRY(check_run_args(c, n_reads, word_nt, method, 64));
  if (WIDE != (word_nt > 32)) return fail(c, HUMID_E_INVALID, "word layout does not match word_nt");
  if (WIDE && ((uintptr_t)d_words & 15)) return fail(c, HUMID_E_INVALID, "wide words must be 16-byte aligned on the device");
  if (n_reads && (!d_words || !d_filt || !d_cid || !d_keep)) return fail(c, HUMID_E_INVALID, "null buffer");
  HIPCHK(hipSetDevice(c->device));
  hipStream_t st = c->stream;
  const u32 N = (u32)n_reads;
  humid_summary s;
  memset(&s, 0, sizeof s);
  s.total = n_reads;
  c->last_unperm_tiled = false;
  c->N = n_reads; c->U = c->E = c->M = c->C = c->usable = 0;
  c->word_nt = word_nt; c->distance = distance; c->method = method;
  c->gU = 0;
  if (N == 0) { if (sum) *sum = s; c->have_run = c->have_graph = true; return HUMID_OK; }
  // the stages' own events only with the per-kernel timing (ms_count .. ms_map are 0 without it; ms_total and the
  // count kernel's time are always measured)
  struct LeanEvents { humid_ctx *c; ~LeanEvents() { c->lean_events = false; } } lean_guard{c};
  c->lean_events = !c->kev_on && getenv("HUMID_ALL_EVENTS") == nullptr;
  if constexpr (WIDE) TRY(stage_count_wide(c, d_words, d_filt, N, word_nt, s));
  else TRY(stage_count(c, d_words, d_filt, N, word_nt, 0ull, ~0ull, 0, s));
  const u32 U = (u32)c->U;
  if (U == 0) {   // everything filtered
    HIPCHK(hipMemsetAsync(d_cid, 0, (size_t)N * 4, st));
    HIPCHK(hipMemsetAsync(d_keep, 0, (size_t)N, st));
    HIPCHK(hipStreamSynchronize(st));
    if (sum) *sum = s;
    c->have_run = c->have_graph = true;
    return HUMID_OK;
  }
  u32 n_pair_segs = 0;
  if (c->edit && distance >= 2) {
    // -e: Levenshtein neighbours (src/humid.cc:140-158); distance <= 1 IS the Hamming search
    if (distance > 5) return fail(c, HUMID_E_UNSUPPORTED, "edit distance %u > 5 is not supported", distance);
    u64 E = 0;
    TRY(edit_edges<WT>(c, c->s_word.as<WT>(), U, word_nt, distance, &E));
    static const u64 no_edges = 0;
    if (c->use_compact)
      TRY(stage_graph_compact<WT>(c, c->s_word.as<WT>(), c->s_cnt.as<u32>(), U, word_nt, distance, method, s, n_pair_segs,
                                  E ? c->e_edges.as<u64>() : &no_edges, E));
    else
      TRY(stage_graph<WT>(c, c->s_word.as<WT>(), c->s_cnt.as<u32>(), U, word_nt, distance, method, s, n_pair_segs,
                          E ? c->e_edges.as<u64>() : &no_edges, E));
  } else if (c->use_compact)
    TRY(stage_graph_compact<WT>(c, c->s_word.as<WT>(), c->s_cnt.as<u32>(), U, word_nt, distance, method, s, n_pair_segs));
  else
    TRY(stage_graph<WT>(c, c->s_word.as<WT>(), c->s_cnt.as<u32>(), U, word_nt, distance, method, s, n_pair_segs));
  TRY(stage_map(c, c->cid.as<u32>(), c->ismax.as<u8>(), N, d_cid, d_keep));
  if (c->cg_valid) TRY(n_clusters_compact(c, U, &c->C));
  else TRY(n_clusters_from_scan(c, U, &c->C));
  s.clusters = c->C;
  const u64 E = c->E, M = c->M;
  // (the last host wait watches a mapped flag, not the stream: the runtime may not have seen the last
  // event's signal yet -- "device not ready" from hipEventElapsedTime once in ~10^3 runs)
  HIPCHK(hipEventSynchronize(c->ev[4]));
  const bool lean = c->lean_events;
  c->lean_events = false;
  if (!lean) {                                               // the stages' shares: option kernel_timing
    HIPCHK(hipEventElapsedTime(&s.ms_count, c->ev[0], c->ev[1]));
    HIPCHK(hipEventElapsedTime(&s.ms_neighbours, c->ev[1], c->ev[2]));
    HIPCHK(hipEventElapsedTime(&s.ms_cluster, c->ev[2], c->ev[3]));
    HIPCHK(hipEventElapsedTime(&s.ms_map, c->ev[3], c->ev[4]));
  }
  HIPCHK(hipEventElapsedTime(&s.ms_total, c->ev[0], c->ev[4]));
  HIPCHK(hipEventElapsedTime(&s.ms_k_insert, c->kev[0], c->kev[1]));
  if (!c->kev_on) s.ms_k_map = s.ms_map;
  else if (c->last_count_lds) HIPCHK(hipEventElapsedTime(&s.ms_k_map, c->ev[3], c->kev[36]));   // first map kernel alone
  else s.ms_k_map = s.ms_map;   // ev[3]..ev[4] bracket exactly the k_read_map launch
  if (c->kev_on && c->last_count_lds && c->last_unperm_tiled) HIPCHK(hipEventElapsedTime(&s.ms_k_unperm, c->kev[36], c->kev[41]));
  if (c->kev_on && c->last_count_lds && c->last_part_tiled && !c->last_count_sorted) HIPCHK(hipEventElapsedTime(&s.ms_k_part, c->kev[39], c->kev[40]));
  s.count_mode_used = (c->last_count_sorted ? 3u : c->last_count_lds ? (c->last_count_ordered ? 2u : 0u) : 1u) | (c->last_rec8 ? 0x100u : 0u);
  if (c->kev_on) HIPCHK(hipEventElapsedTime(&s.ms_k_cluster, c->kev[2], c->kev[3]));
  for (u32 g = 0; c->kev_on && g < n_pair_segs; g++) {
    float t = 0;
    HIPCHK(hipEventElapsedTime(&t, c->kev[20 + 2 * g], c->kev[21 + 2 * g]));   // count phase
    s.ms_k_pairs += t;
    if (E > 0 && !c->cg_valid) {
      HIPCHK(hipEventElapsedTime(&t, c->kev[4 + 2 * g], c->kev[5 + 2 * g]));   // fill phase
      s.ms_k_pairs += t;
    }
  }
  if (sum) *sum = s;
  c->have_run = true;
  c->have_graph = true;
  return HUMID_OK;
}

// --------------------------------------------------------------------------------
// C ABI
// --------------------------------------------------------------------------------
extern "C" {

uint32_t humid_abi_version(void) { return HUMID_ABI_VERSION; }

int humid_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

const char *humid_last_error(const humid_ctx *ctx) { return ctx ? ctx->err.c_str() : g_err.c_str(); }

int humid_ctx_create(humid_ctx **out, int device, void *stream) {
  humid_ctx *c = nullptr;
  if (!out) return fail(nullptr, HUMID_E_INVALID, "out is null");
  *out = nullptr;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0)
    return fail(nullptr, HUMID_E_HIP, "no HIP device available (%s); this library has no CPU fallback",
                hipGetErrorString(e));
  if (device < 0) { if (hipGetDevice(&device) != hipSuccess) device = 0; }
  if (device >= ndev) return fail(nullptr, HUMID_E_INVALID, "device %d out of range (%d devices)", device, ndev);
  c = new (std::nothrow) humid_ctx();
  if (!c) return fail(nullptr, HUMID_E_NOMEM, "host allocation failed");
  c->device = device;
  auto bail = [&](hipError_t err, const char *what) {
    int rc = fail(nullptr, HUMID_E_HIP, "%s: %s", what, hipGetErrorString(err));
    humid_ctx_destroy(c);
    return rc;
  };
  if ((e = hipSetDevice(device)) != hipSuccess) return bail(e, "hipSetDevice");
  if (stream) { c->stream = (hipStream_t)stream; c->own_stream = false; }
  else {
    if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) return bail(e, "hipStreamCreate");
    c->own_stream = true;
  }
  if ((e = hipMalloc((void **)&c->d_ctr, CTR_N * sizeof(ull) + sizeof(PsChain))) != hipSuccess) return bail(e, "hipMalloc");
  c->ps_chain = (PsChain *)(c->d_ctr + CTR_N);                 // the scans' chain (prims.hip.h): zero once, epochs after that
  if ((e = hipMemset(c->ps_chain, 0, sizeof(PsChain))) != hipSuccess) return bail(e, "hipMemset");
  if ((e = hipHostMalloc((void **)&c->h_ctr, (CTR_N + 2) * sizeof(ull), hipHostMallocDefault)) != hipSuccess) return bail(e, "hipHostMalloc");
  memset(c->h_ctr, 0, (CTR_N + 2) * sizeof(ull));
  if (hipHostGetDevicePointer((void **)&c->h_ctr_dev, c->h_ctr, 0) != hipSuccess) { c->h_ctr_dev = nullptr; (void)hipGetLastError(); }
  c->no_poll = getenv("HUMID_NO_POLL") != nullptr;
  c->no_chain = getenv("HUMID_NO_SCAN_CHAIN") != nullptr;
  c->gf_padded = getenv("HUMID_NO_GROUP_PAD") == nullptr;
  for (auto &ev : c->ev)
    if ((e = hipEventCreate(&ev)) != hipSuccess) return bail(e, "hipEventCreate");
  for (auto &ev : c->kev)
    if ((e = hipEventCreate(&ev)) != hipSuccess) return bail(e, "hipEventCreate");
  if (const char *m = getenv("HUMID_COUNT_MODE")) c->count_mode = (atoi(m) == 1) ? 1 : 0;
  if (const char *m = getenv("HUMID_KERNEL_TIMING")) c->kev_on = atoi(m) != 0;
  *out = c;
  return HUMID_OK;
}

void humid_ctx_destroy(humid_ctx *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  DBuf *bufs[] = {&c->in_words, &c->in_filt, &c->in_bases, &c->out_cid, &c->out_keep, &c->table, &c->pk_keys, &c->pk_vals,
                  &c->pbeg, &c->ucount, &c->pusable, &c->ubase, &c->pad_word, &c->pad_cf, &c->pslot,
                  &c->opos, &c->own_packed, &c->owner, &c->owner_sorted, &c->perm, &c->small, &c->pc, &c->poff, &c->share_edges, &c->own_words, &c->heads, &c->had, &c->big_runs, &c->small_roots, &c->e_kx, &c->e_vx, &c->e_ky, &c->e_vy, &c->e_raw, &c->e_sorted, &c->e_edges, &c->e_head, &c->e_hpos, &c->e_runlo, &c->e_nch, &c->e_choff, &c->e_pc2, &c->e_poff2, &c->x_slot, &c->x_slot_s, &c->x_cnt, &c->x_cnts, &c->x_rec, &c->x_ncnt, &c->x_route, &c->x_creator, &c->x_base, &c->x_mark, &c->x_markcr, &c->x_scan, &c->x_lcid, &c->x_lismax, &c->x_items, &c->x_w, &c->x_id, &c->x_ids, &c->x_ends, &c->x_ends_s, &c->x_head, &c->x_hpos, &c->x_nodes, &c->x_cedges, &c->w_heads, &c->w_sorted, &c->w_head, &c->w_hpos, &c->w_start, &c->pt_work, &c->unperm_rec, &c->route_tiles, &c->xr_hist, &c->xr_recv, &c->xr_eloc, &c->xr_got, &c->xr_eall, &c->xr_ret, &c->xr_heads, &c->xr_send, &c->xr_zero,
                  &c->xo_gw, &c->xo_gc, &c->xo_regs, &c->xo_inv, &c->xo_send, &c->xo_int, &c->xo_cross, &c->xo_sel, &c->xo_selall, &c->xo_parent, &c->xo_flag, &c->xo_xroot, &c->xo_xcbits, &c->xo_xcblk,
                  &c->xo_xcid, &c->xo_xcall, &c->xo_ldeg, &c->xo_cnt, &c->pw_a, &c->pw_ai, &c->pw_b, &c->pw_bi, &c->gf_cur, &c->p8_a, &c->p8_b, &c->p8_cur, &c->p8_status, &c->cg_edges, &c->cg_cur, &c->cg_far, &c->cg_bits, &c->cg_nbits, &c->cg_blk, &c->cg_nblk, &c->cg_nodes, &c->cg_ncnt, &c->cg_deg,
                  &c->cg_off, &c->cg_idx, &c->cg_parent, &c->cg_csize, &c->cg_curs, &c->cg_cl_of, &c->cg_maxleaf, &c->cg_cl_size,
                  &c->slot_out, &c->slot_of_read, &c->uniq_slot, &c->uniq_word, &c->s_word, &c->s_slot,
                  &c->s_cnt, &c->s_first, &c->deg, &c->nbr_off, &c->nbr_idx, &c->seg_k0, &c->seg_ks,
                  &c->seg_v0, &c->seg_vs, &c->seg_ws, &c->csize, &c->cur, &c->parent, &c->mk0, &c->mk1, &c->cl_of,
                  &c->maxleaf, &c->cl_size, &c->flag, &c->pos, &c->cid, &c->ismax, &c->stk, &c->tmp,
                  &c->scratch};
  for (DBuf *b : bufs) b->release();
#ifdef HUMID_PHASE_CLOCKS                                    // (experiment builds only: common.hip.h)
  {
    static const char *names[PH_KERNELS] = {"k_dedup_rec", "k_p8_scatter1", "k_p8_scatter2", "k_unperm_bins8", "k_group_fine", "k_pairs_append", "k_unperm_window", "-"};
    ull h[PH_KERNELS][PH_MAX];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(humid_phase), sizeof h) == hipSuccess)
      for (int k = 0; k < PH_KERNELS; k++) {
        if (!h[k][0]) continue;
        fprintf(stderr, "[phase clocks] %-16s %6llu workgroups sampled; us per workgroup by phase:", names[k], h[k][0]);
        double tot = 0;
        for (int t = 1; t < PH_MAX; t++) { fprintf(stderr, " %.2f", (double)h[k][t] / (double)h[k][0] / 100.0); tot += (double)h[k][t]; }
        fprintf(stderr, " | sum %.2f\n", tot / (double)h[k][0] / 100.0);
      }
  }
#endif
  if (c->arena.base) (void)hipFree(c->arena.base);
  if (c->d_ctr) (void)hipFree(c->d_ctr);
  if (c->h_ctr) (void)hipHostFree(c->h_ctr);
  for (auto &ev : c->ev) if (ev) (void)hipEventDestroy(ev);
  for (auto &ev : c->kev) if (ev) (void)hipEventDestroy(ev);
  if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

void *humid_host_alloc(uint64_t bytes) {
  void *p = nullptr;
  if (bytes == 0 || hipHostMalloc(&p, (size_t)bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  return p;
}

void humid_host_free(void *p) {
  if (p) (void)hipHostFree(p);
}

int humid_ctx_reserve(humid_ctx *c, uint64_t n_reads, uint32_t word_nt) {
  if (!c) return fail(nullptr, HUMID_E_INVALID, "ctx is null");
  if (c->arena.base) return HUMID_OK;                       // one slab per context
  if (n_reads == 0) return HUMID_OK;
  HIPCHK(hipSetDevice(c->device));
  // what one run over n_reads reads carves (measured: 112 B per read at 24 nt, unique/reads <= 1
  // assumed worst; wide words: +24 B) plus the host entry point's staging (14 or 22 B per read)
  const size_t per_read = (word_nt > 32 ? 200 : 180) + (word_nt > 32 ? 22 : 14);   // (168 + the padded partition level, round 2)
  size_t want = (size_t)n_reads * per_read + ((size_t)64 << 20);
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && want > free_b / 2) want = free_b / 2;
  void *p = nullptr;
  if (hipMalloc(&p, want) != hipSuccess) { (void)hipGetLastError(); return HUMID_OK; }   // no slab: buffers are allocated one by one
  c->arena.base = (char *)p;
  c->arena.size = want;
  c->arena.used = 0;
  // the first launch of a process loads the library's code object (~1500 kernels with the sort /
  // scan instantiations: tens of milliseconds): pay that here, off the caller's critical path
  hipLaunchKernelGGL(k_iota, dim3(1), dim3(64), 0, c->stream, (u32 *)p, 64u);
  HIPCHK(hipStreamSynchronize(c->stream));
  return HUMID_OK;
}

int humid_ctx_set_option(humid_ctx *c, const char *key, int64_t value) {
  if (!c || !key) return fail(c, HUMID_E_INVALID, "null argument");
  if (strcmp(key, "count_mode") == 0) {
    if (value != 0 && value != 1) return fail(c, HUMID_E_INVALID, "count_mode must be 0 (LDS-partitioned) or 1 (global table)");
    c->count_mode = (int)value;
    return HUMID_OK;
  }
  if (strcmp(key, "count_order") == 0) {
    if (value < -1 || value > 1) return fail(c, HUMID_E_INVALID, "count_order must be -1 (auto), 0 or 1");
    c->count_order = (int)value;
    return HUMID_OK;
  }
  if (strcmp(key, "edit_distance") == 0) {
    c->edit = value != 0;
    return HUMID_OK;
  }
  if (strcmp(key, "tile_partition") == 0) {
    c->use_tile_partition = value != 0;
    return HUMID_OK;
  }
  if (strcmp(key, "coop_big") == 0) {
    c->coop_big = value != 0;
    return HUMID_OK;
  }
  if (strcmp(key, "kernel_timing") == 0) {
    c->kev_on = value != 0;
    return HUMID_OK;
  }
  if (strcmp(key, "test_fail_before_gather") == 0) {
    c->x_test_fail_after = (int)value;
    return HUMID_OK;
  }
  if (strcmp(key, "records8") == 0) {
    c->use_rec8 = value != 0;
    return HUMID_OK;
  }
  if (strcmp(key, "compact_graph") == 0) {
    c->use_compact = value != 0;
    return HUMID_OK;
  }
  if (strcmp(key, "group_buckets") == 0) {
    c->group_buckets = value != 0;
    return HUMID_OK;
  }
  if (strcmp(key, "padded_partition") == 0) {
    c->pt_padded = value != 0;
    return HUMID_OK;
  }
  if (strcmp(key, "force_comm") == 0) {
    c->force_comm = value != 0;
    return HUMID_OK;
  }
  if (strcmp(key, "bucket_walk") == 0) {
    if (value < 0 || value > (1 << 24)) return fail(c, HUMID_E_INVALID, "bucket_walk must be 0 (no limit) .. 2^24");
    c->walk_max = (u32)value;
    return HUMID_OK;
  }
  if (strcmp(key, "plan_segments") == 0) {
    if (value < 0 || value > 32) return fail(c, HUMID_E_INVALID, "plan_segments must be 0 (auto) .. 32");
    c->force_segments = (u32)value;
    return HUMID_OK;
  }
  return fail(c, HUMID_E_INVALID, "unknown option '%s'", key);
}

int humid_dedup_run_device(humid_ctx *c, const uint64_t *d_words, const uint8_t *d_filtered,
                           uint64_t n_reads, uint32_t word_nt, uint32_t distance, uint32_t method,
                           uint32_t *d_cluster_id, uint8_t *d_keep, humid_summary *summary) {
  if (!c) return fail(nullptr, HUMID_E_INVALID, "ctx is null");
  if (word_nt > 32)
    return run_device<W2>(c, (const W2 *)d_words, d_filtered, n_reads, word_nt, distance, method, d_cluster_id, d_keep, summary);
  return run_device<u64>(c, d_words, d_filtered, n_reads, word_nt, distance, method, d_cluster_id, d_keep, summary);
}

// host buffers in, host buffers out: words + flags, or (bases != null) the raw symbols, packed on the device
static int run_host(humid_ctx *c, const uint64_t *words, const uint8_t *filtered, const uint8_t *bases,
                    uint64_t n_reads, uint32_t word_nt, uint32_t distance, uint32_t method, uint32_t *cluster_id,
                    uint8_t *keep, humid_summary *summary) {
  if (!c) return fail(nullptr, HUMID_E_INVALID, "ctx is null");
  if (n_reads && (!(bases || (words && filtered)) || !cluster_id || !keep)) return fail(c, HUMID_E_INVALID, "null buffer");
  if (n_reads > 0x7fffffffull) return fail(c, HUMID_E_OVERFLOW, "n_reads %llu exceeds 2^31-1", (ull)n_reads);
  if (bases) TRY(check_run_args(c, n_reads, word_nt, method, 64));
  HIPCHK(hipSetDevice(c->device));
  hipStream_t st = c->stream;
  humid_summary s;
  memset(&s, 0, sizeof s);
  hipEvent_t e0 = c->ev[5];
  hipEvent_t e1, e2, e3;
  HIPCHK(hipEventCreate(&e1)); HIPCHK(hipEventCreate(&e2)); HIPCHK(hipEventCreate(&e3));
  size_t n = (size_t)n_reads;
  const size_t wbytes = word_nt > 32 ? 16 : 8;
  ENSURE(c->in_words, n * wbytes + 16);
  ENSURE(c->in_filt, n + 8);
  ENSURE(c->out_cid, n * 4 + 8);
  ENSURE(c->out_keep, n + 8);
  if (bases) ENSURE(c->in_bases, n * word_nt + 16);
  HIPCHK(hipEventRecord(e0, st));
  if (n && bases) {
    // device-side packing (makeWord, src/fastq.cc:146-161): the host only gathered the symbols
    HIPCHK(hipMemcpyAsync(c->in_bases.p, bases, n * word_nt, hipMemcpyHostToDevice, st));
    if (word_nt > 32)
      hipLaunchKernelGGL(k_pack_bases<true>, dim3(blocks_for(n)), dim3(256), 0, st, c->in_bases.as<u8>(), (u32)n, word_nt,
                         c->in_words.as<u64>(), c->in_filt.as<u8>());
    else
      hipLaunchKernelGGL(k_pack_bases<false>, dim3(blocks_for(n)), dim3(256), 0, st, c->in_bases.as<u8>(), (u32)n, word_nt,
                         c->in_words.as<u64>(), c->in_filt.as<u8>());
  } else if (n) {
    HIPCHK(hipMemcpyAsync(c->in_words.p, words, n * wbytes, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(c->in_filt.p, filtered, n, hipMemcpyHostToDevice, st));
  }
  HIPCHK(hipEventRecord(e1, st));
  int rc = word_nt > 32
               ? run_device<W2>(c, c->in_words.as<W2>(), c->in_filt.as<u8>(), n_reads, word_nt, distance, method,
                                c->out_cid.as<u32>(), c->out_keep.as<u8>(), &s)
               : run_device<u64>(c, c->in_words.as<u64>(), c->in_filt.as<u8>(), n_reads, word_nt, distance, method,
                                 c->out_cid.as<u32>(), c->out_keep.as<u8>(), &s);
  if (rc == HUMID_OK) {
    hipError_t he = hipEventRecord(e2, st);
    if (he == hipSuccess && n) he = hipMemcpyAsync(cluster_id, c->out_cid.p, n * 4, hipMemcpyDeviceToHost, st);
    if (he == hipSuccess && n) he = hipMemcpyAsync(keep, c->out_keep.p, n, hipMemcpyDeviceToHost, st);
    if (he == hipSuccess) he = hipEventRecord(e3, st);
    if (he == hipSuccess) he = hipStreamSynchronize(st);
    if (he == hipSuccess) he = hipEventElapsedTime(&s.ms_h2d, e0, e1);
    if (he == hipSuccess) he = hipEventElapsedTime(&s.ms_d2h, e2, e3);
    if (he != hipSuccess) rc = fail(c, HUMID_E_HIP, "copy back: %s", hipGetErrorString(he));
  }
  (void)hipEventDestroy(e1); (void)hipEventDestroy(e2); (void)hipEventDestroy(e3);
  if (rc == HUMID_OK && summary) *summary = s;
  return rc;
}

int humid_dedup_run(humid_ctx *c, const uint64_t *words, const uint8_t *filtered, uint64_t n_reads,
                    uint32_t word_nt, uint32_t distance, uint32_t method, uint32_t *cluster_id,
                    uint8_t *keep, humid_summary *summary) {
  return run_host(c, words, filtered, nullptr, n_reads, word_nt, distance, method, cluster_id, keep, summary);
}

int humid_dedup_run_bases(humid_ctx *c, const uint8_t *bases, uint64_t n_reads, uint32_t word_nt, uint32_t distance,
                          uint32_t method, uint32_t *cluster_id, uint8_t *keep, humid_summary *summary) {
  if (n_reads && !bases) return fail(c, HUMID_E_INVALID, "null buffer");
  return run_host(c, nullptr, nullptr, bases ? bases : (const uint8_t *)"", n_reads, word_nt, distance, method,
                  cluster_id, keep, summary);
}

// the packed words and flags of the last humid_dedup_run_bases (what makeWord would have returned)
int humid_get_packed_words(humid_ctx *c, uint64_t *words, uint8_t *filtered) {
  if (!c) return fail(nullptr, HUMID_E_INVALID, "ctx is null");
  if (!c->have_run) return fail(c, HUMID_E_STATE, "no completed dedup run in this context");
  HIPCHK(hipSetDevice(c->device));
  const size_t n = (size_t)c->N, wbytes = c->word_nt > 32 ? 16 : 8;
  if (n * wbytes > c->in_words.cap || n > c->in_filt.cap) return fail(c, HUMID_E_STATE, "the last run did not go through a host entry point");
  if (n && words) HIPCHK(hipMemcpyAsync(words, c->in_words.p, n * wbytes, hipMemcpyDeviceToHost, c->stream));
  if (n && filtered) HIPCHK(hipMemcpyAsync(filtered, c->in_filt.p, n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return HUMID_OK;
}

#define NEED_RUN()                                                                            \
  do {                                                                                        \
    if (!c) return fail(nullptr, HUMID_E_INVALID, "ctx is null");                             \
    if (!c->have_graph || c->graph_mode) return fail(c, HUMID_E_STATE, "no completed dedup run / graph stage in this context"); \
    HIPCHK(hipSetDevice(c->device));                                                          \
    TRY(expand_compact(c));                                                                   \
  } while (0)

#define D2H(dst, src, bytes)                                                                  \
  do { if ((dst) && (bytes)) HIPCHK(hipMemcpyAsync((dst), (src), (bytes), hipMemcpyDeviceToHost, c->stream)); } while (0)

int humid_get_leaves(humid_ctx *c, uint64_t *word, uint32_t *count, uint32_t *first_read,
                     uint32_t *degree, uint32_t *cluster_id, uint8_t *is_max_leaf) {
  NEED_RUN();
  size_t U = (size_t)c->gU;
  if (U == 0) return HUMID_OK;
  if (first_read && !c->have_run)
    return fail(c, HUMID_E_STATE, "first_read is only available after a single-GPU humid_dedup_run*");
  D2H(word, c->g_word, U * 8 * c->g_wpr);
  D2H(count, c->g_cnt, U * 4);
  D2H(first_read, c->s_first.p, U * 4);
  D2H(degree, c->deg.p, U * 4);
  D2H(cluster_id, c->cid.p, U * 4);
  D2H(is_max_leaf, c->ismax.p, U);
  HIPCHK(hipStreamSynchronize(c->stream));
  return HUMID_OK;
}

int humid_get_adjacency(humid_ctx *c, uint32_t *nbr_off, uint32_t *nbr_idx) {
  NEED_RUN();
  size_t U = (size_t)c->gU;
  if (U == 0) { if (nbr_off) nbr_off[0] = 0; return HUMID_OK; }
  D2H(nbr_off, c->nbr_off.p, (U + 1) * 4);
  D2H(nbr_idx, c->nbr_idx.p, (size_t)(2 * c->E) * 4);
  HIPCHK(hipStreamSynchronize(c->stream));
  return HUMID_OK;
}

static int export_clusters(humid_ctx *c, u32 U, u64 C, uint64_t *size, uint32_t *max_count, uint32_t *max_leaf) {
  if (C == 0) return HUMID_OK;
  ENSURE(c->scratch, (size_t)C * 16);
  u64 *d_size = c->scratch.as<u64>();
  u32 *d_mc = (u32 *)(d_size + C);
  u32 *d_ml = d_mc + C;
  hipLaunchKernelGGL(k_export_clusters, dim3(blocks_for(U)), dim3(256), 0, c->stream, c->flag.as<u32>(),
                     c->pos.as<u32>(), c->maxleaf.as<u32>(), c->cl_size.as<u64>(), c->g_cnt, U,
                     d_size, d_mc, d_ml);
  HIPCHK(hipGetLastError());
  D2H(size, d_size, (size_t)C * 8);
  D2H(max_count, d_mc, (size_t)C * 4);
  D2H(max_leaf, d_ml, (size_t)C * 4);
  HIPCHK(hipStreamSynchronize(c->stream));
  return HUMID_OK;
}

int humid_get_clusters(humid_ctx *c, uint64_t *size, uint32_t *max_count, uint32_t *max_leaf) {
  NEED_RUN();
  return export_clusters(c, c->gU, c->C, size, max_count, max_leaf);
}

int humid_get_histogram(humid_ctx *c, uint32_t which, uint64_t *keys, uint64_t *values, uint64_t cap,
                        uint64_t *n_out) {
  NEED_RUN();
  if (which > 2 || !n_out) return fail(c, HUMID_E_INVALID, "bad histogram selector");
  *n_out = 0;
  const u32 U = c->gU;
  u64 n = (which == 2) ? c->C : U;
  if (n == 0) return HUMID_OK;
  hipStream_t st = c->stream;
  // layout of scratch: vals[n] | sorted[n] | uniq[n] | counts u32[n] | head u32[n + 1] | hpos u32[n + 1] | start u32[n + 1]
  ENSURE(c->scratch, (size_t)n * 40 + 64);
  u64 *vals = c->scratch.as<u64>();
  u64 *sorted = vals + n;
  u64 *uniq = sorted + n;
  u32 *counts = (u32 *)(uniq + n);
  u32 *head = counts + n, *hpos = head + n + 1, *start = hpos + n + 1;
  u32 *runs = hpos + n;                                      // the scan's last entry = number of runs
  if (which == 0) hipLaunchKernelGGL(k_widen32, dim3(blocks_for(U)), dim3(256), 0, st, c->g_cnt, U, vals);
  else if (which == 1) hipLaunchKernelGGL(k_widen32, dim3(blocks_for(U)), dim3(256), 0, st, c->deg.as<u32>(), U, vals);
  else hipLaunchKernelGGL(k_creator_sizes, dim3(blocks_for(U)), dim3(256), 0, st, c->flag.as<u32>(),
                          c->pos.as<u32>(), c->cl_size.as<u64>(), U, vals);
  TRY(sort_keys<u64>(c, vals, sorted, n, 0, 64));
  // run-length encode: head flags, their scan, (value, first position) per run, lengths by difference
  hipLaunchKernelGGL(k_rle_heads, dim3(blocks_for(n + 1)), dim3(256), 0, st, sorted, (u32)n, head);
  TRY(exscan_u32(c, head, hpos, n + 1));
  hipLaunchKernelGGL(k_rle_runs, dim3(blocks_for(n + 1)), dim3(256), 0, st, sorted, (const u32 *)head, (const u32 *)hpos, (u32)n, uniq, start);
  hipLaunchKernelGGL(k_rle_counts, dim3(blocks_for(n)), dim3(256), 0, st, (const u32 *)start, (const u32 *)runs, counts);
  HIPCHK(hipGetLastError());
  u32 h_runs = 0;
  HIPCHK(hipMemcpyAsync(&h_runs, runs, 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  *n_out = h_runs;
  u64 take = h_runs < cap ? h_runs : cap;
  if (take && keys && values) {
    std::vector<u32> hc(take);
    HIPCHK(hipMemcpyAsync(keys, uniq, (size_t)take * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(hc.data(), counts, (size_t)take * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    for (u64 i = 0; i < take; i++) values[i] = hc[i];
  }
  return HUMID_OK;
}

int humid_cluster_graph(humid_ctx *c, const uint32_t *count, const uint32_t *nbr_off,
                        const uint32_t *nbr_idx, uint32_t n_leaves, uint32_t method,
                        uint32_t *leaf_cluster, uint64_t *cl_size, uint32_t *cl_max_count,
                        uint32_t *cl_max_leaf, uint32_t *n_clusters) {
  if (!c) return fail(nullptr, HUMID_E_INVALID, "ctx is null");
  if (method > 1) return fail(c, HUMID_E_INVALID, "method must be 0 or 1");
  if (n_clusters) *n_clusters = 0;
  const u32 U = n_leaves;
  if (U == 0) return HUMID_OK;
  if (!count || !nbr_off) return fail(c, HUMID_E_INVALID, "null buffer");
  const u32 twoE = nbr_off[U];
  if (twoE && !nbr_idx) return fail(c, HUMID_E_INVALID, "null nbr_idx");
  for (u32 u = 0; u < U; u++)
    if (nbr_off[u] > nbr_off[u + 1]) return fail(c, HUMID_E_INVALID, "nbr_off not monotone at %u", u);
  for (u32 k = 0; k < twoE; k++)
    if (nbr_idx[k] >= U) return fail(c, HUMID_E_INVALID, "nbr_idx[%u] = %u out of range", k, nbr_idx[k]);
  HIPCHK(hipSetDevice(c->device));
  hipStream_t st = c->stream;
  c->have_run = false;
  c->have_graph = false;
  c->graph_mode = true;
  c->cg_valid = false;
  ENSURE(c->s_cnt, (size_t)U * 4);
  c->g_cnt = c->s_cnt.as<u32>();
  c->gU = U;
  ENSURE(c->deg, (size_t)U * 4);
  ENSURE(c->nbr_off, (size_t)(U + 1) * 4);
  ENSURE(c->nbr_idx, (size_t)(twoE + 1) * 4);
  ENSURE(c->parent, (size_t)U * 4);
  std::vector<u32> hdeg(U);
  for (u32 u = 0; u < U; u++) hdeg[u] = nbr_off[u + 1] - nbr_off[u];
  {
    // NLeaf::neighbours is always symmetric (src/humid.cc:121-122, tests' link()); the component
    // walk relies on it.  Two linked leaves of count 0 make maxNeighbour_ (cluster.cc:39-51)
    // ping-pong forever in the reference: refuse instead of hanging the GPU.
    std::vector<u64> fwd, rev;
    fwd.reserve(twoE); rev.reserve(twoE);
    for (u32 u = 0; u < U; u++)
      for (u32 k = nbr_off[u]; k < nbr_off[u + 1]; k++) {
        const u32 v = nbr_idx[k];
        if (count[u] == 0 && count[v] == 0)
          return fail(c, HUMID_E_INVALID, "leaves %u and %u are linked and both have count 0", u, v);
        fwd.push_back(((u64)u << 32) | v);
        rev.push_back(((u64)v << 32) | u);
      }
    std::sort(fwd.begin(), fwd.end());
    std::sort(rev.begin(), rev.end());
    if (fwd != rev) return fail(c, HUMID_E_INVALID, "neighbour lists are not symmetric");
  }
  HIPCHK(hipMemcpyAsync(c->s_cnt.p, count, (size_t)U * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(c->deg.p, hdeg.data(), (size_t)U * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(c->nbr_off.p, nbr_off, (size_t)(U + 1) * 4, hipMemcpyHostToDevice, st));
  if (twoE) HIPCHK(hipMemcpyAsync(c->nbr_idx.p, nbr_idx, (size_t)twoE * 4, hipMemcpyHostToDevice, st));
  ENSURE(c->csize, (size_t)U * 4);
  ENSURE(c->small_roots, ((size_t)U / 3 + 2) * 4);
  HIPCHK(hipMemsetAsync(c->csize.p, 0, (size_t)U * 4, st));
  HIPCHK(hipMemsetAsync(&c->d_ctr[CTR_EDGES], 0, 3 * sizeof(ull), st));
  HIPCHK(hipMemsetAsync(&c->d_ctr[CTR_SMALLROOTS], 0, sizeof(ull), st));
  hipLaunchKernelGGL(k_iota, dim3(blocks_for(U)), dim3(256), 0, st, c->parent.as<u32>(), U);
  hipLaunchKernelGGL(k_union_csr, dim3(blocks_for(U)), dim3(256), 0, st, c->nbr_off.as<u32>(),
                     c->nbr_idx.as<u32>(), U, c->parent.as<u32>(),
                     method == HUMID_METHOD_MAXIMUM ? (const u32 *)nullptr : c->s_cnt.as<u32>());
  hipLaunchKernelGGL(k_comp_stats, dim3(blocks_for(U)), dim3(256), 0, st, c->deg.as<u32>(),
                     c->parent.as<u32>(), U, c->csize.as<u32>());
  hipLaunchKernelGGL(k_comp_count, dim3(512), dim3(256), 0, st, c->deg.as<u32>(), c->parent.as<u32>(),
                     c->csize.as<u32>(), U, c->d_ctr, c->small_roots.as<u32>());
  HIPCHK(hipGetLastError());
  TRY(read_counters(c));   // also drains the stream: hdeg is a host temporary
  TRY(cluster_stage(c, c->s_cnt.as<u32>(), U, c->h_ctr[CTR_NONSINGLE], c->h_ctr[CTR_MEMBERS], method));
  hipLaunchKernelGGL(k_finalize_nodes, dim3(blocks_for(U)), dim3(256), 0, st, c->cl_of.as<u32>(),
                     c->pos.as<u32>(), c->maxleaf.as<u32>(), U, c->cid.as<u32>(), c->ismax.as<u8>(),
                     (const u32 *)nullptr, (const u32 *)nullptr, (u64 *)nullptr);
  HIPCHK(hipGetLastError());
  u64 C = 0;
  TRY(n_clusters_from_scan(c, U, &C));
  D2H(leaf_cluster, c->cid.p, (size_t)U * 4);
  HIPCHK(hipStreamSynchronize(st));
  if (n_clusters) *n_clusters = (u32)C;
  return export_clusters(c, U, C, cl_size, cl_max_count, cl_max_leaf);
}

static int compact_nodes_impl(humid_ctx *c, const uint64_t *d_edges, uint64_t n_edges, uint32_t record_stride, u64 id_bound,
                              const uint32_t **d_nodes, uint64_t *n_nodes, const uint64_t **d_compact_edges,
                              const uint32_t **d_node_counts);
static int combo_route_wide(humid_ctx *c, const W2 *d_word, const u32 *d_count, u32 n, u64 id_base, const ComboPlan &plan,
                            u32 combo, u32 n_ranks, const Item3 **d_items, u64 *counts);
static int pairs_keyed_wide(humid_ctx *c, const void *d_items, u32 n, bool interleaved, u64 id_base, const u32 *d_count,
                            const ComboPlan &plan, u32 combo, u32 distance, const u64 **d_records, u64 *n_edges);
// ---- the exchange-mode pass of one rank (include/humid_hip.h: humid_dedup_run_exchange) ----------
namespace {
struct XRange { u64 lo = 1, hi = 0; };                         // lo > hi: empty

// P ordered, disjoint, covering value ranges with balanced usable-read counts, cut at histogram bins;
// the same arithmetic on every rank (and in humid_amd/sharded.py splitters_from_hist)
void x_splitters(const std::vector<u64> &hist, u32 P, u32 word_nt, u32 bits, std::vector<XRange> &out) {
  const u32 shift = 2 * word_nt - bits;
  const size_t n_bins = hist.size();
  std::vector<u64> cum(n_bins);
  u64 total = 0;
  for (size_t i = 0; i < n_bins; i++) { total += hist[i]; cum[i] = total; }
  std::vector<size_t> bounds{0};
  for (u32 k = 1; k < P; k++) {
    const u64 target = (total * k + P - 1) / P;
    size_t b = (size_t)(std::lower_bound(cum.begin(), cum.end(), target) - cum.begin()) + 1;
    bounds.push_back(std::min(std::max(b, bounds.back()), n_bins));
  }
  bounds.push_back(n_bins);
  out.assign(P, XRange());
  for (u32 r = 0; r < P; r++) {
    const size_t b0 = bounds[r], b1 = bounds[r + 1];
    if (b1 <= b0) continue;
    out[r].lo = (u64)b0 << shift;
    out[r].hi = r == P - 1 ? ~0ull : ((u64)b1 << shift) - 1;          // (b1 << shift == 2^64 wraps to 0: hi = top)
  }
}

// count_order for the owner's count from the global histogram: 1 = the words of this value range are
// spread evenly (word-ordered LDS buckets fit), 0 = clearly not, -1 = let the count stage sample
int x_order_hint(const std::vector<u64> &hist, const XRange &rg, u32 word_nt, u32 bits) {
  if (rg.lo > rg.hi) return -1;
  const u32 shift = 2 * word_nt - bits;
  const size_t b0 = (size_t)(rg.lo >> shift), b1 = std::min<size_t>((size_t)(rg.hi >> shift), hist.size() - 1);
  if (b1 + 1 - b0 < 4) return -1;
  double sum = 0, mx = 0;
  for (size_t b = b0; b <= b1; b++) { sum += (double)hist[b]; mx = std::max(mx, (double)hist[b]); }
  if (sum < 65536) return -1;
  const double ratio = mx / (sum / (double)(b1 + 1 - b0));
  return ratio <= 1.25 ? 1 : (ratio > 2.5 ? 0 : -1);
}
}  // namespace

// host numbers of all ranks
// Failure is COLLECTIVE: every gather carries a status word per rank, and all gathers of a pass but
// the first (the histograms) have ONE size, X_SLOT -- so a rank that fails between two gathers can still join the
// next one its peers reach (run_exchange's wrapper does that for it, x_announce_failure) and every rank returns
// an error from the same gather instead of waiting for a peer that has left.  (What this covers: a failure in a
// compute phase -- a kernel error, an overflow check, a malformed record -- whose next collective is a gather.
// Not covered: running out of memory for the receive buffer BETWEEN a gather and the device exchange it sized;
// there the transport's own failure handling applies: ncclCommAbort in csrc/host/sharded.cpp, the process
// group's timeout under torch.distributed.)
#define X_SLOT 248u                 // payload bytes of the small gathers (the largest: (P + 2) x 8 = 144 at 16 ranks)
static int x_gather_slots(humid_ctx *c, const humid_comm *cm, const void *mine, u64 bytes, u64 slot, void *all, i64 my_status) {
  const u32 P = cm->world;
  std::vector<u8> out(slot + 8, 0), in((size_t)P * (slot + 8));
  if (bytes) memcpy(out.data(), mine, bytes);
  memcpy(out.data() + slot, &my_status, 8);
  if (cm->host_all_gather(cm->user, out.data(), slot + 8, in.data()) < 0) {
    c->x_peer_failed = true;                                     // (the transport itself failed: nobody is waiting for an announcement)
    return fail(c, HUMID_E_COMM, "humid_comm.host_all_gather failed");
  }
  int bad_rank = -1;
  i64 bad = 0;
  for (u32 q = 0; q < P; q++) {
    i64 stq;
    memcpy(&stq, in.data() + (size_t)q * (slot + 8) + slot, 8);
    if (stq != 0 && bad_rank < 0) { bad_rank = (int)q; bad = stq; }
    if (all && bytes) memcpy((u8 *)all + (size_t)q * bytes, in.data() + (size_t)q * (slot + 8), bytes);
  }
  if (bad_rank >= 0) {
    c->x_peer_failed = true;
    return fail(c, HUMID_E_COMM, "rank %d left the pass with error %lld; every rank returns here", bad_rank, (long long)bad);
  }
  return HUMID_OK;
}
static int x_host_gather(humid_ctx *c, const humid_comm *cm, const void *mine, u64 bytes, void *all) {
  if (!cm || (cm->world == 1 && !c->force_comm)) { memcpy(all, mine, bytes); return HUMID_OK; }
  // (test hook: this rank's compute phase in front of its k-th gather "fails")
  if (++c->x_gathers == c->x_test_fail_after) return fail(c, HUMID_E_INVALID, "test: this rank fails before gather %d", c->x_gathers);
  const bool first = !c->x_hist_done;                            // the first gather of a pass: the histograms (its own size)
  c->x_hist_done = true;
  if (!first && bytes > X_SLOT) return fail(c, HUMID_E_INVALID, "internal: a host gather of %llu bytes", (ull)bytes);
  return x_gather_slots(c, cm, mine, bytes, first ? bytes : X_SLOT, all, 0);
}
// a rank that fails joins the gather its peers reach next, with its error code in the status word
static void x_announce_failure(humid_ctx *c, const humid_comm *cm, int rc, u64 first_gather_bytes) {
  if (!cm || (cm->world == 1 && !c->force_comm) || c->x_peer_failed || !cm->host_all_gather) return;
  const std::string keep = c->err;
  const u64 slot = c->x_hist_done ? X_SLOT : first_gather_bytes;
  c->x_hist_done = true;
  (void)x_gather_slots(c, cm, nullptr, 0, slot, nullptr, rc ? rc : -1);
  c->err = keep;
}
// items of `elem` bytes: send_items[q] to rank q (laid out in rank order in d_send, or the same
// send_items[rank] items to everybody when `same`), recv_items[q] from rank q in rank order in d_recv.
// One rank: a local copy.
static int x_exchange(humid_ctx *c, const humid_comm *cm, const void *d_send, const u64 *send_items, bool same,
                      void *d_recv, const u64 *recv_items, u64 elem) {
  const u32 P = cm ? cm->world : 1, r = cm ? cm->rank : 0;
  u64 so[MAX_RANKS], sb[MAX_RANKS], ro[MAX_RANKS], rb[MAX_RANKS];
  u64 a = 0, b = 0;
  for (u32 q = 0; q < P; q++) {
    so[q] = same ? 0 : a; sb[q] = (same ? send_items[r] : send_items[q]) * elem; a += sb[q];
    ro[q] = b; rb[q] = recv_items[q] * elem; b += rb[q];
  }
  if (sb[r] != rb[r]) return fail(c, HUMID_E_INVALID, "exchange: this rank's own split sizes differ");
  if (P == 1 && !(cm && c->force_comm)) {
    if (sb[0]) HIPCHK(hipMemcpyAsync(d_recv, d_send, sb[0], hipMemcpyDeviceToDevice, c->stream));
    return HUMID_OK;
  }
  if (cm->exchange(cm->user, d_send, so, sb, d_recv, ro, rb, same ? 1 : 0, (void *)c->stream) < 0)
    return fail(c, HUMID_E_COMM, "humid_comm.exchange failed");
  return HUMID_OK;
}

static int run_exchange_impl(humid_ctx *c, const humid_comm *cm, const uint64_t *d_words, const uint8_t *d_filtered,
                             uint64_t n_local, uint32_t word_nt, uint32_t distance, uint32_t method,
                             uint32_t *d_cluster_id, uint8_t *d_keep, humid_summary *summary, humid_exchange_info *info);
int humid_dedup_run_exchange(humid_ctx *c, const humid_comm *cm, const uint64_t *d_words, const uint8_t *d_filtered,
                             uint64_t n_local, uint32_t word_nt, uint32_t distance, uint32_t method,
                             uint32_t *d_cluster_id, uint8_t *d_keep, humid_summary *summary, humid_exchange_info *info) {
  if (!c) return fail(nullptr, HUMID_E_INVALID, "ctx is null");
  c->x_hist_done = false;
  c->x_peer_failed = false;
  c->x_gathers = 0;
  const int rc = run_exchange_impl(c, cm, d_words, d_filtered, n_local, word_nt, distance, method, d_cluster_id, d_keep, summary, info);
  if (rc != HUMID_OK && cm) {
    // the size of the pass's first gather, should this rank have failed before it: the histogram table
    // (the same arithmetic as in run_exchange_impl; word lengths it refuses are refused on every rank alike)
    u64 first = 0;
    u32 nc1 = 0, pbits = 0;
    if (word_nt >= 1 && word_nt <= 64 && humid_stage_plan_info(c, word_nt, distance, 1, &nc1, &pbits) == HUMID_OK && pbits >= 1)
      first = ((u64)1 << std::min<u32>(std::min<u32>(12u, 2 * std::min<u32>(word_nt, 32u)), pbits)) * 4;
    if (first || c->x_hist_done) x_announce_failure(c, cm, rc, first);
  }
  return rc;
}
static int run_exchange_impl(humid_ctx *c, const humid_comm *cm, const uint64_t *d_words, const uint8_t *d_filtered,
                             uint64_t n_local, uint32_t word_nt, uint32_t distance, uint32_t method,
                             uint32_t *d_cluster_id, uint8_t *d_keep, humid_summary *summary, humid_exchange_info *info) {
  const u32 P = cm ? cm->world : 1, r = cm ? cm->rank : 0;
  if (P == 0 || P > MAX_RANKS || r >= P) return fail(c, HUMID_E_UNSUPPORTED, "1 .. %d ranks", MAX_RANKS);
  if ((P > 1 || (cm && c->force_comm)) && (!cm->host_all_gather || !cm->exchange)) return fail(c, HUMID_E_INVALID, "humid_comm without callbacks");
  TRY(check_run_args(c, n_local, word_nt, method, 64));
  if (n_local && (!d_words || !d_filtered || !d_cluster_id || !d_keep)) return fail(c, HUMID_E_INVALID, "null buffer");
  HIPCHK(hipSetDevice(c->device));
  hipStream_t st = c->stream;
  const auto t_begin = std::chrono::steady_clock::now();
  // HUMID_XTRACE=1: host time of every phase of the pass on stderr (each mark waits for the stream: for
  // measurements with the ranks taking turns on one GPU, tools/exchange_phase_cost.py -- never in a timed run)
  static const bool xtrace = getenv("HUMID_XTRACE") != nullptr;
  auto xt_last = t_begin;
  std::string xt_line;
  auto XT = [&](const char *name) {
    if (!xtrace) return;
    (void)hipStreamSynchronize(st);
    const auto now = std::chrono::steady_clock::now();
    char buf[64];
    snprintf(buf, sizeof buf, " %s %.3f", name, std::chrono::duration<double, std::milli>(now - xt_last).count());
    xt_line += buf;
    xt_last = std::chrono::steady_clock::now();
  };
  // 33 <= word_nt <= 64: two uint64 per read.  Value ranges are decided by the top 64 bits of the word
  // (its "head"): histogram, splitters and routing run on an array of heads exactly as they do on
  // one-word words of 32 nucleotides; what travels and what is counted are the two-word words.
  const bool wide = word_nt > 32;
  const u32 head_nt = wide ? 32u : word_nt;
  const u64 *heads = d_words;
  if (wide && n_local) {
    ENSURE(c->xr_heads, (size_t)n_local * 8);
    hipLaunchKernelGGL(k_wide_head64, dim3(blocks_for(n_local)), dim3(256), 0, st, (const W2 *)d_words, (u32)n_local,
                       2 * (word_nt - 32), c->xr_heads.as<u64>(), 0u);
    heads = c->xr_heads.as<u64>();
  }
  const bool moves = P > 1 || (cm && c->force_comm);                 // bytes go through the callbacks
  const u32 n = word_nt, d = distance;
  // ---- 1. histograms of all ranks -> value ranges and every split size of the word exchange ----
  u32 nc1 = 0, pbits = 0;
  TRY(humid_stage_plan_info(c, n, d, 1, &nc1, &pbits));
  if (pbits < 1) return fail(c, HUMID_E_UNSUPPORTED, "distance %u over %u-nt words leaves no prefix to cut value ranges at", d, n);
  const u32 bits = std::min<u32>(std::min<u32>(12u, 2 * head_nt), pbits);
  const size_t n_bins = (size_t)1 << bits;
  ENSURE(c->xr_hist, n_bins * 4);
  TRY(humid_stage_histogram(c, heads, d_filtered, n_local, head_nt, bits, c->xr_hist.as<u32>()));
  std::vector<u32> h_hist(n_bins), all_hist((size_t)P * n_bins);
  HIPCHK(hipMemcpyAsync(h_hist.data(), c->xr_hist.p, n_bins * 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  TRY(x_host_gather(c, cm, h_hist.data(), n_bins * 4, all_hist.data()));
  std::vector<u64> hist_sum(n_bins, 0), cum((size_t)P * (n_bins + 1), 0);
  for (u32 q = 0; q < P; q++)
    for (size_t b = 0; b < n_bins; b++) {
      const u64 v = all_hist[(size_t)q * n_bins + b];
      hist_sum[b] += v;
      cum[(size_t)q * (n_bins + 1) + b + 1] = cum[(size_t)q * (n_bins + 1) + b] + v;
    }
  std::vector<XRange> ranges;
  x_splitters(hist_sum, P, head_nt, bits, ranges);
  const u32 shift = 2 * head_nt - bits;
  auto in_range = [&](u32 src, u32 owner) -> u64 {                  // usable reads of rank src in owner's range
    const XRange &rg = ranges[owner];
    if (rg.lo > rg.hi) return 0;
    const size_t b0 = (size_t)(rg.lo >> shift), b1 = std::min<size_t>((size_t)(rg.hi >> shift), n_bins - 1) + 1;
    return cum[(size_t)src * (n_bins + 1) + b1] - cum[(size_t)src * (n_bins + 1) + b0];
  };
  u64 send_counts[MAX_RANKS], recv_counts[MAX_RANKS], lo[MAX_RANKS], hi[MAX_RANKS];
  u64 n_send = 0, n_recv = 0;
  for (u32 q = 0; q < P; q++) {
    send_counts[q] = in_range(r, q);
    recv_counts[q] = in_range(q, r);
    n_send += send_counts[q];
    n_recv += recv_counts[q];
    lo[q] = ranges[q].lo;
    hi[q] = ranges[q].hi;
  }
  if (n_recv > 0x7fffffffull) return fail(c, HUMID_E_OVERFLOW, "%llu reads arrive at rank %u: more than 2^31-1", (ull)n_recv, r);
  u64 lo_r = ranges[r].lo, hi_r = ranges[r].hi;
  if (lo_r > hi_r) { lo_r = 0; hi_r = ~0ull; }                      // empty range: nothing arrives
  const int saved_order = c->count_order, saved_mode = c->count_mode;
  c->count_order = x_order_hint(hist_sum, ranges[r], head_nt, bits);
  c->count_mode = 0;
  struct Restore { humid_ctx *c; int o, m; ~Restore() { c->count_order = o; c->count_mode = m; } } restore{c, saved_order, saved_mode};

  XT("hist+ranges");
  // ---- 2. usable words -> owner of their range (stable: input order inside every block) ----
  const u64 *d_routed = nullptr;
  const u32 *d_perm = nullptr;
  TRY(humid_stage_route(c, heads, d_filtered, n_local, lo, hi, P, send_counts, &d_routed, &d_perm));
  const u64 wbytes = wide ? 16 : 8;
  if (wide) {                                                        // the routed order, now of the two-word words
    ENSURE(c->xr_send, n_send * 16 + 16);
    if (n_send)
      hipLaunchKernelGGL(k_gather_w2, dim3(blocks_for(n_send)), dim3(256), 0, st, (const W2 *)d_words, d_perm, (u32)n_send,
                         c->xr_send.as<W2>());
    d_routed = c->xr_send.as<u64>();
  }
  u64 usable_all = 0;                                                // over all ranks: nothing travels when it is 0 (every rank knows)
  for (size_t b = 0; b < n_bins; b++) usable_all += hist_sum[b];
  const u64 *recv_w = d_routed;                                     // one rank: what was routed is what arrives
  if (moves && usable_all) {
    ENSURE(c->xr_recv, n_recv * wbytes + 16);
    TRY(x_exchange(c, cm, d_routed, send_counts, false, c->xr_recv.p, recv_counts, wbytes));
    recv_w = c->xr_recv.as<u64>();
  }

  XT("route+exchange");
  // ---- 3. exact counts of the received words (all usable, all in this rank's range) ----
  const u64 shard_begin[2] = {0, n_recv};
  u64 cnt_one = 0, u_local = 0, usable_local = 0;
  if (!wide) {
    TRY(humid_stage_count_dense(c, recv_w, nullptr, n_recv, n, lo_r, hi_r, shard_begin, 1, &cnt_one, &u_local,
                                &usable_local));
  } else {
    // counts by sorting (kernels_wide.hip.h), as on one GPU; every received read is usable
    c->have_run = c->have_graph = false;
    c->graph_mode = false;
    c->N = c->U = c->E = c->M = c->C = c->usable = 0;
    c->word_nt = n;
    c->dense_mode = true;
    c->stage_map_timed = false;
    if (n_recv) {
      ENSURE(c->xr_zero, n_recv + 16);
      HIPCHK(hipMemsetAsync(c->xr_zero.p, 0, n_recv, st));
      humid_summary ws;
      memset(&ws, 0, sizeof ws);
      c->N = n_recv;
      TRY(stage_count_wide(c, (const W2 *)recv_w, c->xr_zero.as<u8>(), (u32)n_recv, n, ws, lo_r, hi_r, true));
      HIPCHK(hipStreamSynchronize(st));
      if (c->usable != n_recv) return fail(c, HUMID_E_INVALID, "a filtered read among the routed wide words");
    }
    u_local = c->U;
    usable_local = c->usable;
  }
  TRY(humid_stage_route_check(c));                                  // (the stream has drained: no extra wait)
  const u64 meta[3] = {u_local, usable_local, n_local};
  u64 metas[3 * MAX_RANKS];
  TRY(x_host_gather(c, cm, meta, sizeof meta, metas));
  u64 u_total = 0, goff = 0, usable = 0, total = 0;
  for (u32 q = 0; q < P; q++) {
    if (q < r) goff += metas[3 * q];
    u_total += metas[3 * q];
    usable += metas[3 * q + 1];
    total += metas[3 * q + 2];
  }
  if (u_total >= 0xffffffffull) return fail(c, HUMID_E_OVERFLOW, "more than 2^32-2 unique words in total");
  const u64 *lw = nullptr;
  const u32 *lc = nullptr, *lfirst = nullptr;
  if (u_local) TRY(humid_stage_unique(c, &lw, &lc, &lfirst));

  XT("count");
  if (c->edit && d >= 2) {
    // ---- the edit-distance road (-e -m 2..5, src/humid.cc:140-158): the unique words of all ranks are all-gathered
    // (they are slices of the walk order: rank order = walk order), every rank runs every P-th shifted-segment join
    // over the whole array (edit_edges), the shares are gathered and made unique, and every rank clusters the WHOLE
    // graph -- no owner-local split: the joins, not the clustering, are what this mode spends its time on ----
    if (d > 5) return fail(c, HUMID_E_UNSUPPORTED, "edit distance %u > 5 is not supported", d);
    if (u_total + 8 > 0xffffffffull) return fail(c, HUMID_E_OVERFLOW, "more than 2^32-10 unique words in total");
    const u64 wb = wide ? 16 : 8;
    u64 ucnt[MAX_RANKS];
    for (u32 q = 0; q < P; q++) ucnt[q] = metas[3 * q];
    const void *gw = lw;
    const u32 *gc = lc;
    if (moves && u_total) {
      ENSURE(c->xo_gw, u_total * wb + 16);
      ENSURE(c->xo_gc, u_total * 4 + 16);
      ENSURE(c->s_word, 16);
      ENSURE(c->s_cnt, 16);
      TRY(x_exchange(c, cm, c->s_word.p, ucnt, true, c->xo_gw.p, ucnt, wb));
      TRY(x_exchange(c, cm, c->s_cnt.p, ucnt, true, c->xo_gc.p, ucnt, 4));
      gw = c->xo_gw.p;
      gc = c->xo_gc.as<u32>();
    }
    u64 e_raw = 0;
    if (u_total > 1) {
      if (wide) TRY(edit_edges<W2>(c, (const W2 *)gw, (u32)u_total, n, d, &e_raw, r, P, false));
      else TRY(edit_edges<u64>(c, (const u64 *)gw, (u32)u_total, n, d, &e_raw, r, P, false));
    }
    u64 raw_from[MAX_RANKS], raw_all = 0;
    TRY(x_host_gather(c, cm, &e_raw, 8, raw_from));
    for (u32 q = 0; q < P; q++) raw_all += raw_from[q];
    if (raw_all >= 0x7fffffffull) return fail(c, HUMID_E_OVERFLOW, "too many candidate pairs in the edit-distance search");
    const u64 *all_raw = c->e_raw.as<u64>();
    if (moves && raw_all) {
      ENSURE(c->xo_cross, raw_all * 8 + 16);
      ENSURE(c->e_raw, 16);
      u64 rs[MAX_RANKS];
      for (u32 q = 0; q < P; q++) rs[q] = e_raw;
      TRY(x_exchange(c, cm, c->e_raw.p, rs, true, c->xo_cross.p, raw_from, 8));
      all_raw = c->xo_cross.as<u64>();
    }
    u64 E_e = 0;
    if (raw_all) TRY(unique_edges(c, all_raw, raw_all, (u32)u_total, &E_e));
    XT("edit joins");
    const u32 n_ids = (u32)u_total;
    const u32 nw = (((n_ids + 31) / 32) + 7) & ~7u, nblk = nw / 8;
    c->cg_valid = false;
    c->cg_nblocks = nblk;
    ENSURE(c->cg_bits, (size_t)nw * 4);
    ENSURE(c->cg_nbits, (size_t)nw * 4);
    ENSURE(c->cg_cur, (size_t)(ER_REGIONS * ER_STRIDE + 8) * 4);
    ENSURE(c->xo_cnt, 64 * 4);
    {
      ZeroList z;
      memset(&z, 0, sizeof z);
      z.p[0] = c->cg_bits.as<u32>(); z.n[0] = nw;
      z.p[1] = c->cg_nbits.as<u32>(); z.n[1] = nw;
      z.p[2] = (u32 *)&c->d_ctr[CTR_EDGES]; z.n[2] = 2 * (CTR_GOVER - CTR_EDGES + 1);
      z.p[3] = c->cg_cur.as<u32>(); z.n[3] = ER_REGIONS * ER_STRIDE + 8;
      hipLaunchKernelGGL(k_zero_many, dim3(64), dim3(256), 0, st, z);
    }
    CgStatus cgs;
    u64 M_e = 0;
    if (E_e) {
      hipLaunchKernelGGL(k_mark_pairs, dim3(blocks_for(E_e)), dim3(256), 0, st, (const u64 *)c->e_edges.as<u64>(), (u32)E_e, n_ids,
                         c->cg_bits.as<u32>(), (u32 *)&c->d_ctr[CTR_OVERFULL]);
      CgSource src;
      src.er.e = nullptr; src.er.cap_r = 0; src.er.cur = c->cg_cur.as<u32>(); src.er.far = c->e_edges.as<u64>(); src.er.n_far = (u32)E_e;
      src.recs = nullptr; src.n_recs = 0; src.segs = nullptr; src.cnt_by_id = gc; src.n_ids = n_ids;
      src.pairs_bound = E_e;
      if (!c->lean_events) HIPCHK(hipEventRecord(c->ev[2], st));
      TRY(cg_build(c, src, method, cgs));
      if (c->h_ctr[CTR_OVERFULL]) return fail(c, HUMID_E_INVALID, "internal: an edit-distance pair outside the unique words");
      M_e = cgs.M;
      TRY(cg_cluster_rest(c, n_ids, cgs.M, cgs.Mbig, method));
    } else {
      ENSURE(c->cg_blk, ((size_t)nblk + 1) * 4);
      ENSURE(c->cg_nblk, ((size_t)nblk + 1) * 4);
      HIPCHK(hipMemsetAsync(c->cg_blk.p, 0, ((size_t)nblk + 1) * 4, st));
      HIPCHK(hipMemsetAsync(c->cg_nblk.p, 0, ((size_t)nblk + 1) * 4, st));
    }
    const GraphArrays cg = cg_arrays(c);
    const BitRank br_in{c->cg_bits.as<u32>(), c->cg_blk.as<u32>()}, br_nc{c->cg_nbits.as<u32>(), c->cg_nblk.as<u32>()};
    const u32 *l_cid = nullptr;
    const u8 *l_ismax = nullptr;
    if (u_local) {
      ENSURE(c->x_lcid, (size_t)u_local * 4);
      ENSURE(c->x_lismax, (size_t)u_local);
      ENSURE(c->xo_ldeg, (size_t)u_local * 4);
      hipLaunchKernelGGL(k_own_results, dim3(blocks_for(u_local)), dim3(256), 0, st, br_in, br_nc, br_nc, (const u32 *)nullptr,
                         (const u32 *)c->cg_nodes.as<u32>(), (const u32 *)cg.cl_of, (const u32 *)cg.maxleaf, (const u32 *)cg.deg, (u32)goff,
                         (u32)u_local, 0u, c->x_lcid.as<u32>(), c->x_lismax.as<u8>(), c->xo_ldeg.as<u32>(),
                         (const u32 *)c->s_first.as<u32>(), (const u32 *)c->s_slot.as<u32>(), c->slot_out.as<u64>(), true);
      c->slots_done = true;
      l_cid = c->x_lcid.as<u32>();
      l_ismax = c->x_lismax.as<u8>();
    }
    HIPCHK(hipGetLastError());
    TRY(read_counters(c, c->cg_nblk.as<u32>() + nblk));                 // nodes that created no cluster, all ranks
    const u64 clusters_e = u_total - (c->h_ctr[CTR_N - 1] & 0xffffffffull);
    if (clusters_e >= (1ull << 31)) return fail(c, HUMID_E_OVERFLOW, "cluster ids exceed 31 bits");
    XT("graph+ids");
    const u32 *packed = nullptr;
    u64 n_packed = 0;
    TRY(humid_stage_map_dense(c, l_cid, l_ismax, &packed, &n_packed));
    if (n_packed != n_recv) return fail(c, HUMID_E_INVALID, "map_dense returned %llu reads, %llu were counted", (ull)n_packed, (ull)n_recv);
    const u32 *ret = packed;
    if (moves && usable_all) {
      ENSURE(c->xr_ret, n_send * 4 + 8);
      TRY(x_exchange(c, cm, packed, recv_counts, false, c->xr_ret.p, send_counts, 4));
      ret = c->xr_ret.as<u32>();
    }
    if (n_local)
      hipLaunchKernelGGL(k_gather_results, dim3(grid_stride_blocks(n_local)), dim3(256), 0, st, (const u32 *)c->xo_inv.as<u32>(), ret,
                         (u32)n_send, (u32)n_local, d_cluster_id, d_keep);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));
    XT("return");
    if (xtrace) fprintf(stderr, "[xtrace] rank %u/%u (edit distance) |%s\n", r, P, xt_line.c_str());
    if (summary) {
      memset(summary, 0, sizeof *summary);
      summary->total = total;
      summary->usable = usable;
      summary->unique = u_total;
      summary->clusters = clusters_e;
      summary->edges = E_e;
      summary->nonsingle = M_e;
      summary->ms_total = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    }
    if (info) {
      info->unique_local = u_local;
      info->id_base = goff;
      info->n_nodes = M_e;
      info->n_pairs = E_e;
      info->d_unique_count = lc;
      info->d_unique_degree = u_local ? c->xo_ldeg.as<u32>() : nullptr;
    }
    return HUMID_OK;
  }
  // ---- 4. neighbour pairs in global unique indices, each with the counts of its endpoints ----
  u64 e_mine = 0;                                                    // 16-byte records in xr_eloc
  auto append_pairs = [&](const u64 *rec, u64 n_rec) -> int {
    if (!n_rec) return HUMID_OK;
    if ((e_mine + n_rec) * 16 > c->xr_eloc.cap) {                    // grow, keeping what is there
      DBuf bigger;
      HIPCHK(bigger.ensure((e_mine + n_rec) * 32, nullptr));
      if (e_mine) HIPCHK(hipMemcpyAsync(bigger.p, c->xr_eloc.p, e_mine * 16, hipMemcpyDeviceToDevice, st));
      HIPCHK(hipStreamSynchronize(st));
      c->xr_eloc.release();
      c->xr_eloc = bigger;                                           // (DBuf owns nothing by itself: a plain hand-over)
    }
    HIPCHK(hipMemcpyAsync(c->xr_eloc.as<u8>() + e_mine * 16, rec, n_rec * 16, hipMemcpyDeviceToDevice, st));
    e_mine += n_rec;
    return HUMID_OK;
  };
  // (rounds 1-2: count, scan, host wait, fill, records, copy -- per combination.  Kept as the road for inputs with
  // large buckets and for a pass whose record regions overflowed.)
  auto discover_dense = [&]() -> int {
    if (d > 0 && u_total > 1) {
      u32 n_combos = 0, pb2 = 0;
      TRY(humid_stage_plan_info(c, n, d, u_total, &n_combos, &pb2));
      const ComboPlan wplan = make_plan(n, d, u_total, c->force_segments);      // (the wide helpers take the plan itself)
      const u64 ibytes = wide ? sizeof(Item3) : 16;
      const u64 *rec = nullptr;
      u64 n_rec = 0;
      if (u_local > 1) {
        if (wide) TRY(pairs_keyed_wide(c, lw, (u32)u_local, false, goff, lc, wplan, 0, d, &rec, &n_rec));
        else TRY(humid_stage_pairs_keyed(c, lw, u_local, 0, goff, lc, n, d, u_total, 0, &rec, &n_rec));
        TRY(append_pairs(rec, n_rec));
      }
      for (u32 cb = 1; cb < n_combos; cb++) {
        const u64 *items = nullptr;
        u64 sc[MAX_RANKS] = {0}, all_sc[MAX_RANKS * MAX_RANKS], rc[MAX_RANKS];
        if (wide) {
          const Item3 *it3 = nullptr;
          TRY(combo_route_wide(c, (const W2 *)lw, lc, (u32)u_local, goff, wplan, cb, P, &it3, sc));
          items = (const u64 *)it3;
        } else
          TRY(humid_stage_combo_route(c, lw, lc, u_local, goff, n, d, u_total, cb, P, &items, sc));
        TRY(x_host_gather(c, cm, sc, P * 8, all_sc));
        u64 n_got = 0;
        for (u32 q = 0; q < P; q++) { rc[q] = all_sc[(size_t)q * P + r]; n_got += rc[q]; }
        if (n_got > 0x7fffffffull) return fail(c, HUMID_E_OVERFLOW, "%llu unique words arrive at rank %u for one combination", (ull)n_got, r);
        u64 items_all = 0;
        for (u32 q = 0; q < P * P; q++) items_all += all_sc[q];
        const u64 *got = items;
        if (moves && items_all) {
          ENSURE(c->xr_got, n_got * ibytes + 32);
          TRY(x_exchange(c, cm, items, sc, false, c->xr_got.p, rc, ibytes));
          got = c->xr_got.as<u64>();
        }
        if (n_got > 1) {
          if (wide) TRY(pairs_keyed_wide(c, got, (u32)n_got, true, 0, nullptr, wplan, cb, d, &rec, &n_rec));
          else TRY(humid_stage_pairs_keyed(c, got, n_got, 1, 0, nullptr, n, d, u_total, cb, &rec, &n_rec));
          TRY(append_pairs(rec, n_rec));
        }
      }
    }
    return HUMID_OK;
  };
  // Round 3: the search APPENDS its pairs as records (k_pairs_records: one walk per position, one global atomic
  // per workgroup, 64 append regions) -- no count / scan / fill phases and no host wait per combination.
  const bool by_count = (method & 1) == 0;
  RecRegs mine;                                                      // this rank's discoveries
  mine.e = nullptr; mine.cap_r = 0; mine.cur = nullptr; mine.far = nullptr; mine.n_far = 0;
  ENSURE(c->cg_cur, (size_t)(ER_REGIONS * ER_STRIDE + 8) * 4);
  ENSURE(c->xo_cnt, 64 * 4);
  ENSURE(c->small, 64);
  u32 *dcnt = c->xo_cnt.as<u32>();                                  // [0, P]: records per destination; [32, 32 + P]: scatter cursors; 48: flagged; 56..: totals
  bool use_regions = !c->edit && d > 0 && u_total > 1 && c->walk_max > 0;
  bool flagged_mine = false;                                         // a region overflowed / a bucket beyond the walk: this pass takes the dense road
  auto zero_discovery = [&]() -> int {
    ZeroList z;
    memset(&z, 0, sizeof z);
    z.p[0] = c->cg_cur.as<u32>(); z.n[0] = ER_REGIONS * ER_STRIDE + 8;
    z.p[1] = dcnt; z.n[1] = 64;
    z.p[2] = (u32 *)&c->d_ctr[CTR_EDGES]; z.n[2] = 2 * (CTR_GOVER - CTR_EDGES + 1);
    hipLaunchKernelGGL(k_zero_many, dim3(8), dim3(256), 0, st, z);
    return HUMID_OK;
  };
  TRY(zero_discovery());
  if (use_regions) {
    if (c->xr_ecap == 0) c->xr_ecap = std::max<u64>(u_local / 4, 4096);
    mine.cap_r = (u32)std::min<u64>((c->xr_ecap + ER_REGIONS - 1) / ER_REGIONS, 0x7fffffffull / ER_REGIONS);
    ENSURE(c->xo_regs, (size_t)ER_REGIONS * mine.cap_r * 16);
    mine.e = c->xo_regs.as<ulonglong2>();
    mine.cur = c->cg_cur.as<u32>();
    const ComboPlan plan = make_plan(n, d, u_total, c->force_segments);
    const u64 ibytes = wide ? sizeof(Item3) : 16;
    ull *big = &c->d_ctr[CTR_BIGMASK];
    u32 *over = (u32 *)&c->d_ctr[CTR_EOVER];
#define PAIRS_RECORDS(WT, P0, W, V, NN, CB, IDOF, IDBASE, CNTOF)                                                              \
  do {                                                                                                                        \
    EarlierMasksT<WT> em_;                                                                                                    \
    for (u32 t_ = 0; t_ < MAX_COMBOS; t_++) em_.m[t_] = w_from<WT>(plan.mask[t_]);                                            \
    hipLaunchKernelGGL((k_pairs_records<P0, WT>), dim3(blocks_for(NN)), dim3(256), 0, st, (const WT *)(W), (const u32 *)(V), \
                       (u32)(NN), w_from<WT>(plan.mask[CB]), em_, (u32)(CB), d, c->walk_max, (const u32 *)(IDOF), (u32)(IDBASE), \
                       (const u32 *)(CNTOF), mine, big, over);                                                                \
  } while (0)
    if (u_local > 1) {
      if (wide) PAIRS_RECORDS(W2, true, lw, nullptr, u_local, 0, nullptr, goff, lc);
      else PAIRS_RECORDS(u64, true, lw, nullptr, u_local, 0, nullptr, goff, lc);
    }
    for (u32 cb = 1; cb < plan.ncombo; cb++) {
      const u64 *items = nullptr;
      u64 sc[MAX_RANKS] = {0}, all_sc[MAX_RANKS * MAX_RANKS], rc[MAX_RANKS];
      if (!moves) sc[0] = u_local;                       // (one rank: no item list is made, see below)
      else if (wide) {
        const Item3 *it3 = nullptr;
        TRY(combo_route_wide(c, (const W2 *)lw, lc, (u32)u_local, goff, plan, cb, P, &it3, sc));
        items = (const u64 *)it3;
      } else
        TRY(humid_stage_combo_route(c, lw, lc, u_local, goff, n, d, u_total, cb, P, &items, sc));
      TRY(x_host_gather(c, cm, sc, P * 8, all_sc));
      u64 n_got = 0;
      for (u32 q = 0; q < P; q++) { rc[q] = all_sc[(size_t)q * P + r]; n_got += rc[q]; }
      if (n_got > 0x7fffffffull) return fail(c, HUMID_E_OVERFLOW, "%llu unique words arrive at rank %u for one combination", (ull)n_got, r);
      u64 items_all = 0;
      for (u32 q = 0; q < P * P; q++) items_all += all_sc[q];
      const u64 *got = items;
      if (moves && items_all) {
        ENSURE(c->xr_got, n_got * ibytes + 32);
        TRY(x_exchange(c, cm, items, sc, false, c->xr_got.p, rc, ibytes));
        got = c->xr_got.as<u64>();
      }
      if (n_got > 1 && !moves) {
        // one rank, nothing travels: the unique array itself is the item list (ids goff + position, counts lc)
        const u32 ng = (u32)u_local;
        ENSURE(c->seg_k0, (size_t)ng * 8);
        ENSURE(c->seg_v0, (size_t)ng * 4);
        ENSURE(c->seg_ks, (size_t)ng * 8);
        ENSURE(c->seg_vs, (size_t)ng * 4);
        ENSURE(c->seg_ws, (size_t)ng * (wide ? 16 : 8));
        if (wide) {
          TRY(bucket_order<W2>(c, plan, cb, (const W2 *)lw, ng, c->seg_ws.as<W2>(), c->seg_vs.as<u32>()));
          PAIRS_RECORDS(W2, false, c->seg_ws.p, c->seg_vs.p, ng, cb, nullptr, goff, lc);
        } else {
          TRY(bucket_order<u64>(c, plan, cb, lw, ng, c->seg_ws.as<u64>(), c->seg_vs.as<u32>()));
          PAIRS_RECORDS(u64, false, c->seg_ws.p, c->seg_vs.p, ng, cb, nullptr, goff, lc);
        }
      } else if (n_got > 1) {
        const u32 ng = (u32)n_got;
        ENSURE(c->x_w, (size_t)ng * (wide ? 16 : 8));
        ENSURE(c->x_id, (size_t)ng * 4);
        ENSURE(c->x_cnt, (size_t)ng * 4);
        ENSURE(c->seg_k0, (size_t)ng * 8);
        ENSURE(c->seg_v0, (size_t)ng * 4);
        ENSURE(c->seg_ks, (size_t)ng * 8);
        ENSURE(c->seg_vs, (size_t)ng * 4);
        ENSURE(c->seg_ws, (size_t)ng * (wide ? 16 : 8));
        if (wide) {
          hipLaunchKernelGGL(k_split_items_w2, dim3(blocks_for(ng)), dim3(256), 0, st, (const Item3 *)got, ng, c->x_w.as<W2>(),
                             c->x_id.as<u32>(), c->x_cnt.as<u32>());
          TRY(bucket_order<W2>(c, plan, cb, c->x_w.as<W2>(), ng, c->seg_ws.as<W2>(), c->seg_vs.as<u32>()));
          PAIRS_RECORDS(W2, false, c->seg_ws.p, c->seg_vs.p, ng, cb, c->x_id.p, 0, c->x_cnt.p);
        } else {
          hipLaunchKernelGGL(k_split_items, dim3(blocks_for(ng)), dim3(256), 0, st, (const ulonglong2 *)got, ng, c->x_w.as<u64>(),
                             c->x_id.as<u32>(), c->x_cnt.as<u32>());
          TRY(bucket_order<u64>(c, plan, cb, c->x_w.as<u64>(), ng, c->seg_ws.as<u64>(), c->seg_vs.as<u32>()));
          PAIRS_RECORDS(u64, false, c->seg_ws.p, c->seg_vs.p, ng, cb, c->x_id.p, 0, c->x_cnt.p);
        }
      }
    }
#undef PAIRS_RECORDS
    hipLaunchKernelGGL(k_rec_regions_max, dim3(1), dim3(64), 0, st, mine, c->small.as<u32>());
    HIPCHK(hipGetLastError());
    TRY(read_counters(c, c->small.as<u32>(), c->small.as<u32>() + 1));     // fullest region's demand, records held
    const u64 want_r = c->h_ctr[CTR_N - 1] & 0xffffffffull;
    e_mine = c->h_ctr[CTR_N - 2] & 0xffffffffull;
    flagged_mine = (c->h_ctr[CTR_EOVER] & 0xffffffffull) != 0 || c->h_ctr[CTR_BIGMASK] != 0;
    const u64 wanted = want_r * ER_REGIONS;
    if (c->h_ctr[CTR_EOVER] & 0xffffffffull) c->xr_ecap = wanted + wanted / 2 + ER_REGIONS * 64;
    else if (2 * (wanted + wanted / 4 + ER_REGIONS * 64) < c->xr_ecap) c->xr_ecap = wanted + wanted / 4 + ER_REGIONS * 64;
  } else {
    TRY(discover_dense());
    mine.cur = c->cg_cur.as<u32>();
    mine.far = (const ulonglong2 *)c->xr_eloc.p;
    mine.n_far = (u32)e_mine;
  }

  XT("pairs");
  // ---- 5. every pair to the owner of its ends; pairs with two owners, and the components they touch, to everybody ----
  if (u_total + 8 > 0xffffffffull) return fail(c, HUMID_E_OVERFLOW, "more than 2^32-10 unique words in total");
  IdRanges idr;
  {
    u64 at = 0;
    for (u32 q = 0; q <= MAX_RANKS; q++) { idr.b[q] = (u32)at; if (q < P) at += metas[3 * q]; }
  }
  u32 *x_bad = (u32 *)&c->d_ctr[CTR_OVERFULL];                       // a malformed record (read at the graph's host wait)
  u64 dest_cnt[MAX_RANKS + 2] = {0};                                 // [P + 1]: this rank asks everybody for the dense road
  u64 all_dest[MAX_RANKS * (MAX_RANKS + 2)];
  u32 cgx = 1;
  for (int round = 0;; round++) {
    if (e_mine > 0x7fffffffull) return fail(c, HUMID_E_OVERFLOW, "%llu neighbour pairs found by one rank", (ull)e_mine);
    cgx = (u32)std::min<u64>(std::max<u64>(blocks_for(std::max<u64>(mine.cap_r, mine.n_far)), 1), 1024);
    for (u32 q = 0; q <= P + 1; q++) dest_cnt[q] = 0;
    dest_cnt[P + 1] = flagged_mine ? 1 : 0;
    if (flagged_mine) {
    } else if (P == 1) dest_cnt[0] = e_mine;
    else if (e_mine) {
      std::vector<u32> h(P + 1);
      hipLaunchKernelGGL(k_rec_dest_count, dim3(cgx, ER_REGIONS + 1), dim3(256), 0, st, mine, idr, P, dcnt);
      HIPCHK(hipMemcpyAsync(h.data(), dcnt, (P + 1) * 4, hipMemcpyDeviceToHost, st));
      HIPCHK(hipStreamSynchronize(st));
      for (u32 q = 0; q <= P; q++) dest_cnt[q] = h[q];
    }
    TRY(x_host_gather(c, cm, dest_cnt, (P + 2) * 8, all_dest));
    bool anybody = false;
    for (u32 q = 0; q < P; q++) anybody = anybody || all_dest[(size_t)q * (P + 2) + P + 1] != 0;
    if (!anybody) break;
    if (round) return fail(c, HUMID_E_INVALID, "internal: the dense pair search asked for itself");
    // some rank's regions overflowed or met a bucket beyond the walk: EVERY rank repeats the search on the dense
    // road (its exchanges are collective), this pass only
    e_mine = 0;
    TRY(zero_discovery());
    TRY(discover_dense());
    mine.e = nullptr; mine.cap_r = 0; mine.cur = c->cg_cur.as<u32>();
    mine.far = (const ulonglong2 *)c->xr_eloc.p;
    mine.n_far = (u32)e_mine;
    flagged_mine = false;
    use_regions = false;
  }
  u64 E = 0, X_total = 0, n_int = 0, int_from[MAX_RANKS], cross_from[MAX_RANKS], int_to[MAX_RANKS];
  for (u32 q = 0; q < P; q++) {
    for (u32 dd = 0; dd <= P; dd++) E += all_dest[(size_t)q * (P + 2) + dd];
    int_from[q] = all_dest[(size_t)q * (P + 2) + r];
    cross_from[q] = all_dest[(size_t)q * (P + 2) + P];
    int_to[q] = dest_cnt[q];
    n_int += int_from[q];
    X_total += cross_from[q];
  }
  if (n_int > 0x7fffffffull || X_total > 0x7fffffffull) return fail(c, HUMID_E_OVERFLOW, "too many neighbour pairs for one rank");
  // destination-major copy of this rank's records (one rank: the list as it stands)
  const ulonglong2 *sendbuf = (const ulonglong2 *)c->xr_eloc.p;
  u64 send_base[MAX_RANKS + 2] = {0};
  for (u32 q = 0; q <= P; q++) send_base[q + 1] = send_base[q] + dest_cnt[q];
  if ((P > 1 || use_regions) && e_mine) {                            // (regions: also what makes one dense list of them)
    ENSURE(c->xo_send, e_mine * 16 + 16);
    IdRanges base;
    for (u32 q = 0; q <= MAX_RANKS; q++) base.b[q] = (u32)send_base[q <= P ? q : P + 1];
    hipLaunchKernelGGL(k_rec_dest_scatter, dim3(cgx, ER_REGIONS + 1), dim3(256), 0, st, mine, idr, P, base, dcnt + 32,
                       c->xo_send.as<ulonglong2>());
    sendbuf = c->xo_send.as<ulonglong2>();
  }
  const ulonglong2 *d_int = sendbuf, *d_cross = nullptr;             // interior records of this rank, crossing records of all
  if (moves) {
    ENSURE(c->xo_int, n_int * 16 + 16);
    ENSURE(c->xo_cross, X_total * 16 + 16);
    if (E - X_total) TRY(x_exchange(c, cm, sendbuf, int_to, false, c->xo_int.p, int_from, 16));
    d_int = c->xo_int.as<ulonglong2>();
    if (X_total) {
      u64 xs[MAX_RANKS];
      for (u32 q = 0; q < P; q++) xs[q] = dest_cnt[P];
      TRY(x_exchange(c, cm, sendbuf + send_base[P], xs, true, c->xo_cross.p, cross_from, 16));
      d_cross = c->xo_cross.as<ulonglong2>();
    }
  }
  XT("classify+exchange");
  // the interior pairs of the components a crossing pair touches: to everybody as well
  u64 k_mine = 0, k_from[MAX_RANKS] = {0}, K_total = 0;
  const ulonglong2 *d_kall = nullptr;
  if (X_total) {
    if (u_local) {
      ENSURE(c->xo_parent, (size_t)u_local * 4);
      ENSURE(c->xo_flag, (size_t)u_local + 16);
      ENSURE(c->xo_sel, n_int * 16 + 16);
      hipLaunchKernelGGL(k_iota, dim3(blocks_for(u_local)), dim3(256), 0, st, c->xo_parent.as<u32>(), (u32)u_local);
      HIPCHK(hipMemsetAsync(c->xo_flag.p, 0, (size_t)u_local, st));
      if (n_int)
        hipLaunchKernelGGL(k_union_records, dim3(blocks_for(n_int)), dim3(256), 0, st, d_int, (u32)n_int, (u32)goff, (u32)u_local,
                           c->xo_parent.as<u32>(), by_count, x_bad);
      hipLaunchKernelGGL(k_flag_crossing, dim3(blocks_for(X_total)), dim3(256), 0, st, d_cross, (u32)X_total, (u32)goff, (u32)u_local,
                         (const u32 *)c->xo_parent.as<u32>(), c->xo_flag.as<u8>(), by_count);
      if (n_int) {
        hipLaunchKernelGGL(k_select_flagged<false>, dim3(std::min<u32>(blocks_for(n_int), 1024)), dim3(256), 0, st, d_int, (u32)n_int,
                           (u32)goff, (const u32 *)c->xo_parent.as<u32>(), (const u8 *)c->xo_flag.as<u8>(), dcnt + 48,
                           c->xo_sel.as<ulonglong2>());
        u32 h = 0;
        HIPCHK(hipMemcpyAsync(&h, dcnt + 48, 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        k_mine = h;
      }
    }
    TRY(x_host_gather(c, cm, &k_mine, 8, k_from));
    for (u32 q = 0; q < P; q++) K_total += k_from[q];
    if (K_total > 0x7fffffffull) return fail(c, HUMID_E_OVERFLOW, "too many neighbour pairs for one rank");
    if (K_total) {
      ENSURE(c->xo_selall, K_total * 16 + 16);
      ENSURE(c->xo_sel, 16);
      u64 ks[MAX_RANKS];
      for (u32 q = 0; q < P; q++) ks[q] = k_mine;
      TRY(x_exchange(c, cm, c->xo_sel.p, ks, true, c->xo_selall.p, k_from, 16));
      d_kall = c->xo_selall.as<ulonglong2>();
    }
  }

  XT("flagged");
  // ---- 5b. ONE compact graph over global unique indices: own pairs + crossing pairs + the others' flagged pairs ----
  RecSegs segs;
  memset(&segs, 0, sizeof segs);
  {
    u64 kb = 0;                                                      // records of the lower ranks in the gathered flagged list
    for (u32 q = 0; q < r; q++) kb += k_from[q];
    segs.p[0] = d_int; segs.n[0] = (u32)n_int;
    segs.p[1] = d_cross; segs.n[1] = (u32)X_total;
    segs.p[2] = d_kall; segs.n[2] = (u32)kb;
    segs.p[3] = d_kall ? d_kall + kb + k_mine : nullptr; segs.n[3] = (u32)(K_total - kb - k_mine);
    for (u32 q = 0; q < REC_SEGS; q++) segs.first[q + 1] = segs.first[q] + segs.n[q];
  }
  const u64 n_recs_all = segs.first[REC_SEGS];
  const u32 n_ids = (u32)u_total;
  const u32 nw = (((n_ids + 31) / 32) + 7) & ~7u, nblk = nw / 8;
  c->cg_valid = false;
  c->cg_nblocks = nblk;
  ENSURE(c->cg_bits, (size_t)nw * 4);
  ENSURE(c->cg_nbits, (size_t)nw * 4);
  ENSURE(c->xo_xcbits, (size_t)nw * 4);
  ENSURE(c->xo_xcblk, ((size_t)nblk + 1) * 4);
  {
    ZeroList z;
    memset(&z, 0, sizeof z);
    z.p[0] = c->cg_bits.as<u32>(); z.n[0] = nw;
    z.p[1] = c->cg_nbits.as<u32>(); z.n[1] = nw;
    z.p[2] = X_total ? c->xo_xcbits.as<u32>() : nullptr; z.n[2] = X_total ? nw : 0;
    z.p[3] = (u32 *)&c->d_ctr[CTR_EDGES]; z.n[3] = 2 * (CTR_GOVER - CTR_EDGES + 1);
    z.p[4] = c->cg_cur.as<u32>(); z.n[4] = ER_REGIONS * ER_STRIDE;   // (not the bad flag behind them)
    hipLaunchKernelGGL(k_zero_many, dim3(64), dim3(256), 0, st, z);
  }
  humid_summary gs;
  memset(&gs, 0, sizeof gs);
  CgStatus cgs;
  u64 M_mine = 0;
  if (n_recs_all) {
    u32 n_max = 1;
    for (u32 q = 0; q < REC_SEGS; q++) n_max = std::max(n_max, segs.n[q]);
    hipLaunchKernelGGL(k_mark_segs, dim3(std::min<u32>(blocks_for(n_max), 4096), REC_SEGS), dim3(256), 0, st, segs, n_ids,
                       c->cg_bits.as<u32>(), x_bad);
    CgSource src;
    src.er.e = nullptr; src.er.cap_r = 0; src.er.cur = c->cg_cur.as<u32>(); src.er.far = nullptr; src.er.n_far = 0;
    src.recs = nullptr; src.n_recs = 0; src.segs = &segs; src.cnt_by_id = nullptr; src.n_ids = n_ids;
    src.pairs_bound = n_recs_all;
    if (!c->lean_events) HIPCHK(hipEventRecord(c->ev[2], st));
    TRY(cg_build(c, src, method, cgs));
    if (c->h_ctr[CTR_OVERFULL]) return fail(c, HUMID_E_INVALID, "a pair record with an index outside the unique words");
    M_mine = cgs.M;
    TRY(cg_cluster_rest(c, n_ids, cgs.M, cgs.Mbig, method));
  } else {
    ENSURE(c->cg_blk, ((size_t)nblk + 1) * 4);
    ENSURE(c->cg_nblk, ((size_t)nblk + 1) * 4);
    HIPCHK(hipMemsetAsync(c->cg_blk.p, 0, ((size_t)nblk + 1) * 4, st));
    HIPCHK(hipMemsetAsync(c->cg_nblk.p, 0, ((size_t)nblk + 1) * 4, st));
  }
  const GraphArrays cg = cg_arrays(c);
  const BitRank br_in{c->cg_bits.as<u32>(), c->cg_blk.as<u32>()}, br_nc{c->cg_nbits.as<u32>(), c->cg_nblk.as<u32>()};
  BitRank br_xc{c->xo_xcbits.as<u32>(), c->xo_xcblk.as<u32>()};

  XT("graph");
  // ---- 5c. cluster ids: creators before a leaf = the lower ranks' creators + its owner's creators before it ----
  u64 C_x = 0;
  if (X_total && M_mine) {
    ENSURE(c->xo_xroot, (size_t)M_mine + 16);
    HIPCHK(hipMemsetAsync(c->xo_xroot.p, 0, (size_t)M_mine, st));
    hipLaunchKernelGGL(k_flag_xroots, dim3(std::min<u32>(blocks_for(X_total), 4096)), dim3(256), 0, st, segs, 1u, n_ids, br_in,
                       (const u32 *)cg.parent, c->xo_xroot.as<u8>(), by_count);
    hipLaunchKernelGGL(k_xcreator_bits, dim3(blocks_for(M_mine)), dim3(256), 0, st, (const u32 *)cg.cl_of, (const u32 *)cg.parent,
                       (const u8 *)c->xo_xroot.as<u8>(), (const u32 *)c->cg_nodes.as<u32>(), (u32)M_mine, c->xo_xcbits.as<u32>());
  }
  if (X_total) {
    TRY(exscan_in<u32>(c, BitsBlockIn{c->xo_xcbits.as<u32>(), nblk}, c->xo_xcblk.as<u32>(), (u64)nblk + 1));
  }
  hipLaunchKernelGGL(k_own_totals, dim3(1), dim3(64), 0, st, br_nc, br_in, (u32)goff, (u32)u_local, dcnt + 56);
  HIPCHK(hipGetLastError());
  TRY(read_counters(c, dcnt + 56, dcnt + 57, X_total ? c->xo_xcblk.as<u32>() + nblk : nullptr));
  u64 tot_mine[4] = {c->h_ctr[CTR_N - 1] & 0xffffffffull, c->h_ctr[CTR_N - 2] & 0xffffffffull,
                     X_total ? (c->h_ctr[CTR_N - 3] & 0xffffffffull) : 0ull, u_local};   // non-creators, nodes, crossing creators, leaves
  u64 tot_all[4 * MAX_RANKS];
  TRY(x_host_gather(c, cm, tot_mine, sizeof tot_mine, tot_all));
  u64 creators_before = 0, clusters = 0, M = 0;
  C_x = tot_mine[2];
  for (u32 q = 0; q < P; q++) {
    const u64 cr = tot_all[4 * q + 3] - tot_all[4 * q];
    if (q < r) creators_before += cr;
    clusters += cr;
    M += tot_all[4 * q + 1];
    if (tot_all[4 * q + 2] != C_x) return fail(c, HUMID_E_INVALID, "internal: the ranks disagree on the crossing clusters (%llu vs %llu)", (ull)tot_all[4 * q + 2], (ull)C_x);
  }
  if (clusters >= (1ull << 31)) return fail(c, HUMID_E_OVERFLOW, "cluster ids exceed 31 bits");
  const u32 *xcid_all = nullptr;
  if (C_x) {
    ENSURE(c->xo_xcid, C_x * 4 + 16);
    hipLaunchKernelGGL(k_xcreator_ids, dim3(blocks_for(nw)), dim3(256), 0, st, br_xc, nw, (u32)goff, (u32)u_local, (u32)creators_before,
                       br_nc, c->xo_xcid.as<u32>());
    xcid_all = c->xo_xcid.as<u32>();
    if (moves) {
      ENSURE(c->xo_xcall, (size_t)P * C_x * 4 + 16);
      u64 cs[MAX_RANKS];
      for (u32 q = 0; q < P; q++) cs[q] = C_x;
      TRY(x_exchange(c, cm, c->xo_xcid.p, cs, true, c->xo_xcall.p, cs, 4));
      hipLaunchKernelGGL(k_max_rows, dim3(blocks_for(C_x)), dim3(256), 0, st, (const u32 *)c->xo_xcall.as<u32>(), P, (u32)C_x,
                         c->xo_xcid.as<u32>());
    }
  }
  const u32 *l_cid = nullptr;
  const u8 *l_ismax = nullptr;
  if (u_local) {
    ENSURE(c->x_lcid, (size_t)u_local * 4);
    ENSURE(c->x_lismax, (size_t)u_local);
    ENSURE(c->xo_ldeg, (size_t)u_local * 4);
    hipLaunchKernelGGL(k_own_results, dim3(blocks_for(u_local)), dim3(256), 0, st, br_in, br_nc, br_xc, xcid_all,
                       (const u32 *)c->cg_nodes.as<u32>(), (const u32 *)cg.cl_of, (const u32 *)cg.maxleaf, (const u32 *)cg.deg, (u32)goff,
                       (u32)u_local, (u32)creators_before, c->x_lcid.as<u32>(), c->x_lismax.as<u8>(), c->xo_ldeg.as<u32>(),
                       (const u32 *)c->s_first.as<u32>(), (const u32 *)c->s_slot.as<u32>(), c->slot_out.as<u64>());
    c->slots_done = true;                                            // (humid_stage_map_dense skips k_slot_results)
    HIPCHK(hipGetLastError());
    l_cid = c->x_lcid.as<u32>();
    l_ismax = c->x_lismax.as<u8>();
  }

  XT("ids");
  // ---- 6. per-read results at the owner, back to the home shards ----
  const u32 *packed = nullptr;
  u64 n_packed = 0;
  TRY(humid_stage_map_dense(c, l_cid, l_ismax, &packed, &n_packed));
  if (n_packed != n_recv) return fail(c, HUMID_E_INVALID, "map_dense returned %llu reads, %llu were counted", (ull)n_packed, (ull)n_recv);
  const u32 *ret = packed;
  if (moves && usable_all) {
    ENSURE(c->xr_ret, n_send * 4 + 8);
    TRY(x_exchange(c, cm, packed, recv_counts, false, c->xr_ret.p, send_counts, 4));
    ret = c->xr_ret.as<u32>();
  }
  // per read: its routed position -> its result (coalesced stores; filtered reads: cluster 0, not kept)
  if (n_local)
    hipLaunchKernelGGL(k_gather_results, dim3(grid_stride_blocks(n_local)), dim3(256), 0, st, (const u32 *)c->xo_inv.as<u32>(), ret,
                       (u32)n_send, (u32)n_local, d_cluster_id, d_keep);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(st));
  XT("return");
  if (xtrace)
    fprintf(stderr, "[xtrace] rank %u/%u reads %llu pairs: mine %llu interior %llu crossing %llu flagged %llu (all ranks) graph %llu nodes |%s\n", r, P,
            (ull)n_local, (ull)e_mine, (ull)n_int, (ull)X_total, (ull)K_total, (ull)M_mine, xt_line.c_str());
  if (summary) {
    *summary = gs;                                                   // the kernel times of the graph stage
    summary->total = total;
    summary->usable = usable;
    summary->unique = u_total;
    summary->clusters = clusters;
    summary->edges = E;
    summary->nonsingle = M;
    summary->ms_total = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
  }
  if (info) {
    info->unique_local = u_local;
    info->id_base = goff;
    info->n_nodes = M;
    info->n_pairs = E;
    info->d_unique_count = lc;
    info->d_unique_degree = u_local ? c->xo_ldeg.as<u32>() : nullptr;
  }
  return HUMID_OK;
}

// (host_all_gather through shared memory -- humid_shm_open / _all_gather / _abort / _close -- is a translation unit of
// its own without any HIP in it: shm.cpp)

int humid_at_least_double(humid_ctx *c, uint64_t a, uint64_t b, int *result) {
  if (!c || !result) return fail(c, HUMID_E_INVALID, "null argument");
  HIPCHK(hipSetDevice(c->device));
  ENSURE(c->scratch, 64);
  hipLaunchKernelGGL(k_at_least_double, dim3(1), dim3(1), 0, c->stream, a, b, c->scratch.as<int>());
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(result, c->scratch.p, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return HUMID_OK;
}

// ---- multi-GPU stages (device pointers; see humid_amd/sharded.py) ----------------------------
// Two-word words (33 <= word_nt <= 64; two uint64 per word, 16-byte aligned) in the stages of the ALL-GATHER mode
// (round 3: humid_stage_histogram, _count_dense, _unique, _graph, _graph_edges, _owner_perm): value ranges are ranges
// of HEADS -- the top 64 bits of a word's 2n-bit value -- as in the exchange pass, so the histogram and the
// splitters are those of 32-nt words over the heads.
static int stage_heads(humid_ctx *c, const u64 *d_words, u32 n, u32 word_nt, const u64 **heads) {
  if ((uintptr_t)d_words & 15) return fail(c, HUMID_E_INVALID, "wide words must be 16-byte aligned on the device");
  ENSURE(c->xr_heads, (size_t)n * 8 + 16);
  hipLaunchKernelGGL(k_wide_head64, dim3(blocks_for(n)), dim3(256), 0, c->stream, (const W2 *)d_words, n, 2 * (word_nt - 32),
                     c->xr_heads.as<u64>(), 0u);
  HIPCHK(hipGetLastError());
  *heads = c->xr_heads.as<u64>();
  return HUMID_OK;
}

int humid_stage_histogram(humid_ctx *c, const uint64_t *d_words, const uint8_t *d_filtered,
                          uint64_t n_reads, uint32_t word_nt, uint32_t bits, uint32_t *d_hist) {
  if (!c) return fail(nullptr, HUMID_E_INVALID, "ctx is null");
  TRY(check_run_args(c, n_reads, word_nt, 0, 64));
  if (bits == 0 || bits > 12 || bits > 2 * word_nt || !d_hist) return fail(c, HUMID_E_INVALID, "bits must be 1..min(12, 2*word_nt)");
  HIPCHK(hipSetDevice(c->device));
  const u32 n_bins = 1u << bits;
  HIPCHK(hipMemsetAsync(d_hist, 0, n_bins * 4, c->stream));
  if (n_reads) {
    const u64 *keys = d_words;
    if (word_nt > 32) TRY(stage_heads(c, d_words, (u32)n_reads, word_nt, &keys));
    hipLaunchKernelGGL(k_top_hist, dim3(256), dim3(1024), n_bins * 4, c->stream, keys, d_filtered,
                       (u32)n_reads, (u64)0, word_nt >= 32 ? (u64)1 : ((u64)1 << (64 - 2 * word_nt)), bits, d_hist);
  }
  HIPCHK(hipGetLastError());
  return HUMID_OK;              // queued on the context's stream; no host value is returned
}

int humid_stage_count(humid_ctx *c, const uint64_t *d_words, const uint8_t *d_filtered, uint64_t n_reads,
                      uint32_t word_nt, uint64_t range_lo, uint64_t range_hi, uint64_t expected_reads,
                      uint64_t *n_unique, uint64_t *n_usable) {
  if (!c) return fail(nullptr, HUMID_E_INVALID, "ctx is null");
  c->have_run = c->have_graph = false;
  c->graph_mode = false;
  c->dense_mode = false;
  TRY(check_run_args(c, n_reads, word_nt, 0));
  if (n_reads && (!d_words || !d_filtered)) return fail(c, HUMID_E_INVALID, "null buffer");
  HIPCHK(hipSetDevice(c->device));
  humid_summary s;
  memset(&s, 0, sizeof s);
  c->N = n_reads; c->U = c->E = c->M = c->C = c->usable = 0;
  c->word_nt = word_nt;
  if (n_reads) TRY(stage_count(c, d_words, d_filtered, (u32)n_reads, word_nt, range_lo, range_hi, expected_reads, s));
  HIPCHK(hipStreamSynchronize(c->stream));
  if (n_unique) *n_unique = c->U;
  if (n_usable) *n_usable = c->usable;
  return HUMID_OK;
}

// Dense variant for a multi-GPU rank: the usable reads of [range_lo, range_hi] are first compacted
// (in read order) and then counted by the LDS-partitioned path like a single-GPU read set.  The
// dense order IS the order of the per-shard result streams (humid_stage_map_dense), and
// counts[q] = owned reads in [shard_begin[q], shard_begin[q+1]) are the all-to-all split sizes.
int humid_stage_count_dense(humid_ctx *c, const uint64_t *d_words, const uint8_t *d_filtered,
                            uint64_t n_reads, uint32_t word_nt, uint64_t range_lo, uint64_t range_hi,
                            const uint64_t *shard_begin, uint32_t n_shards, uint64_t *counts,
                            uint64_t *n_unique, uint64_t *n_usable) {
  if (!c) return fail(nullptr, HUMID_E_INVALID, "ctx is null");
  c->have_run = c->have_graph = false;
  c->graph_mode = false;
  c->dense_mode = false;
  TRY(check_run_args(c, n_reads, word_nt, 0, 64));
  const bool wide = word_nt > 32;
  if (wide && d_filtered == nullptr)
    return fail(c, HUMID_E_UNSUPPORTED, "two-word words in the stage-by-stage exchange form: use humid_dedup_run_exchange");
  if (!shard_begin || !counts || n_shards == 0 || n_shards > 4096) return fail(c, HUMID_E_INVALID, "bad argument");
  // d_filtered == NULL: every read is usable and lies in [range_lo, range_hi] (exchange mode: the
  // reads were routed here because they do); the array is counted as it stands, no compaction
  // pass, and the range only shapes the word-ordered buckets
  const bool all_owned = d_filtered == nullptr;
  if (n_reads && (!d_words || (!d_filtered && !all_owned))) return fail(c, HUMID_E_INVALID, "null buffer");
  HIPCHK(hipSetDevice(c->device));
  hipStream_t st = c->stream;
  const u32 N = (u32)n_reads;
  for (u32 q = 0; q <= n_shards; q++)
    if (shard_begin[q] > N || (q && shard_begin[q] < shard_begin[q - 1])) return fail(c, HUMID_E_INVALID, "shard_begin must ascend within [0, n_reads]");
  if (shard_begin[0] != 0 || shard_begin[n_shards] != N) return fail(c, HUMID_E_INVALID, "shards must cover [0, n_reads)");
  humid_summary s;
  memset(&s, 0, sizeof s);
  c->N = c->U = c->E = c->M = c->C = c->usable = 0;
  c->word_nt = word_nt;
  for (u32 q = 0; q < n_shards; q++) counts[q] = 0;
  if (n_unique) *n_unique = 0;
  if (n_usable) *n_usable = 0;
  c->dense_mode = true;
  c->stage_map_timed = false;
  if (N == 0) return HUMID_OK;
  if (all_owned) {
    for (u32 q = 0; q < n_shards; q++) counts[q] = shard_begin[q + 1] - shard_begin[q];
    c->N = N;
    TRY(stage_count(c, d_words, nullptr, N, word_nt, range_lo, range_hi, 0, s, true));
    HIPCHK(hipStreamSynchronize(st));
    if (c->usable != N) return fail(c, HUMID_E_INVALID, "a read outside [range_lo, range_hi] in an all-owned count");
    if (n_unique) *n_unique = c->U;
    if (n_usable) *n_usable = c->usable;
    return HUMID_OK;
  }
  ENSURE(c->opos, ((size_t)N + 1) * 4);
  const u64 *range_keys = d_words;                      // what the range is a range of: the words, or their heads
  if (wide) TRY(stage_heads(c, d_words, N, word_nt, &range_keys));
  {
    ComposeIn<OwnedRangeFlagOp, IotaIn> fin{OwnedRangeFlagOp{range_keys, d_filtered, range_lo, range_hi, N}, IotaIn{}};
    TRY(exscan_in<u32>(c, fin, c->opos.as<u32>(), (u64)N + 1));
  }
  std::vector<u32> got(n_shards + 1);
  for (u32 q = 0; q <= n_shards; q++)
    HIPCHK(hipMemcpyAsync(&got[q], c->opos.as<u32>() + shard_begin[q], 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  const u32 n_own = got[n_shards];
  for (u32 q = 0; q < n_shards; q++) counts[q] = got[q + 1] - got[q];
  c->N = n_own;
  if (n_own == 0) return HUMID_OK;
  ENSURE(c->own_words, (size_t)n_own * (wide ? 16 : 8));
  if (wide) {
    hipLaunchKernelGGL(k_gather_owned_w2, dim3(grid_stride_blocks(N)), dim3(256), 0, st, (const W2 *)d_words, range_keys, d_filtered,
                       (const u32 *)c->opos.as<u32>(), range_lo, range_hi, N, c->own_words.as<W2>());
    HIPCHK(hipGetLastError());
    TRY(stage_count_wide(c, c->own_words.as<W2>(), nullptr, n_own, word_nt, s));
    HIPCHK(hipStreamSynchronize(st));
    if (n_unique) *n_unique = c->U;
    if (n_usable) *n_usable = c->usable;
    return HUMID_OK;
  }
  hipLaunchKernelGGL(k_gather_owned, dim3(grid_stride_blocks(N)), dim3(256), 0, st, d_words, d_filtered,
                     c->opos.as<u32>(), range_lo, range_hi, N, c->own_words.as<u64>());
  HIPCHK(hipGetLastError());
  TRY(stage_count(c, c->own_words.as<u64>(), nullptr, n_own, word_nt, 0ull, ~0ull, 0, s));
  HIPCHK(hipStreamSynchronize(st));
  if (n_unique) *n_unique = c->U;
  if (n_usable) *n_usable = c->usable;
  return HUMID_OK;
}

// The result stream of the dense variant: packed (cluster_id | keep << 31) of this rank's reads in
// dense (= read) order, n = sum of the counts humid_stage_count_dense returned.
int humid_stage_map_dense(humid_ctx *c, const uint32_t *d_local_cluster_id, const uint8_t *d_local_is_max,
                          const uint32_t **d_packed, uint64_t *n_packed) {
  if (!c) return fail(nullptr, HUMID_E_INVALID, "ctx is null");
  if (!c->dense_mode) return fail(c, HUMID_E_STATE, "no preceding humid_stage_count_dense");
  if (!d_packed || !n_packed) return fail(c, HUMID_E_INVALID, "bad argument");
  HIPCHK(hipSetDevice(c->device));
  hipStream_t st = c->stream;
  const u32 N = (u32)c->N, U = (u32)c->U;
  *d_packed = nullptr;
  *n_packed = N;
  if (N == 0) return HUMID_OK;
  if (U && (!d_local_cluster_id || !d_local_is_max)) return fail(c, HUMID_E_INVALID, "null buffer");
  if (U > 0 && !c->slots_done)                         // (slots_done: the caller's id kernel wrote the slot results itself)
    hipLaunchKernelGGL(k_slot_results, dim3(blocks_for(U)), dim3(256), 0, st, d_local_cluster_id, d_local_is_max,
                       c->s_first.as<u32>(), c->s_slot.as<u32>(), U, c->slot_out.as<u64>());
  c->slots_done = false;
  ENSURE(c->own_packed, ((size_t)N + 1) * 4);
  if (c->kev_on) HIPCHK(hipEventRecord(c->kev[37], st));
  bool tiled = false;
  if (c->last_count_lds) TRY(unpermute_tiled(c, N, true, c->own_packed.as<u32>(), (u8 *)nullptr, c->kev[42], &tiled));
  if (tiled) {
    // both kernels of the un-permute are inside kev[37]..kev[38]
  } else if (c->last_count_lds && c->n_parts && !c->last_count_sorted) {
    HIPCHK(hipMemsetAsync(c->own_packed.p, 0, (size_t)N * 4, st));
    hipLaunchKernelGGL(k_read_map_bucket, dim3(c->n_parts), dim3(256), 0, st, c->pk_vals.as<u32>(),
                       c->pslot.as<u32>(), c->slot_out.as<u64>(), c->pbeg.as<u32>(), c->ucount.as<u32>(), N,
                       c->own_packed.as<u32>());
  } else if (c->last_count_lds) {
    HIPCHK(hipMemsetAsync(c->own_packed.p, 0, (size_t)N * 4, st));
    hipLaunchKernelGGL(k_read_map_part, dim3(grid_stride_blocks(N)), dim3(256), 0, st, c->pk_vals.as<u32>(),
                       c->pslot.as<u32>(), c->slot_out.as<u64>(), N, c->own_packed.as<u32>());
  } else
    hipLaunchKernelGGL(k_read_map_packed, dim3(grid_stride_blocks(N)), dim3(256), 0, st, c->slot_of_read.as<u32>(),
                       c->slot_out.as<u64>(), N, c->own_packed.as<u32>());
  if (c->kev_on) HIPCHK(hipEventRecord(c->kev[38], st));
  c->stage_map_timed = true;
  HIPCHK(hipGetLastError());
  *d_packed = c->own_packed.as<u32>();     // queued on the context's stream
  return HUMID_OK;
}

int humid_stage_unique(humid_ctx *c, const uint64_t **d_word, const uint32_t **d_count,
                       const uint32_t **d_first) {
  if (!c) return fail(nullptr, HUMID_E_INVALID, "ctx is null");
  if (d_word) *d_word = c->U ? c->s_word.as<u64>() : nullptr;
  if (d_count) *d_count = c->U ? c->s_cnt.as<u32>() : nullptr;
  if (d_first) *d_first = c->U ? c->s_first.as<u32>() : nullptr;
  return HUMID_OK;
}

int humid_stage_graph(humid_ctx *c, const uint64_t *d_g_word, const uint32_t *d_g_count,
                      uint64_t n_unique, uint32_t word_nt, uint32_t distance, uint32_t method,
                      const uint32_t **d_cluster_id, const uint8_t **d_is_max, humid_summary *summary) {
  if (!c) return fail(nullptr, HUMID_E_INVALID, "ctx is null");
  c->have_graph = false;
  c->graph_mode = false;
  TRY(check_run_args(c, n_unique, word_nt, method, 64));
  HIPCHK(hipSetDevice(c->device));
  humid_summary s;
  memset(&s, 0, sizeof s);
  s.unique = n_unique;
  c->distance = distance; c->method = method;
  c->gU = 0; c->E = c->M = c->C = 0;
  if (d_cluster_id) *d_cluster_id = nullptr;
  if (d_is_max) *d_is_max = nullptr;
  if (n_unique) {
    if (!d_g_word || !d_g_count) return fail(c, HUMID_E_INVALID, "null buffer");
    u32 nps = 0;
    if (word_nt > 32) {
      if ((uintptr_t)d_g_word & 15) return fail(c, HUMID_E_INVALID, "wide words must be 16-byte aligned on the device");
      TRY(stage_graph<W2>(c, (const W2 *)d_g_word, d_g_count, (u32)n_unique, word_nt, distance, method, s, nps));
    } else
    TRY(stage_graph(c, d_g_word, d_g_count, (u32)n_unique, word_nt, distance, method, s, nps));
    TRY(n_clusters_from_scan(c, (u32)n_unique, &c->C));
    s.clusters = c->C;
    if (d_cluster_id) *d_cluster_id = c->cid.as<u32>();
    if (d_is_max) *d_is_max = c->ismax.as<u8>();
  }
  HIPCHK(hipStreamSynchronize(c->stream));
  if (summary) *summary = s;
  c->have_graph = true;
  return HUMID_OK;
}

int humid_stage_map(humid_ctx *c, const uint32_t *d_local_cluster_id, const uint8_t *d_local_is_max,
                    uint64_t n_reads, uint32_t *d_cluster_id, uint8_t *d_keep) {
  if (!c) return fail(nullptr, HUMID_E_INVALID, "ctx is null");
  if (n_reads != c->N) return fail(c, HUMID_E_STATE, "n_reads differs from the preceding humid_stage_count");
  if (n_reads && (!d_cluster_id || !d_keep)) return fail(c, HUMID_E_INVALID, "null buffer");
  if (c->U && (!d_local_cluster_id || !d_local_is_max)) return fail(c, HUMID_E_INVALID, "null buffer");
  HIPCHK(hipSetDevice(c->device));
  if (n_reads) TRY(stage_map(c, d_local_cluster_id, d_local_is_max, (u32)n_reads, d_cluster_id, d_keep));
  HIPCHK(hipStreamSynchronize(c->stream));
  return HUMID_OK;
}

int humid_stage_pairs(humid_ctx *c, const uint64_t *d_g_word, uint64_t n_unique, uint32_t word_nt,
                      uint32_t distance, uint32_t part_rank, uint32_t part_world, const uint64_t **d_edges,
                      uint64_t *n_edges) {
  if (!c) return fail(nullptr, HUMID_E_INVALID, "ctx is null");
  if (!d_edges || !n_edges || part_world == 0 || part_rank >= part_world) return fail(c, HUMID_E_INVALID, "bad argument");
  TRY(check_run_args(c, n_unique, word_nt, 0));
  HIPCHK(hipSetDevice(c->device));
  *d_edges = nullptr;
  *n_edges = 0;
  if (n_unique && !d_g_word) return fail(c, HUMID_E_INVALID, "null buffer");
  u64 E = 0;
  if (n_unique) TRY(stage_pairs_share(c, d_g_word, (u32)n_unique, word_nt, distance, part_rank, part_world, &E));
  HIPCHK(hipStreamSynchronize(c->stream));
  *n_edges = E;
  *d_edges = E ? c->share_edges.as<u64>() : nullptr;
  return HUMID_OK;
}

int humid_stage_graph_edges(humid_ctx *c, const uint64_t *d_g_word, const uint32_t *d_g_count, uint64_t n_unique,
                            const uint64_t *d_edges, uint64_t n_edges, uint32_t word_nt, uint32_t distance,
                            uint32_t method, const uint32_t **d_cluster_id, const uint8_t **d_is_max,
                            humid_summary *summary) {
  if (!c) return fail(nullptr, HUMID_E_INVALID, "ctx is null");
  c->have_graph = false;
  c->graph_mode = false;
  TRY(check_run_args(c, n_unique, word_nt, method, 64));
  if (n_edges >= 0x7fffffffull) return fail(c, HUMID_E_OVERFLOW, "2*edges exceeds 32 bits");
  HIPCHK(hipSetDevice(c->device));
  humid_summary s;
  memset(&s, 0, sizeof s);
  s.unique = n_unique;
  c->distance = distance; c->method = method;
  c->gU = 0; c->E = c->M = c->C = 0;
  if (d_cluster_id) *d_cluster_id = nullptr;
  if (d_is_max) *d_is_max = nullptr;
  if (n_unique) {
    if (!d_g_word || !d_g_count || (n_edges && !d_edges)) return fail(c, HUMID_E_INVALID, "null buffer");
    u32 nps = 0;
    static const u64 no_edges = 0;
    if (word_nt > 32) {
      if ((uintptr_t)d_g_word & 15) return fail(c, HUMID_E_INVALID, "wide words must be 16-byte aligned on the device");
      TRY(stage_graph<W2>(c, (const W2 *)d_g_word, d_g_count, (u32)n_unique, word_nt, distance, method, s, nps,
                          n_edges ? d_edges : &no_edges, n_edges));
    } else
    TRY(stage_graph(c, d_g_word, d_g_count, (u32)n_unique, word_nt, distance, method, s, nps,
                    n_edges ? d_edges : &no_edges, n_edges));
    TRY(n_clusters_from_scan(c, (u32)n_unique, &c->C));
    s.clusters = c->C;
    if (d_cluster_id) *d_cluster_id = c->cid.as<u32>();
    if (d_is_max) *d_is_max = c->ismax.as<u8>();
  }
  HIPCHK(hipStreamSynchronize(c->stream));
  if (summary) *summary = s;
  c->have_graph = true;
  return HUMID_OK;
}

// ---- multi-GPU exchange mode (humid_amd/sharded.py, mode "exchange") -------------------------
// Words travel to the rank that owns their VALUE range (all-to-all) instead of every word to every
// rank; each rank counts its range, and for every non-prefix combination the unique words travel
// once more, to the rank that owns their combination key.  Pairs carry global unique indices.
static u32 min_prefix_bits(u32 n, u32 d, u32 force_segments) {
  if (d >= n) return 0;
  u32 best = ~0u;
  for (u32 sgm = d + 1; sgm <= n && sgm <= d + MAX_FIELDS; sgm++) {
    if (n_choose_k(sgm, sgm - d) > MAX_COMBOS) break;
    if (force_segments && sgm != force_segments) continue;
    const u32 base = n / sgm, rem = n % sgm, k = sgm - d;
    u32 len = 0;
    for (u32 t = 0; t < k; t++) len += base + (t < rem ? 1 : 0);
    if (2 * len < best) best = 2 * len;
  }
  if (best == ~0u) best = 2 * (n / (d + 1));     // forced s not legal: make_plan falls back to d + 1
  return best > 64 ? 64 : best;
}

int humid_stage_plan_info(humid_ctx *c, uint32_t word_nt, uint32_t distance, uint64_t plan_unique,
                          uint32_t *n_combos, uint32_t *prefix_bits) {
  // pure host arithmetic: ctx may be NULL (no GPU needed; the automatic plan is reported)
  TRY(check_run_args(c, 0, word_nt, 0, 64));
  const u32 force = c ? c->force_segments : 0u;
  const ComboPlan plan = make_plan(word_nt, distance, plan_unique, force);
  if (plan.ncombo == 0 || plan.ncombo > MAX_COMBOS) return fail(c, HUMID_E_INVALID, "internal: bad pigeonhole plan");
  if (n_combos) *n_combos = plan.ncombo;
  if (prefix_bits) {
    const u32 mp = min_prefix_bits(word_nt, distance, force);
    const u32 mbits = (u32)__builtin_popcountll(plan.mask[0].lo) + (u32)__builtin_popcountll(plan.mask[0].hi);   // (.hi: two-word words)
    *prefix_bits = mp < mbits ? mp : mbits;
  }
  return HUMID_OK;
}

static ComboFields plan_fields(const ComboPlan &plan, u32 cb) {
  ComboFields cf;
  cf.nf = plan.nfield[cb];
  for (u32 f = 0; f < MAX_FIELDS; f++) { cf.shift[f] = plan.shift[cb][f]; cf.width[f] = plan.width[cb][f]; }
  return cf;
}

int humid_stage_combo_route(humid_ctx *c, const uint64_t *d_word, const uint32_t *d_count, uint64_t n_unique,
                            uint64_t id_base, uint32_t word_nt, uint32_t distance, uint64_t plan_unique, uint32_t combo,
                            uint32_t n_ranks, const uint64_t **d_items, uint64_t *counts) {
  if (!c) return fail(nullptr, HUMID_E_INVALID, "ctx is null");
  if (!d_items || !counts || n_ranks == 0) return fail(c, HUMID_E_INVALID, "bad argument");
  if (n_ranks > 255) return fail(c, HUMID_E_UNSUPPORTED, "more than 255 ranks");
  TRY(check_run_args(c, n_unique, word_nt, 0));
  if (id_base + n_unique > 0xffffffffull) return fail(c, HUMID_E_OVERFLOW, "global unique index exceeds 32 bits");
  HIPCHK(hipSetDevice(c->device));
  hipStream_t st = c->stream;
  const ComboPlan plan = make_plan(word_nt, distance, plan_unique, c->force_segments);
  if (combo >= plan.ncombo) return fail(c, HUMID_E_INVALID, "combo %u out of range (%u)", combo, plan.ncombo);
  *d_items = nullptr;
  for (u32 q = 0; q < n_ranks; q++) counts[q] = 0;
  const u32 n = (u32)n_unique;
  if (n == 0) return HUMID_OK;
  if (!d_word) return fail(c, HUMID_E_INVALID, "null buffer");
  ENSURE(c->x_items, (size_t)n * 16);
  if (n_ranks == 1) {                                  // everything stays here: no owners, no sort, no host wait
    hipLaunchKernelGGL(k_route_items, dim3(blocks_for(n)), dim3(256), 0, st, d_word, d_count, (const u32 *)nullptr, n,
                       (u64)id_base, c->x_items.as<ulonglong2>());
    HIPCHK(hipGetLastError());
    counts[0] = n;
    *d_items = c->x_items.as<u64>();
    return HUMID_OK;
  }
  ENSURE(c->owner, (size_t)n);
  ENSURE(c->owner_sorted, (size_t)n);
  ENSURE(c->x_ids, (size_t)n * 4);
  hipLaunchKernelGGL(k_combo_owner<u64>, dim3(blocks_for(n)), dim3(256), 0, st, (const u64 *)d_word, n, plan_fields(plan, combo),
                     n_ranks, c->owner.as<u8>());
  {
    TRY((sort_pairs_in<u8, u32>(c, PtrIn<u8>{c->owner.as<u8>()}, c->owner_sorted.as<u8>(), IotaIn{}, c->x_ids.as<u32>(), n, 0, 8)));
  }
  ENSURE(c->small, (size_t)(n_ranks + 2) * 4);
  hipLaunchKernelGGL(k_owner_bounds, dim3(1), dim3(256), 0, st, c->owner_sorted.as<u8>(), n, n_ranks,
                     c->small.as<u32>());
  hipLaunchKernelGGL(k_route_items, dim3(blocks_for(n)), dim3(256), 0, st, d_word, d_count, c->x_ids.as<u32>(), n,
                     (u64)id_base, c->x_items.as<ulonglong2>());
  std::vector<u32> b(n_ranks + 2);
  HIPCHK(hipMemcpyAsync(b.data(), c->small.p, (n_ranks + 2) * 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(st));
  for (u32 q = 0; q < n_ranks; q++) counts[q] = b[q + 1] - b[q];
  *d_items = c->x_items.as<u64>();
  return HUMID_OK;
}

// pairs among W[0, n) walked in bucket order of combination cb -> c->share_edges, as
// (V[i] << 32 | V[j]) ordered by value; V == null: positions themselves
extern "C++" {
template <class WT>
static int emit_pairs(humid_ctx *c, const WT *W, const u32 *V, u32 n, const ComboPlan &plan, u32 cb,
                      u32 distance, u64 *E_out) {
  hipStream_t st = c->stream;
  *E_out = 0;
  EarlierMasksT<WT> d_masks;
  for (u32 t = 0; t < MAX_COMBOS; t++) d_masks.m[t] = w_from<WT>(plan.mask[t]);
  const WT cmask = w_from<WT>(plan.mask[cb]);
  ENSURE(c->pc, ((size_t)n + 1) * 4);
  ENSURE(c->poff, ((size_t)n + 1) * 4);
  HIPCHK(hipMemsetAsync(c->pc.as<u32>() + n, 0, 4, st));
  HIPCHK(hipMemsetAsync(&c->d_ctr[CTR_BIGMASK], 0, sizeof(ull), st));
  // the walk of a position is bounded as on one GPU; buckets beyond it are finished as tiles below
  const u32 walk_max = c->walk_max;
  const dim3 grid(blocks_for(n)), blk(256);
  if (V)
    hipLaunchKernelGGL((k_pairs<false, PM_EMIT_COUNT, WT>), grid, blk, 0, st, W, V, n, 0u, n, cmask, d_masks,
                       cb, distance, (u32 *)nullptr, (u32 *)nullptr, (const u32 *)nullptr, (u32 *)nullptr,
                       (u32 *)nullptr, c->pc.as<u32>(), c->poff.as<u32>(), c->share_edges.as<u64>(), (u32 *)nullptr, walk_max,
                       &c->d_ctr[CTR_BIGMASK]);
  else
    hipLaunchKernelGGL((k_pairs<true, PM_EMIT_COUNT, WT>), grid, blk, 0, st, W, V, n, 0u, n, cmask, d_masks,
                       cb, distance, (u32 *)nullptr, (u32 *)nullptr, (const u32 *)nullptr, (u32 *)nullptr,
                       (u32 *)nullptr, c->pc.as<u32>(), c->poff.as<u32>(), c->share_edges.as<u64>(), (u32 *)nullptr, walk_max,
                       &c->d_ctr[CTR_BIGMASK]);
  TRY(exscan_u32(c, c->pc.as<u32>(), c->poff.as<u32>(), (u64)n + 1));
  HIPCHK(hipGetLastError());
  TRY(read_counters(c, c->poff.as<u32>() + n));
  u64 E = c->h_ctr[CTR_N - 1] & 0xffffffffull;
  // pairs further apart than the walk inside large buckets: counted, then appended behind the others
  std::vector<BigRun> runs;
  const BigRun *d_runs = nullptr;
  u64 E_far = 0;
  ull tiles = 0;
  if (c->h_ctr[CTR_BIGMASK]) {
    TRY(find_big_runs<WT>(c, W, n, cmask, walk_max, 0, runs, &d_runs));
    tiles = runs.back().tile0;
  }
  const u32 tgrid = (u32)std::min<ull>(tiles ? tiles : 1, 1u << 20);
#define EMIT_TILES(P0, M)                                                                                              \
  hipLaunchKernelGGL((k_pairs_tiles<P0, M, WT>), dim3(tgrid), dim3(PT2_THREADS), 0, st, W, V, d_runs, (u32)runs.size() - 1, \
                     tiles, d_masks, cb, distance, walk_max, (u32 *)nullptr, (u32 *)nullptr, (const u32 *)nullptr,         \
                     (u32 *)nullptr, (u32 *)nullptr, (const u32 *)nullptr, c->share_edges.as<u64>(), &c->d_ctr[CTR_SPECIAL])
  if (tiles) {
    HIPCHK(hipMemsetAsync(&c->d_ctr[CTR_SPECIAL], 0, sizeof(ull), st));
    if (V) EMIT_TILES(false, PM_EMIT_COUNT); else EMIT_TILES(true, PM_EMIT_COUNT);
    HIPCHK(hipGetLastError());
    TRY(read_counters(c));
    E_far = c->h_ctr[CTR_SPECIAL];
  }
  if (E + E_far > 0xffffffffull) return fail(c, HUMID_E_OVERFLOW, "%llu neighbour pairs in one share", (ull)(E + E_far));
  *E_out = E + E_far;
  if (E + E_far == 0) return HUMID_OK;
  ENSURE(c->share_edges, (size_t)(E + E_far) * 8);
  if (E) {
    if (V)
      hipLaunchKernelGGL((k_pairs<false, PM_EMIT_FILL, WT>), grid, blk, 0, st, W, V, n, 0u, n, cmask, d_masks,
                         cb, distance, (u32 *)nullptr, (u32 *)nullptr, (const u32 *)nullptr, (u32 *)nullptr,
                         (u32 *)nullptr, c->pc.as<u32>(), c->poff.as<u32>(), c->share_edges.as<u64>(), (u32 *)nullptr, walk_max);
    else
      hipLaunchKernelGGL((k_pairs<true, PM_EMIT_FILL, WT>), grid, blk, 0, st, W, V, n, 0u, n, cmask, d_masks,
                         cb, distance, (u32 *)nullptr, (u32 *)nullptr, (const u32 *)nullptr, (u32 *)nullptr,
                         (u32 *)nullptr, c->pc.as<u32>(), c->poff.as<u32>(), c->share_edges.as<u64>(), (u32 *)nullptr, walk_max);
  }
  if (E_far) {
    const ull at = E;                                               // the cursor of the append starts behind k_pairs' pairs
    HIPCHK(hipMemcpyAsync(&c->d_ctr[CTR_SPECIAL], &at, sizeof(ull), hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));                               // (`at` is a host temporary)
    if (V) EMIT_TILES(false, PM_EMIT_FILL); else EMIT_TILES(true, PM_EMIT_FILL);
  }
#undef EMIT_TILES
  HIPCHK(hipGetLastError());
  return HUMID_OK;
}
}  // extern "C++"

// ---- two-word (wide) words in the exchange pass: items of 24 bytes (hi, lo, id | count << 32) ----
// humid_stage_combo_route for W2: this rank's unique words in destination-major order
static int combo_route_wide(humid_ctx *c, const W2 *d_word, const u32 *d_count, u32 n, u64 id_base, const ComboPlan &plan,
                            u32 combo, u32 n_ranks, const Item3 **d_items, u64 *counts) {
  hipStream_t st = c->stream;
  *d_items = nullptr;
  for (u32 q = 0; q < n_ranks; q++) counts[q] = 0;
  if (n == 0) return HUMID_OK;
  ENSURE(c->x_items, (size_t)n * sizeof(Item3));
  if (n_ranks == 1) {
    hipLaunchKernelGGL(k_route_items_w2, dim3(blocks_for(n)), dim3(256), 0, st, d_word, d_count, (const u32 *)nullptr, n, id_base,
                       c->x_items.as<Item3>());
    HIPCHK(hipGetLastError());
    counts[0] = n;
    *d_items = c->x_items.as<Item3>();
    return HUMID_OK;
  }
  ENSURE(c->owner, (size_t)n);
  ENSURE(c->owner_sorted, (size_t)n);
  ENSURE(c->x_ids, (size_t)n * 4);
  hipLaunchKernelGGL(k_combo_owner<W2>, dim3(blocks_for(n)), dim3(256), 0, st, d_word, n, plan_fields(plan, combo), n_ranks,
                     c->owner.as<u8>());
  {
    TRY((sort_pairs_in<u8, u32>(c, PtrIn<u8>{c->owner.as<u8>()}, c->owner_sorted.as<u8>(), IotaIn{}, c->x_ids.as<u32>(), n, 0, 8)));
  }
  ENSURE(c->small, (size_t)(n_ranks + 2) * 4);
  hipLaunchKernelGGL(k_owner_bounds, dim3(1), dim3(256), 0, st, c->owner_sorted.as<u8>(), n, n_ranks, c->small.as<u32>());
  hipLaunchKernelGGL(k_route_items_w2, dim3(blocks_for(n)), dim3(256), 0, st, d_word, d_count, c->x_ids.as<u32>(), n, id_base,
                     c->x_items.as<Item3>());
  std::vector<u32> b(n_ranks + 2);
  HIPCHK(hipMemcpyAsync(b.data(), c->small.p, (n_ranks + 2) * 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(st));
  for (u32 q = 0; q < n_ranks; q++) counts[q] = b[q + 1] - b[q];
  *d_items = c->x_items.as<Item3>();
  return HUMID_OK;
}

// humid_stage_pairs_keyed for W2.  items: received Item3 records (interleaved) or, for combination 0, the
// plain ascending W2 array with ids id_base + index and counts d_count
static int pairs_keyed_wide(humid_ctx *c, const void *d_items, u32 n, bool interleaved, u64 id_base, const u32 *d_count,
                            const ComboPlan &plan, u32 combo, u32 distance, const u64 **d_records, u64 *n_edges) {
  hipStream_t st = c->stream;
  *d_records = nullptr;
  *n_edges = 0;
  if (n < 2 || distance == 0) return HUMID_OK;
  u64 E = 0;
  const u32 *id_of = nullptr, *cnt_of = d_count;
  if (!interleaved) {
    TRY(emit_pairs<W2>(c, (const W2 *)d_items, nullptr, n, plan, 0, distance, &E));
  } else {
    ENSURE(c->x_w, (size_t)n * sizeof(W2));
    ENSURE(c->x_id, (size_t)n * 4);
    ENSURE(c->x_cnt, (size_t)n * 4);
    ENSURE(c->seg_k0, (size_t)n * 8);
    ENSURE(c->seg_v0, (size_t)n * 4);
    ENSURE(c->seg_ks, (size_t)n * 8);
    ENSURE(c->seg_vs, (size_t)n * 4);
    ENSURE(c->seg_ws, (size_t)n * sizeof(W2));
    hipLaunchKernelGGL(k_split_items_w2, dim3(blocks_for(n)), dim3(256), 0, st, (const Item3 *)d_items, n, c->x_w.as<W2>(),
                       c->x_id.as<u32>(), c->x_cnt.as<u32>());
    const u32 kb = plan.key_bits ? plan.key_bits : 1;
    if (kb <= 32) {
      hipLaunchKernelGGL((k_combo_keys<u32, W2>), dim3(blocks_for(n)), dim3(256), 0, st, c->x_w.as<W2>(), n, plan_fields(plan, combo),
                         c->seg_k0.as<u32>(), c->seg_v0.as<u32>());
      TRY(sort_pairs<u32, u32>(c, c->seg_k0.as<u32>(), c->seg_ks.as<u32>(), c->seg_v0.as<u32>(), c->seg_vs.as<u32>(), n, 0, kb));
    } else {
      hipLaunchKernelGGL((k_combo_keys<u64, W2>), dim3(blocks_for(n)), dim3(256), 0, st, c->x_w.as<W2>(), n, plan_fields(plan, combo),
                         c->seg_k0.as<u64>(), c->seg_v0.as<u32>());
      TRY(sort_pairs<u64, u32>(c, c->seg_k0.as<u64>(), c->seg_ks.as<u64>(), c->seg_v0.as<u32>(), c->seg_vs.as<u32>(), n, 0, kb));
    }
    hipLaunchKernelGGL(k_gather_bucket_words<W2>, dim3(blocks_for(n)), dim3(256), 0, st, c->x_w.as<W2>(), c->seg_vs.as<u32>(), n,
                       c->seg_ws.as<W2>());
    TRY(emit_pairs<W2>(c, c->seg_ws.as<W2>(), c->seg_vs.as<u32>(), n, plan, combo, distance, &E));
    id_of = c->x_id.as<u32>();
    cnt_of = c->x_cnt.as<u32>();
  }
  *n_edges = E;
  if (E == 0) return HUMID_OK;
  ENSURE(c->x_rec, (size_t)E * 16);
  hipLaunchKernelGGL(k_edge_records, dim3(blocks_for(E)), dim3(256), 0, st, c->share_edges.as<u64>(), (u32)E, id_of, (u32)id_base,
                     cnt_of, c->x_rec.as<ulonglong2>());
  HIPCHK(hipGetLastError());
  *d_records = c->x_rec.as<u64>();
  return HUMID_OK;
}


int humid_stage_pairs_keyed(humid_ctx *c, const uint64_t *d_items, uint64_t n_items, int interleaved,
                            uint64_t id_base, const uint32_t *d_count, uint32_t word_nt, uint32_t distance,
                            uint64_t plan_unique, uint32_t combo, const uint64_t **d_records, uint64_t *n_edges) {
  if (!c) return fail(nullptr, HUMID_E_INVALID, "ctx is null");
  if (!d_records || !n_edges) return fail(c, HUMID_E_INVALID, "bad argument");
  TRY(check_run_args(c, n_items, word_nt, 0));
  HIPCHK(hipSetDevice(c->device));
  hipStream_t st = c->stream;
  *d_records = nullptr;
  *n_edges = 0;
  const u32 n = (u32)n_items;
  if (n < 2 || distance == 0) return HUMID_OK;
  if (!d_items) return fail(c, HUMID_E_INVALID, "null buffer");
  const ComboPlan plan = make_plan(word_nt, distance, plan_unique, c->force_segments);
  if (combo >= plan.ncombo) return fail(c, HUMID_E_INVALID, "combo %u out of range (%u)", combo, plan.ncombo);
  if (!interleaved && combo != 0) return fail(c, HUMID_E_INVALID, "a plain word array is in bucket order for combination 0 only");
  if (!interleaved && id_base + n_items > 0xffffffffull) return fail(c, HUMID_E_OVERFLOW, "global unique index exceeds 32 bits");
  u64 E = 0;
  const u32 *id_of = nullptr, *cnt_of = d_count;
  if (!interleaved) {
    TRY(emit_pairs(c, d_items, nullptr, n, plan, 0, distance, &E));     // pairs of positions
  } else {
    ENSURE(c->x_w, (size_t)n * 8);
    ENSURE(c->x_id, (size_t)n * 4);
    ENSURE(c->x_cnt, (size_t)n * 4);
    ENSURE(c->seg_k0, (size_t)n * 8);
    ENSURE(c->seg_v0, (size_t)n * 4);
    ENSURE(c->seg_ks, (size_t)n * 8);
    ENSURE(c->seg_vs, (size_t)n * 4);
    ENSURE(c->seg_ws, (size_t)n * 8);
    hipLaunchKernelGGL(k_split_items, dim3(blocks_for(n)), dim3(256), 0, st, (const ulonglong2 *)d_items, n,
                       c->x_w.as<u64>(), c->x_id.as<u32>(), c->x_cnt.as<u32>());
    const u32 kb = plan.key_bits ? plan.key_bits : 1;
    bool stretch = false;
    TRY((group_words_by_stretch<FieldsSrc, u64>(c, plan, combo, c->x_w.as<u64>(), n, c->seg_ws.as<u64>(), c->seg_vs.as<u32>(), &stretch)));
    if (!stretch) TRY(sort_words_by_stretch(c, plan, combo, c->x_w.as<u64>(), n, c->seg_ws.as<u64>(), c->seg_vs.as<u32>(), &stretch));
    if (!stretch) {
      if (kb <= 32) {
        hipLaunchKernelGGL((k_combo_keys<u32, u64>), dim3(blocks_for(n)), dim3(256), 0, st, c->x_w.as<u64>(), n,
                           plan_fields(plan, combo), c->seg_k0.as<u32>(), c->seg_v0.as<u32>());
        TRY(sort_pairs<u32, u32>(c, c->seg_k0.as<u32>(), c->seg_ks.as<u32>(), c->seg_v0.as<u32>(), c->seg_vs.as<u32>(), n, 0, kb));
      } else {
        hipLaunchKernelGGL((k_combo_keys<u64, u64>), dim3(blocks_for(n)), dim3(256), 0, st, c->x_w.as<u64>(), n,
                           plan_fields(plan, combo), c->seg_k0.as<u64>(), c->seg_v0.as<u32>());
        TRY(sort_pairs<u64, u32>(c, c->seg_k0.as<u64>(), c->seg_ks.as<u64>(), c->seg_v0.as<u32>(), c->seg_vs.as<u32>(), n, 0, kb));
      }
      hipLaunchKernelGGL(k_gather_bucket_words<u64>, dim3(blocks_for(n)), dim3(256), 0, st, c->x_w.as<u64>(),
                         c->seg_vs.as<u32>(), n, c->seg_ws.as<u64>());
    }
    // pairs of positions in the received array (V = bucket order -> received position)
    TRY(emit_pairs(c, c->seg_ws.as<u64>(), c->seg_vs.as<u32>(), n, plan, combo, distance, &E));
    id_of = c->x_id.as<u32>();
    cnt_of = c->x_cnt.as<u32>();
  }
  *n_edges = E;                  // known since emit_pairs' count phase; the rest is queued on the stream
  if (E == 0) return HUMID_OK;
  ENSURE(c->x_rec, (size_t)E * 16);
  hipLaunchKernelGGL(k_edge_records, dim3(blocks_for(E)), dim3(256), 0, st, c->share_edges.as<u64>(), (u32)E, id_of,
                     (u32)id_base, cnt_of, c->x_rec.as<ulonglong2>());
  HIPCHK(hipGetLastError());
  *d_records = c->x_rec.as<u64>();
  return HUMID_OK;
}

// distinct endpoints of an edge list, ascending, and the edges relabelled to positions in that list.
// record_stride 1: d_edges[k] = (a << 32 | b).  record_stride 2: the 16-byte records of
// humid_stage_pairs_keyed; *d_node_counts then holds the count of every node.
static int compact_nodes_impl(humid_ctx *c, const uint64_t *d_edges, uint64_t n_edges, uint32_t record_stride, u64 id_bound,
                              const uint32_t **d_nodes, uint64_t *n_nodes, const uint64_t **d_compact_edges,
                              const uint32_t **d_node_counts);
int humid_stage_compact_nodes(humid_ctx *c, const uint64_t *d_edges, uint64_t n_edges, uint32_t record_stride,
                              const uint32_t **d_nodes, uint64_t *n_nodes, const uint64_t **d_compact_edges,
                              const uint32_t **d_node_counts) {
  return compact_nodes_impl(c, d_edges, n_edges, record_stride, 0, d_nodes, n_nodes, d_compact_edges, d_node_counts);
}
// id_bound > 0: every endpoint is below it (the caller knows the number of unique words): the distinct
// endpoints are found with a mark array and one scan over the ids -- no sort (four radix passes over
// the 2E endpoints and their per-pass memsets were a tenth of the multi-GPU pass)
static int compact_nodes_impl(humid_ctx *c, const uint64_t *d_edges, uint64_t n_edges, uint32_t record_stride, u64 id_bound,
                              const uint32_t **d_nodes, uint64_t *n_nodes, const uint64_t **d_compact_edges,
                              const uint32_t **d_node_counts) {
  if (!c) return fail(nullptr, HUMID_E_INVALID, "ctx is null");
  if (!d_nodes || !n_nodes || !d_compact_edges) return fail(c, HUMID_E_INVALID, "bad argument");
  if (record_stride != 1 && record_stride != 2) return fail(c, HUMID_E_INVALID, "record_stride must be 1 or 2");
  if (n_edges >= 0x7fffffffull) return fail(c, HUMID_E_OVERFLOW, "2*edges exceeds 32 bits");
  HIPCHK(hipSetDevice(c->device));
  hipStream_t st = c->stream;
  *d_nodes = nullptr;
  *d_compact_edges = nullptr;
  if (d_node_counts) *d_node_counts = nullptr;
  *n_nodes = 0;
  const u32 E = (u32)n_edges;
  if (E == 0) return HUMID_OK;
  if (!d_edges) return fail(c, HUMID_E_INVALID, "null buffer");
  if (id_bound > 0 && id_bound <= (1ull << 28)) {
    const u32 B = (u32)id_bound;
    ENSURE(c->x_head, (size_t)B + 16);                       // mark bytes
    ENSURE(c->x_ends, (size_t)B * 4);                        // count by id
    ENSURE(c->x_hpos, ((size_t)B + 1) * 4);                  // position by id
    ENSURE(c->x_cedges, (size_t)E * 8);
    HIPCHK(hipMemsetAsync(c->x_head.p, 0, (size_t)B + 1, st));
    HIPCHK(hipMemsetAsync(&c->d_ctr[CTR_OVERFULL], 0, sizeof(ull), st));
    hipLaunchKernelGGL(k_mark_ends, dim3(blocks_for(E)), dim3(256), 0, st, d_edges, E, record_stride, B, c->x_head.as<u8>(),
                       c->x_ends.as<u32>(), c->d_ctr);
    TRY(exscan_in<u32>(c, CastIn<u32, u8>{c->x_head.as<u8>()}, c->x_hpos.as<u32>(), (u64)B + 1));
    HIPCHK(hipGetLastError());
    TRY(read_counters(c, c->x_hpos.as<u32>() + B));
    if (c->h_ctr[CTR_OVERFULL]) return fail(c, HUMID_E_INVALID, "pair record with an index beyond the unique words");
    const u32 M = (u32)(c->h_ctr[CTR_N - 1] & 0xffffffffull);
    ENSURE(c->x_nodes, ((size_t)M + 1) * 4);
    ENSURE(c->x_ncnt, ((size_t)M + 1) * 4);
    hipLaunchKernelGGL(k_marked_nodes, dim3(blocks_for(B)), dim3(256), 0, st, c->x_head.as<u8>(), c->x_hpos.as<u32>(),
                       c->x_ends.as<u32>(), B, record_stride == 2, c->x_nodes.as<u32>(), c->x_ncnt.as<u32>());
    hipLaunchKernelGGL(k_relabel_pairs, dim3(blocks_for(E)), dim3(256), 0, st, d_edges, E, record_stride, B, c->x_hpos.as<u32>(),
                       c->x_cedges.as<u64>());
    HIPCHK(hipGetLastError());
    *d_nodes = c->x_nodes.as<u32>();
    *n_nodes = M;
    *d_compact_edges = c->x_cedges.as<u64>();
    if (d_node_counts && record_stride == 2) *d_node_counts = c->x_ncnt.as<u32>();
    return HUMID_OK;
  }
  const u32 n2 = 2 * E;
  ENSURE(c->x_ends, (size_t)n2 * 4);
  ENSURE(c->x_ends_s, (size_t)n2 * 4);
  ENSURE(c->x_slot, (size_t)n2 * 4);
  ENSURE(c->x_slot_s, (size_t)n2 * 4);
  ENSURE(c->x_head, ((size_t)n2 + 1) * 4);
  ENSURE(c->x_hpos, ((size_t)n2 + 1) * 4);
  hipLaunchKernelGGL(k_edge_ends, dim3(blocks_for(E)), dim3(256), 0, st, d_edges, E, record_stride, c->x_ends.as<u32>(),
                     c->x_slot.as<u32>());
  TRY(sort_pairs<u32, u32>(c, c->x_ends.as<u32>(), c->x_ends_s.as<u32>(), c->x_slot.as<u32>(), c->x_slot_s.as<u32>(), n2, 0, 32));   // (global indices: all 32 bits may be in use)
  hipLaunchKernelGGL(k_heads_u32, dim3(blocks_for((u64)n2 + 1)), dim3(256), 0, st, c->x_ends_s.as<u32>(), n2,
                     c->x_head.as<u32>());
  TRY(exscan_u32(c, c->x_head.as<u32>(), c->x_hpos.as<u32>(), (u64)n2 + 1));
  HIPCHK(hipGetLastError());
  TRY(read_counters(c, c->x_hpos.as<u32>() + n2));
  const u32 M = (u32)(c->h_ctr[CTR_N - 1] & 0xffffffffull);
  ENSURE(c->x_nodes, ((size_t)M + 1) * 4);
  ENSURE(c->x_ncnt, ((size_t)M + 1) * 4);
  ENSURE(c->x_cedges, (size_t)E * 8);
  hipLaunchKernelGGL(k_compact_heads_u32, dim3(blocks_for(n2)), dim3(256), 0, st, c->x_ends_s.as<u32>(),
                     c->x_head.as<u32>(), c->x_hpos.as<u32>(), n2, c->x_nodes.as<u32>());
  // x_ends is free again: the positions of both ends of every edge, by slot
  hipLaunchKernelGGL(k_relabel_ends, dim3(blocks_for(n2)), dim3(256), 0, st, c->x_slot_s.as<u32>(), c->x_head.as<u32>(),
                     c->x_hpos.as<u32>(), n2, d_edges, record_stride, c->x_ends.as<u32>(), c->x_ncnt.as<u32>());
  hipLaunchKernelGGL(k_pack_cedges, dim3(blocks_for(E)), dim3(256), 0, st, c->x_ends.as<u32>(), E, c->x_cedges.as<u64>());
  HIPCHK(hipGetLastError());
  *d_nodes = c->x_nodes.as<u32>();     // M is known; the node list and the relabelling are queued
  *n_nodes = M;
  *d_compact_edges = c->x_cedges.as<u64>();
  if (d_node_counts && record_stride == 2) *d_node_counts = c->x_ncnt.as<u32>();
  return HUMID_OK;
}

// ---- multi-GPU, edit distance: this rank's share of the Levenshtein neighbour search over the
// whole (replicated) unique array; the shares may overlap in pairs (a pair can be found by several
// joins): gather them and pass them through humid_stage_unique_edges before humid_stage_graph_edges.
int humid_stage_pairs_edit(humid_ctx *c, const uint64_t *d_g_word, uint64_t n_unique, uint32_t word_nt,
                           uint32_t distance, uint32_t part_rank, uint32_t part_world, const uint64_t **d_edges,
                           uint64_t *n_edges) {
  if (!c) return fail(nullptr, HUMID_E_INVALID, "ctx is null");
  if (!d_edges || !n_edges || part_world == 0 || part_rank >= part_world) return fail(c, HUMID_E_INVALID, "bad argument");
  TRY(check_run_args(c, n_unique, word_nt, 0));
  if (distance > 5) return fail(c, HUMID_E_UNSUPPORTED, "edit distance %u > 5 is not supported", distance);
  HIPCHK(hipSetDevice(c->device));
  *d_edges = nullptr;
  *n_edges = 0;
  if (n_unique && !d_g_word) return fail(c, HUMID_E_INVALID, "null buffer");
  u64 E = 0;
  if (n_unique > 1 && distance > 0)
    TRY(edit_edges<u64>(c, d_g_word, (u32)n_unique, word_nt, distance, &E, part_rank, part_world, false));
  HIPCHK(hipStreamSynchronize(c->stream));
  *n_edges = E;
  *d_edges = E ? c->e_raw.as<u64>() : nullptr;
  return HUMID_OK;
}

int humid_stage_unique_edges(humid_ctx *c, const uint64_t *d_edges, uint64_t n_edges, uint64_t n_unique,
                             const uint64_t **d_unique_edges, uint64_t *n_unique_edges) {
  if (!c) return fail(nullptr, HUMID_E_INVALID, "ctx is null");
  if (!d_unique_edges || !n_unique_edges || n_unique > 0xffffffffull) return fail(c, HUMID_E_INVALID, "bad argument");
  HIPCHK(hipSetDevice(c->device));
  *d_unique_edges = nullptr;
  *n_unique_edges = 0;
  if (n_edges && !d_edges) return fail(c, HUMID_E_INVALID, "null buffer");
  u64 E = 0;
  TRY(unique_edges(c, d_edges, n_edges, (u32)n_unique, &E));
  HIPCHK(hipStreamSynchronize(c->stream));
  *n_unique_edges = E;
  *d_unique_edges = E ? c->e_edges.as<u64>() : nullptr;
  return HUMID_OK;
}

// HIP-event times of the two N-proportional kernels of the last count_dense / map_dense pair on
// this context (bench.py's roofline leg in multi-GPU runs); waits for the stream.
int humid_stage_kernel_ms(humid_ctx *c, float *ms_k_insert, float *ms_k_map, uint32_t *count_mode_used) {
  if (!c) return fail(nullptr, HUMID_E_INVALID, "ctx is null");
  if (!c->dense_mode || (c->N && !c->stage_map_timed))
    return fail(c, HUMID_E_STATE, "no completed humid_stage_count_dense + humid_stage_map_dense");
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipStreamSynchronize(c->stream));
  float a = 0, b = 0;
  if (c->N) {
    HIPCHK(hipEventElapsedTime(&a, c->kev[0], c->kev[1]));
    if (c->kev_on) HIPCHK(hipEventElapsedTime(&b, c->kev[37], c->kev[38]));
  }
  if (ms_k_insert) *ms_k_insert = a;
  if (ms_k_map) *ms_k_map = b;
  if (count_mode_used) *count_mode_used = (c->last_count_sorted ? 3u : c->last_count_lds ? (c->last_count_ordered ? 2u : 0u) : 1u) | (c->last_rec8 ? 0x100u : 0u);
  return HUMID_OK;
}

// words of this rank's usable reads in the owner-major order humid_stage_owner_perm just computed
int humid_stage_route_words(humid_ctx *c, const uint64_t *d_words, uint64_t n_send, const uint64_t **d_routed) {
  if (!c) return fail(nullptr, HUMID_E_INVALID, "ctx is null");
  if (!d_routed) return fail(c, HUMID_E_INVALID, "bad argument");
  *d_routed = nullptr;
  if (n_send == 0) return HUMID_OK;
  if (!d_words) return fail(c, HUMID_E_INVALID, "null buffer");
  if (n_send * 4 > c->perm.cap) return fail(c, HUMID_E_STATE, "no preceding humid_stage_owner_perm of at least n_send reads");
  HIPCHK(hipSetDevice(c->device));
  ENSURE(c->x_route, (size_t)n_send * 8);
  hipLaunchKernelGGL(k_route_words, dim3(grid_stride_blocks(n_send)), dim3(256), 0, c->stream, d_words,
                     c->perm.as<u32>(), (u32)n_send, c->x_route.as<u64>());
  HIPCHK(hipGetLastError());
  *d_routed = c->x_route.as<u64>();
  return HUMID_OK;
}

// Stable routing of this rank's usable reads to the owners of their value ranges, without a host
// wait: send_counts[q] (reads of owner q; the caller knows them from the all-gathered histograms)
// fix the block bases.  *d_routed: the words, owner-major, input order inside every block;
// *d_perm: routed position -> read index (humid_stage_scatter takes it).  A count that does not match
// the data raises the flag humid_stage_route_check reports.
int humid_stage_route(humid_ctx *c, const uint64_t *d_words, const uint8_t *d_filtered, uint64_t n_reads,
                      const uint64_t *range_lo, const uint64_t *range_hi, uint32_t n_ranks,
                      const uint64_t *send_counts, const uint64_t **d_routed, const uint32_t **d_perm) {
  if (!c) return fail(nullptr, HUMID_E_INVALID, "ctx is null");
  if (!range_lo || !range_hi || !send_counts || !d_routed || !d_perm || n_ranks == 0) return fail(c, HUMID_E_INVALID, "bad argument");
  if (n_ranks > MAX_RANKS) return fail(c, HUMID_E_UNSUPPORTED, "more than %d ranks", MAX_RANKS);
  if (n_reads > 0x7fffffffull) return fail(c, HUMID_E_OVERFLOW, "n_reads exceeds 2^31-1");
  HIPCHK(hipSetDevice(c->device));
  hipStream_t st = c->stream;
  *d_routed = nullptr;
  *d_perm = nullptr;
  const u32 n = (u32)n_reads;
  if (n == 0) return HUMID_OK;
  if (!d_words || !d_filtered) return fail(c, HUMID_E_INVALID, "null buffer");
  OwnerRanges rg;
  OwnerBases ob;
  u64 tot = 0;
  for (u32 q = 0; q < MAX_RANKS; q++) {
    rg.lo[q] = q < n_ranks ? range_lo[q] : 1;
    rg.hi[q] = q < n_ranks ? range_hi[q] : 0;
    ob.b[q] = (u32)tot;
    if (q < n_ranks) tot += send_counts[q];
  }
  ob.b[MAX_RANKS] = (u32)tot;
  if (tot > n) return fail(c, HUMID_E_INVALID, "send_counts exceed n_reads");
  const u32 n_tiles = (n + ROUTE_TILE - 1) / ROUTE_TILE;
  ENSURE(c->route_tiles, ((size_t)n_tiles * MAX_RANKS + 16) * 4 + ROUTE_BINS);   // tile counts | bad flag | owner table
  ENSURE(c->perm, (size_t)n * 4);
  ENSURE(c->xo_inv, (size_t)n * 4);
  ENSURE(c->x_route, (size_t)(tot ? tot : 1) * 8);
  u32 *tile_cnt = c->route_tiles.as<u32>(), *bad = tile_cnt + (size_t)n_tiles * MAX_RANKS;
  // ranges cut at the bins of a prefix histogram (every boundary a multiple of 2^shift, at most
  // ROUTE_BINS bins below the last boundary): owners come from a table in LDS
  u64 bits_or = 0, top_lo = 0;
  for (u32 q = 0; q < n_ranks; q++)
    if (range_lo[q] <= range_hi[q]) {
      bits_or |= range_lo[q] | (range_hi[q] + 1);
      top_lo = std::max<u64>(top_lo, range_lo[q]);
    }
  const u32 shift = bits_or ? (u32)__builtin_ctzll(bits_or) : 63u;
  const bool table = (top_lo >> shift) < ROUTE_BINS;
  u8 *d_table = (u8 *)(bad + 4);
  if (table) {
    hipLaunchKernelGGL(k_route_table, dim3(1), dim3(1024), 0, st, rg, n_ranks, shift, d_table);
    hipLaunchKernelGGL(k_route_tile_hist<true>, dim3(n_tiles), dim3(1024), 0, st, d_words, d_filtered, n, rg, n_ranks, shift,
                       (const u8 *)d_table, tile_cnt, bad);
  } else
    hipLaunchKernelGGL(k_route_tile_hist<false>, dim3(n_tiles), dim3(1024), 0, st, d_words, d_filtered, n, rg, n_ranks, shift,
                       (const u8 *)d_table, tile_cnt, bad);
  hipLaunchKernelGGL(k_route_scan, dim3(1), dim3(1024), 0, st, tile_cnt, n_tiles, ob, bad);
  if (table)
    hipLaunchKernelGGL(k_route_scatter<true>, dim3(n_tiles), dim3(1024), 0, st, d_words, d_filtered, n, rg, n_ranks, shift,
                       (const u8 *)d_table, (const u32 *)tile_cnt, c->x_route.as<u64>(), c->perm.as<u32>(), c->xo_inv.as<u32>());
  else
    hipLaunchKernelGGL(k_route_scatter<false>, dim3(n_tiles), dim3(1024), 0, st, d_words, d_filtered, n, rg, n_ranks, shift,
                       (const u8 *)d_table, (const u32 *)tile_cnt, c->x_route.as<u64>(), c->perm.as<u32>(), c->xo_inv.as<u32>());
  HIPCHK(hipGetLastError());
  c->route_checked = false;
  c->route_bad = bad;
  *d_routed = c->x_route.as<u64>();
  *d_perm = c->perm.as<u32>();
  return HUMID_OK;
}

// waits for the stream; HUMID_E_INVALID if the send_counts of the last humid_stage_route did not match
int humid_stage_route_check(humid_ctx *c) {
  if (!c) return fail(nullptr, HUMID_E_INVALID, "ctx is null");
  if (c->route_checked || !c->route_bad) return HUMID_OK;
  HIPCHK(hipSetDevice(c->device));
  u32 flag = 0;
  HIPCHK(hipMemcpyAsync(&flag, c->route_bad, 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  c->route_checked = true;
  if (flag) return fail(c, HUMID_E_INVALID, "humid_stage_route: send_counts do not match the reads");
  return HUMID_OK;
}

// Cluster id and maxLeaf flag of this rank's u_local unique words (global walk indices
// id_base .. id_base + u_local - 1) from the replicated compact graph: d_nodes[n_nodes] ascending
// global indices of the leaves that have neighbours, d_ccid / d_cismax their results from
// humid_stage_graph_edges (compact ids 1..n_clusters in creator order).  A leaf outside the compact
// graph is a singleton: its own cluster, its own maxLeaf.  Ids follow src/humid.cc:177-180: 1 + the
// number of cluster-creating leaves before it in the walk.  No host synchronisation.
int humid_stage_exchange_ids(humid_ctx *c, const uint32_t *d_nodes, const uint32_t *d_ccid,
                             const uint8_t *d_cismax, uint64_t n_nodes, uint64_t n_clusters, uint64_t id_base,
                             uint64_t u_local, const uint32_t **d_l_cid, const uint8_t **d_l_ismax) {
  if (!c) return fail(nullptr, HUMID_E_INVALID, "ctx is null");
  if (!d_l_cid || !d_l_ismax) return fail(c, HUMID_E_INVALID, "bad argument");
  if (id_base + u_local > 0xffffffffull || n_nodes > 0xfffffffeull || n_clusters > n_nodes)
    return fail(c, HUMID_E_OVERFLOW, "index out of range");
  *d_l_cid = nullptr;
  *d_l_ismax = nullptr;
  if (u_local == 0) return HUMID_OK;
  if (n_nodes && (!d_nodes || !d_ccid || !d_cismax)) return fail(c, HUMID_E_INVALID, "null buffer");
  HIPCHK(hipSetDevice(c->device));
  hipStream_t st = c->stream;
  const u32 M = (u32)n_nodes, Cc = (u32)n_clusters, goff = (u32)id_base, U = (u32)u_local;
  ENSURE(c->x_creator, ((size_t)Cc + 1) * 4);
  ENSURE(c->x_base, ((size_t)Cc + 1) * 4);
  // x_mark | x_markcr | the two counters of k_xid_first: one allocation, one memset
  ENSURE(c->x_mark, ((size_t)U * 2 + 4) * 4);
  ENSURE(c->x_scan, ((size_t)U + 1) * 8);
  ENSURE(c->x_lcid, (size_t)U * 4);
  ENSURE(c->x_lismax, (size_t)U);
  u32 *x_mark = c->x_mark.as<u32>(), *x_markcr = x_mark + U, *x_first = x_markcr + U;
  HIPCHK(hipMemsetAsync(x_mark, 0, ((size_t)U * 2 + 2) * 4, st));
  if (M) {
    HIPCHK(hipMemsetAsync(c->x_creator.p, 0xff, ((size_t)Cc + 1) * 4, st));
    hipLaunchKernelGGL(k_xid_creators, dim3(blocks_for(M)), dim3(256), 0, st, d_ccid, M, Cc, c->x_creator.as<u32>());
    if (Cc)
      hipLaunchKernelGGL(k_xid_base, dim3(blocks_for(Cc)), dim3(256), 0, st, d_nodes, c->x_creator.as<u32>(), M, Cc,
                         goff, U, c->x_base.as<u32>(), x_markcr);
    hipLaunchKernelGGL(k_xid_mark, dim3(blocks_for(M)), dim3(256), 0, st, d_nodes, M, goff, U, x_mark);
    hipLaunchKernelGGL(k_xid_first, dim3(1), dim3(64), 0, st, d_nodes, c->x_creator.as<u32>(), M, Cc, goff,
                       x_first);
  }
  {
    ComposeIn<XidFlagOp, IotaIn> fin{XidFlagOp{x_mark, x_markcr, U}, IotaIn{}};
    TRY(exscan_in<u64>(c, fin, c->x_scan.as<u64>(), (u64)U));
  }
  hipLaunchKernelGGL(k_xid_assign, dim3(blocks_for(U)), dim3(256), 0, st, x_mark, c->x_scan.as<u64>(),
                     x_first, d_ccid, d_cismax, c->x_base.as<u32>(), Cc, goff, U, c->x_lcid.as<u32>(),
                     c->x_lismax.as<u8>());
  HIPCHK(hipGetLastError());
  *d_l_cid = c->x_lcid.as<u32>();
  *d_l_ismax = c->x_lismax.as<u8>();
  return HUMID_OK;
}

// ---- multi-GPU result return ------------------------------------------------------------------
int humid_stage_owned_results(humid_ctx *c, const uint32_t *d_local_cluster_id, const uint8_t *d_local_is_max,
                              const uint64_t *shard_begin, uint32_t n_shards, const uint32_t **d_packed,
                              uint64_t *counts) {
  if (!c) return fail(nullptr, HUMID_E_INVALID, "ctx is null");
  if (c->last_count_lds) return fail(c, HUMID_E_STATE, "owned results need the global-table count variant (count_mode 1)");
  if (!shard_begin || !counts || !d_packed || n_shards == 0 || n_shards > 4096) return fail(c, HUMID_E_INVALID, "bad argument");
  HIPCHK(hipSetDevice(c->device));
  hipStream_t st = c->stream;
  const u32 N = (u32)c->N, U = (u32)c->U;
  *d_packed = nullptr;
  for (u32 q = 0; q < n_shards; q++) counts[q] = 0;
  if (N == 0) return HUMID_OK;
  for (u32 q = 0; q <= n_shards; q++)
    if (shard_begin[q] > N || (q && shard_begin[q] < shard_begin[q - 1])) return fail(c, HUMID_E_INVALID, "shard_begin must ascend within [0, n_reads]");
  if (U && (!d_local_cluster_id || !d_local_is_max)) return fail(c, HUMID_E_INVALID, "null buffer");
  if (U > 0)
    hipLaunchKernelGGL(k_slot_results, dim3(blocks_for(U)), dim3(256), 0, st, d_local_cluster_id, d_local_is_max,
                       c->s_first.as<u32>(), c->s_slot.as<u32>(), U, c->slot_out.as<u64>());
  ENSURE(c->opos, ((size_t)N + 1) * 4);
  {
    ComposeIn<OwnedFlagOp, IotaIn> fin{OwnedFlagOp{c->slot_of_read.as<u32>(), N}, IotaIn{}};
    TRY(exscan_in<u32>(c, fin, c->opos.as<u32>(), (u64)N + 1));
  }
  // per-shard counts: opos at the shard boundaries (a handful of 4-byte copies, one sync)
  std::vector<u32> got(n_shards + 1);
  for (u32 q = 0; q <= n_shards; q++)
    HIPCHK(hipMemcpyAsync(&got[q], c->opos.as<u32>() + shard_begin[q], 4, hipMemcpyDeviceToHost, st));
  u32 total = 0;
  HIPCHK(hipMemcpyAsync(&total, c->opos.as<u32>() + N, 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  // reads outside [shard_begin[0], shard_begin[n_shards]) must not be owned
  if (got[0] != 0 || got[n_shards] != total) return fail(c, HUMID_E_INVALID, "owned reads outside the shard table");
  for (u32 q = 0; q < n_shards; q++) counts[q] = got[q + 1] - got[q];
  ENSURE(c->own_packed, ((size_t)total + 1) * 4);
  if (total)
    hipLaunchKernelGGL(k_owned_results, dim3(grid_stride_blocks(N)), dim3(256), 0, st, c->slot_of_read.as<u32>(),
                       c->opos.as<u32>(), c->slot_out.as<u64>(), N, c->own_packed.as<u32>());
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(st));
  *d_packed = c->own_packed.as<u32>();
  return HUMID_OK;
}

int humid_stage_owner_perm(humid_ctx *c, const uint64_t *d_words, const uint8_t *d_filtered, uint64_t n_reads,
                           const uint64_t *range_lo, const uint64_t *range_hi, uint32_t n_ranks,
                           const uint32_t **d_perm, uint64_t *counts) {
  return humid_stage_owner_perm_wide(c, d_words, d_filtered, n_reads, 32, range_lo, range_hi, n_ranks, d_perm, counts);
}

int humid_stage_owner_perm_wide(humid_ctx *c, const uint64_t *d_words, const uint8_t *d_filtered, uint64_t n_reads,
                                uint32_t word_nt, const uint64_t *range_lo, const uint64_t *range_hi, uint32_t n_ranks,
                                const uint32_t **d_perm, uint64_t *counts) {
  if (!c) return fail(nullptr, HUMID_E_INVALID, "ctx is null");
  if (word_nt == 0 || word_nt > 64) return fail(c, HUMID_E_UNSUPPORTED, "word_nt must be 1 .. 64");
  if (!range_lo || !range_hi || !counts || !d_perm || n_ranks == 0) return fail(c, HUMID_E_INVALID, "bad argument");
  if (n_ranks > MAX_RANKS) return fail(c, HUMID_E_UNSUPPORTED, "more than %d ranks", MAX_RANKS);
  if (n_reads > 0x7fffffffull) return fail(c, HUMID_E_OVERFLOW, "n_reads exceeds 2^31-1");
  HIPCHK(hipSetDevice(c->device));
  hipStream_t st = c->stream;
  const u32 n = (u32)n_reads;
  *d_perm = nullptr;
  for (u32 q = 0; q < n_ranks; q++) counts[q] = 0;
  if (n == 0) return HUMID_OK;
  if (!d_words || !d_filtered) return fail(c, HUMID_E_INVALID, "null buffer");
  OwnerRanges rg;
  for (u32 q = 0; q < MAX_RANKS; q++) { rg.lo[q] = q < n_ranks ? range_lo[q] : 1; rg.hi[q] = q < n_ranks ? range_hi[q] : 0; }
  ENSURE(c->owner, (size_t)n);
  ENSURE(c->owner_sorted, (size_t)n);
  ENSURE(c->perm, (size_t)n * 4);
  const u64 *range_keys = d_words;                      // (two-word words: their heads)
  if (word_nt > 32) TRY(stage_heads(c, d_words, n, word_nt, &range_keys));
  hipLaunchKernelGGL(k_owner_of, dim3(blocks_for(n)), dim3(256), 0, st, range_keys, d_filtered, n, rg, n_ranks,
                     c->owner.as<u8>());
  {
    TRY((sort_pairs_in<u8, u32>(c, PtrIn<u8>{c->owner.as<u8>()}, c->owner_sorted.as<u8>(), IotaIn{}, c->perm.as<u32>(), n, 0, 8)));
  }
  ENSURE(c->small, (size_t)(n_ranks + 2) * 4);
  hipLaunchKernelGGL(k_owner_bounds, dim3(1), dim3(64), 0, st, c->owner_sorted.as<u8>(), n, n_ranks,
                     c->small.as<u32>());
  std::vector<u32> b(n_ranks + 2);
  HIPCHK(hipMemcpyAsync(b.data(), c->small.p, (n_ranks + 2) * 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(st));
  for (u32 q = 0; q < n_ranks; q++) counts[q] = b[q + 1] - b[q];
  *d_perm = c->perm.as<u32>();
  return HUMID_OK;
}

int humid_stage_scatter(humid_ctx *c, const uint32_t *d_perm, const uint32_t *d_packed, uint64_t n_recv,
                        uint64_t n_reads, uint32_t *d_cluster_id, uint8_t *d_keep) {
  if (!c) return fail(nullptr, HUMID_E_INVALID, "ctx is null");
  if (n_recv > n_reads) return fail(c, HUMID_E_INVALID, "n_recv > n_reads");
  HIPCHK(hipSetDevice(c->device));
  hipStream_t st = c->stream;
  if (n_reads) {
    if (!d_cluster_id || !d_keep) return fail(c, HUMID_E_INVALID, "null buffer");
    HIPCHK(hipMemsetAsync(d_cluster_id, 0, (size_t)n_reads * 4, st));
    HIPCHK(hipMemsetAsync(d_keep, 0, (size_t)n_reads, st));
  }
  if (n_recv) {
    if (!d_perm || !d_packed) return fail(c, HUMID_E_INVALID, "null buffer");
    hipLaunchKernelGGL(k_scatter_results, dim3(grid_stride_blocks(n_recv)), dim3(256), 0, st, d_perm, d_packed,
                       (u32)n_recv, d_cluster_id, d_keep);
  }
  HIPCHK(hipGetLastError());
  return HUMID_OK;              // queued on the context's stream
}

}  // extern "C"
