// pipeline.hip.h -- the pipeline of libhumid_hip.so behind its C ABI: context, stage A (exact counts + walk order),
// stage B (neighbours + clusters), stage C (per-read outputs), the helpers around them.  Everything here has INTERNAL
// linkage (static functions, templates, static kernels in the headers below), so the two HIP translation units of the
// library -- humid_hip.hip (context, single-GPU entry points, accessors) and humid_exchange.hip (the exchange pass and
// the multi-GPU stage entry points) -- each compile what they use of it; round 3 split them (round 2: one file).
#ifndef HUMID_PIPELINE_HIP_H
#define HUMID_PIPELINE_HIP_H

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

// the error text of calls without a context lives in ONE place (humid_hip.hip): humid_last_error(NULL) reads it
extern "C" void humid_set_global_error(const char *text);

static int fail(humid_ctx *c, int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (c) c->err = buf; else humid_set_global_error(buf);
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
static __global__ void k_publish_counters(const ull *__restrict__ ctr, const u32 *__restrict__ extra32, const u32 *__restrict__ extra32b,
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
  RecKey rk;
  rk.lo = km.lo; rk.scale = km.scale;
  rk.pow2 = km.shift < 64 ? 1u : 0u;
  rk.z = rk.pow2 ? km.shift : 63u - (u32)__builtin_clzll(km.scale);
  rk.kbits = 64 - rk.z;
  const u32 ibits = bits_for(N);
  // a record = the key bits below the coarse bin + the read index: kbits - d1 + ibits <= 64.  24-nt words over their
  // whole range have 48 key bits: with the balanced split d1 = pb / 2 <= 9 that ends at 2^25 = 33 M reads; one more
  // bit in the FIRST level (1024 coarse bins: half as long runs per bin and tile there) carries the record path to the
  // 67 M reads of the un-permute's bin table (BASELINE configs 3 and 5: 50 M reads)
  u32 d1 = (pb + 1) / 2;
  while (d1 < 10 && pb - d1 > 1 && rk.kbits - d1 + ibits > 64) d1++;
  const u32 d2 = pb - d1, nb1 = 1u << d1, n_parts = 1u << pb;
  if (rk.kbits < pb + 1 || rk.kbits - d1 + ibits > 64 || d2 > 9 || d2 == 0) return HUMID_OK;
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
  // p8_cur, in u32: [cursor1 1024 | cursor2 n_parts] zeroed, then [cbase 1025 | tprefix 1025].  (Not pt_work: the
  // reads per bucket, cursor2, are read again by the un-permute at the end of the pass, and the graph
  // stage's grouping uses pt_work in between.)
  ENSURE(c->p8_cur, ((size_t)PT_MAXBINS1 + n_parts + 2 * (PT_MAXBINS1 + 1)) * 4);
  u32 *cursor1 = c->p8_cur.as<u32>(), *cursor2 = cursor1 + PT_MAXBINS1, *cbase = cursor2 + n_parts, *tprefix = cbase + PT_MAXBINS1 + 1;
  HIPCHK(hipEventRecord(c->ev[0], st));
  {
    ZeroList z;
    memset(&z, 0, sizeof z);
    z.p[0] = cursor1; z.n[0] = PT_MAXBINS1 + n_parts;
    z.p[1] = (u32 *)c->d_ctr; z.n[1] = 2 * CTR_N;
    z.p[2] = (u32 *)(agg + n_parts); z.n[2] = 2;
    hipLaunchKernelGGL(k_zero_many, dim3(32), dim3(256), 0, st, z);
  }
  const bool check_range = !(range_lo == 0 && range_hi == ~0ull);
  const Reads8 src{d_words, d_filt, range_lo, range_hi, check_range ? 1u : 0u, rk};
  const u32 tiles1 = (N + PT_TILE - 1) / PT_TILE, tiles2 = tiles1 + nb1;
  static const bool s1_small = getenv("HUMID_S1_THREADS") ? atoi(getenv("HUMID_S1_THREADS")) == 512 : false;  // (experiments: 512 is 20 us slower)
  if (s1_small && nb1 <= 512)
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
// the fields of combination cb as a kernel argument
static ComboFields plan_fields(const ComboPlan &plan, u32 cb) {
  ComboFields cf;
  cf.nf = plan.nfield[cb];
  for (u32 f = 0; f < MAX_FIELDS; f++) { cf.shift[f] = plan.shift[cb][f]; cf.width[f] = plan.width[cb][f]; }
  return cf;
}
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
    const u32 r_first = src.er.e ? 0u : ER_REGIONS;      // no regions (one dense pair list): the `far` region alone
    hipLaunchKernelGGL(k_fill_and_stats, dim3(gx * (ER_REGIONS + 1 - r_first) + blocks_for(Mb)), dim3(256), 0, st, src.er, (const u32 *)g.off,
                       c->cg_curs.as<u32>(), g.idx, c->small.as<u32>(), gx, (const u32 *)g.deg, g.parent, Mb, g.csize, m_dev, r_first);
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
                   else if (D == 2) EDIT_JOIN(false, u32, 2, c->pc.as<u32>(), (const u32 *)nullptr, (u64 *)nullptr);
                   else EDIT_JOIN(false, u32, 0, c->pc.as<u32>(), (const u32 *)nullptr, (u64 *)nullptr); }
        else { if (D <= 1) EDIT_JOIN(false, u64, 1, c->pc.as<u32>(), (const u32 *)nullptr, (u64 *)nullptr);
               else if (D == 2) EDIT_JOIN(false, u64, 2, c->pc.as<u32>(), (const u32 *)nullptr, (u64 *)nullptr);
               else EDIT_JOIN(false, u64, 0, c->pc.as<u32>(), (const u32 *)nullptr, (u64 *)nullptr); }
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
                     else if (D == 2) EDIT_CHUNKS(false, u32, 2, n_pieces, c->e_pc2.as<u32>(), (const u32 *)nullptr, (u64 *)nullptr);
                     else EDIT_CHUNKS(false, u32, 0, n_pieces, c->e_pc2.as<u32>(), (const u32 *)nullptr, (u64 *)nullptr); }
          else { if (D <= 1) EDIT_CHUNKS(false, u64, 1, n_pieces, c->e_pc2.as<u32>(), (const u32 *)nullptr, (u64 *)nullptr);
                 else if (D == 2) EDIT_CHUNKS(false, u64, 2, n_pieces, c->e_pc2.as<u32>(), (const u32 *)nullptr, (u64 *)nullptr);
                 else EDIT_CHUNKS(false, u64, 0, n_pieces, c->e_pc2.as<u32>(), (const u32 *)nullptr, (u64 *)nullptr); }
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
                     else if (D == 2) EDIT_CHUNKS(true, u32, 2, n_pieces, (u32 *)nullptr, (const u32 *)c->e_poff2.as<u32>(), c->e_raw.as<u64>() + raw);
                     else EDIT_CHUNKS(true, u32, 0, n_pieces, (u32 *)nullptr, (const u32 *)c->e_poff2.as<u32>(), c->e_raw.as<u64>() + raw); }
          else { if (D <= 1) EDIT_CHUNKS(true, u64, 1, n_pieces, (u32 *)nullptr, (const u32 *)c->e_poff2.as<u32>(), c->e_raw.as<u64>() + raw);
                 else if (D == 2) EDIT_CHUNKS(true, u64, 2, n_pieces, (u32 *)nullptr, (const u32 *)c->e_poff2.as<u32>(), c->e_raw.as<u64>() + raw);
                 else EDIT_CHUNKS(true, u64, 0, n_pieces, (u32 *)nullptr, (const u32 *)c->e_poff2.as<u32>(), c->e_raw.as<u64>() + raw); }
        } else
        if (k32) { if (D <= 1) EDIT_JOIN(true, u32, 1, (u32 *)nullptr, c->poff.as<u32>(), c->e_raw.as<u64>() + raw);
                   else if (D == 2) EDIT_JOIN(true, u32, 2, (u32 *)nullptr, c->poff.as<u32>(), c->e_raw.as<u64>() + raw);
                   else EDIT_JOIN(true, u32, 0, (u32 *)nullptr, c->poff.as<u32>(), c->e_raw.as<u64>() + raw); }
        else { if (D <= 1) EDIT_JOIN(true, u64, 1, (u32 *)nullptr, c->poff.as<u32>(), c->e_raw.as<u64>() + raw);
               else if (D == 2) EDIT_JOIN(true, u64, 2, (u32 *)nullptr, c->poff.as<u32>(), c->e_raw.as<u64>() + raw);
               else EDIT_JOIN(true, u64, 0, (u32 *)nullptr, c->poff.as<u32>(), c->e_raw.as<u64>() + raw); }
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
    const u32 chunk = (n_bins > 1024 && n_bins <= 1536) ? 7u * PT_THREADS : PT_TILE;     // records the variant below stages at a time
    const u32 B = (u32)std::min<u64>(64, std::max<u64>(1, (chunk - chunk / 8) / mean));
    const u32 grid = (c->n_parts + B - 1) / B;
    if (n_bins <= 1024)
      hipLaunchKernelGGL(k_unperm_bins8<1024>, dim3(grid), dim3(1024), 0, st, (const u64 *)c->p8_b.as<u64>(), c->rec_cursor2,
                         (const u64 *)c->slot_out.as<u64>(), c->n_parts, B, N, wshift, n_bins, ucur, rec);
    else if (n_bins <= 1536)
      hipLaunchKernelGGL((k_unperm_bins8<1536, 7>), dim3(grid), dim3(1024), 0, st, (const u64 *)c->p8_b.as<u64>(), c->rec_cursor2,
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

#endif  // HUMID_PIPELINE_HIP_H
