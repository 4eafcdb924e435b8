// humid_exchange.hip -- the multi-GPU side of libhumid_hip.so (include/humid_hip.h): humid_dedup_run_exchange, the pass
// of ONE rank of the exchange mode, and the humid_stage_* entry points that humid_amd/sharded.py
// drives stage by stage (sections 4a, 4b).  A translation unit of its own since round 3; the pipeline it calls is
// pipeline.hip.h (internal linkage: compiled here for what this file uses of it).
#include "pipeline.hip.h"

extern "C" {

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
  u32 *dcnt = c->xo_cnt.as<u32>();                                  // [0, P]: records per destination; [32, 32 + P]: scatter cursors (P <= 16: up to index 48); 52: flagged; 56..: totals
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
    hipLaunchKernelGGL((k_pairs_records<P0, WT>), dim3(blocks_for(NN, PA_PPT * 256)), dim3(256), 0, st, (const WT *)(W), (const u32 *)(V), \
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
                           (u32)goff, (const u32 *)c->xo_parent.as<u32>(), (const u8 *)c->xo_flag.as<u8>(), dcnt + 52,
                           c->xo_sel.as<ulonglong2>());
        u32 h = 0;
        HIPCHK(hipMemcpyAsync(&h, dcnt + 52, 4, hipMemcpyDeviceToHost, st));
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
    hipLaunchKernelGGL(k_top_hist, dim3(512), dim3(1024), n_bins * 4, c->stream, keys, d_filtered,
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

// (plan_fields: pipeline.hip.h)

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
