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
//
// Translation units (round 3): this file = the context, the single-GPU entry points and the accessors of
// include/humid_hip.h; humid_exchange.hip = the exchange pass and the multi-GPU stage entry points; shm.cpp = the
// shared-memory gather (no HIP); pipeline.hip.h = the pipeline itself, with internal linkage, compiled into both.
#include "pipeline.hip.h"

static std::string g_err;
// (the error text of calls without a context: also set from humid_exchange.hip and shm.cpp)
extern "C" __attribute__((visibility("hidden"))) void humid_set_global_error(const char *text) { g_err = text ? text : ""; }

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

}  // extern "C"
