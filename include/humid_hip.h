/*
 * humid_hip.h -- C ABI of libhumid_hip.so: HUMID's neighbour-search-and-cluster hot
 * path on one MI355X (gfx950), hand-written HIP.
 *
 * The reference (jfjlaros/HUMID, /root/reference) has no FFI/plugin seam; the seam
 * this library fills is the C++ surface src/humid.cc uses between FastQ pass 1 and
 * pass 2: lib/trie's Trie<4,NLeaf> (add / walk /
 * asymmetricHamming / find) plus src/cluster.{h,cc} and src/leaf.h.  Each entry
 * point below names the reference interface it replaces (paths relative to
 * /root/reference).  Plain pointers and sizes only; nothing throws or aborts across
 * the ABI; every function returns HUMID_OK (0) or a negative HUMID_E_* code and
 * humid_last_error() gives the text.  There is NO CPU fallback in this library.
 *
 * Packed word: nucleotide i (A0 C1 G2 T3, src/fastq.cc:12) of an n-symbol word
 * (n = -n word length, src/humid.cc:419) occupies bits [2(n-1-i), 2(n-1-i)+1] of a
 * uint64, so unsigned integer order == lexicographic order == Trie::walk() order.
 * n <= 32: one uint64 per read (all BASELINE.json configs use n = 24).
 * 33 <= n <= 64 ("wide" words): TWO uint64 per read, [2r] = the first n-32 nucleotides packed the
 * same way (right-aligned), [2r+1] = the last 32; every words / word array of the single-GPU entry
 * points (humid_dedup_run, humid_dedup_run_device, humid_get_leaves) then holds 2 entries per
 * read / leaf, 16-byte aligned on the device.  n > 64 returns HUMID_E_UNSUPPORTED.  Several GPUs:
 * humid_dedup_run_exchange takes wide words; of the humid_stage_* entry points those of the all-gather mode do
 * (humid_stage_histogram, _count_dense with a filter array, _unique, _graph, _graph_edges,
 * humid_stage_owner_perm_wide: value ranges are then ranges of HEADS, the top 64 bits of a word's 2n-bit value, and
 * the histogram is that of 32-nt words over the heads); the others return HUMID_E_UNSUPPORTED for n > 32.
 */
#ifndef HUMID_HIP_H
#define HUMID_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HUMID_OK             0
#define HUMID_E_INVALID     -1   /* bad argument                                    */
#define HUMID_E_UNSUPPORTED -2   /* word_nt > 64 (stages: > 32)                      */
#define HUMID_E_NOMEM       -3   /* device or host allocation failed                */
#define HUMID_E_HIP         -4   /* HIP runtime error (text in humid_last_error)    */
#define HUMID_E_OVERFLOW    -5   /* an index exceeded 32 bits (reads, 2*edges)      */
#define HUMID_E_STATE       -6   /* accessor called before a successful run         */

#define HUMID_METHOD_DIRECTIONAL 0u  /* default; src/cluster.cc:82-87               */
#define HUMID_METHOD_MAXIMUM     1u  /* -x;      src/cluster.cc:72-80               */

#define HUMID_ABI_VERSION 5u   /* 5: humid_stage_owner_perm_wide, wide words in the all-gather stages; 4: humid_exchange_info.d_unique_degree replaces d_compact_edges (owner-local clustering, round 3); 3: humid_dedup_run_exchange, humid_comm, humid_shm_* (round 2); 2: humid_dedup_run_bases, humid_stage_route ... */

typedef struct humid_ctx humid_ctx;   /* device workspace + stream; not thread-safe */

/* total/usable/unique/clusters are the four lines of stats.dat
 * (src/humid.cc:351-355).  ms_* are device times (hipEvents on the ctx stream). */
typedef struct humid_summary {
  uint64_t total;       /* reads seen                    src/humid.cc:98          */
  uint64_t usable;      /* reads without non-ACGT        src/humid.cc:96          */
  uint64_t unique;      /* distinct words                src/humid.cc:125         */
  uint64_t clusters;    /* clusters.size()               src/humid.cc:403         */
  uint64_t edges;       /* undirected neighbour pairs                             */
  uint64_t nonsingle;   /* unique words with >= 1 neighbour                       */
  /* the four stage times: filled by humid_dedup_run* only with option "kernel_timing" (0 otherwise; round 3: an
   * event record between two kernels costs ~4 us of idle GPU); ms_total and ms_k_insert are always measured       */
  float ms_count;       /* hash insert + unique sort     (Trie::add, walk order)  */
  float ms_neighbours;  /* bucket passes + CSR           (findHammingNeighbours)  */
  float ms_cluster;     /* components + cluster kernel   (findClusters)           */
  float ms_map;         /* per-read map                  (writeFiltered/Annotated)*/
  float ms_total;       /* first kernel to last kernel on the stream              */
  float ms_h2d, ms_d2h; /* host-buffer entry point only                           */
  /* single kernels, HIP events directly around the launches on the ctx stream:      */
  float ms_k_insert;    /* the count kernel: k_dedup_rec / k_dedup_lds / k_hash_insert (one launch) */
  float ms_k_pairs;     /* sum over the 2(d+1) k_pairs launches (count + fill)      */
  float ms_k_cluster;   /* k_cluster_pairs + _small (+ _components): 2-3 launches   */
  float ms_k_map;       /* first kernel of the un-permute: k_unperm_bins (or k_read_map_bucket, _part, k_read_map) */
  float ms_k_part;      /* front partition: the second-level scatter k_pt_scatter<2> (0: library radix passes) */
  float ms_k_unperm;    /* second kernel of the un-permute: k_unperm_window (0: one-kernel forms)   */
  uint32_t count_mode_used;  /* 0 = LDS tables, hashed buckets; 2 = LDS tables, word-ordered buckets (words of
                              * 33-64 nt: buckets on their top 64 bits); 1 = global HBM table (option or
                              * fallback); 3 = sorted (words of 33-64 nt: small inputs, uneven top bits, fallback).
                              * Bit 8 (0x100) is set on top of 2 when the count ran on 8-byte records (round 3). */
} humid_summary;

uint32_t humid_abi_version(void);
int      humid_device_count(void);

/* stream: a hipStream_t (e.g. torch.cuda.current_stream().cuda_stream) or NULL for
 * a stream owned by the context.  device < 0: current device. */
int  humid_ctx_create(humid_ctx **out, int device, void *stream);
void humid_ctx_destroy(humid_ctx *ctx);
const char *humid_last_error(const humid_ctx *ctx);   /* ctx may be NULL */
/* Options.  All but "edit_distance" are tuning knobs that never change results.  "count_mode": 0 = exact counts in hash-partitioned
 * LDS-resident tables (default; falls back to 1 by itself when a bucket overflows), 1 = one
 * open-address table in HBM.  Environment HUMID_COUNT_MODE presets it.
 * "plan_segments": 0 = automatic choice of the pigeonhole plan (s segments, buckets on every
 * combination of s-d of them), else force s (ignored when illegal for the given n, d).
 * "count_order": LDS buckets formed on the word prefix instead of its hash, which makes the unique
 * sort unnecessary: -1 (default) = when a sampled histogram of the top word bits says the fullest
 * bucket fits its LDS table (UMI-first layouts), 0 = never, 1 = always (either way a bucket
 * overflow falls back to hashed buckets).  Words of 33-64 nt: the same choice between LDS tables over
 * buckets of their top 64 bits and the sorting count (0 = always sort; "count_mode" 1 sorts too).
 * "edit_distance": 1 = neighbours under Levenshtein instead of Hamming distance (-e,
 *   findEditNeighbours src/humid.cc:140-158 / Trie::asymmetricLevenshtein) in humid_dedup_run*.
 *   Between equal-length words distance <= 1 is the Hamming search itself; 2 and 3 add the pairs that
 *   need one deletion + one insertion, 4 and 5 those with two of each (banded dynamic programmes); beyond 5
 *   every candidate is verified with the whole dynamic programme (no limit on the distance; the joins' keys
 *   shrink to one segment of n / (d + 1) nucleotides, so the search approaches all pairs, as the trie's does).
 * "coop_big": 1 (default) = components of more than 32 leaves are clustered by one workgroup each
 * (parallel flood), 0 = by one lane each (the literal sequential loop).
 * "tile_partition": 1 (default) = reads reach their count buckets, and results their reads, through
 * the hand-written LDS-staged partition (two coalesced passes each way); 0 = library radix passes
 * in front and one scattered store per read at the end (the round-1 form, also taken by itself for
 * read sets beyond ~180 M / ~67 M reads).
 * "kernel_timing": 1 = HIP events around the single kernels and between the stages, so that
 * humid_summary.ms_k_pairs / ms_k_cluster / ms_k_map / ms_k_part / ms_k_unperm and the stage times ms_count ..
 * ms_map of humid_dedup_run* are filled (default 0, also HUMID_KERNEL_TIMING: the 17 extra event records cost
 * 30-50 us of a 0.7 ms pass; ms_k_insert and ms_total are always there).
 * "padded_partition": 1 (default) = the first level of the tile partition scatters into coarse bins of a
 * fixed room and needs no histogram pass over the reads; a bin that outgrows its room (heavily duplicated
 * words) is detected, the run repeated with the histogram pass, and the option stays 0 for this context.
 * "force_comm": 1 = humid_dedup_run_exchange goes through the humid_comm callbacks even with one rank
 * (a transport can be exercised on a one-GPU box); default 0: with one rank nothing is called or copied.
 * "bucket_walk": how many following words of its pigeonhole bucket a position is compared with by
 * its own thread (default 1024); the pairs further apart inside longer buckets are compared as
 * 1024 x 1024 tiles by whole workgroups.  0 = no bound (every pair by the position's thread, the
 * round-1 form: quadratic per lane on buckets of 10^5 words).  Results do not depend on it. */
int  humid_ctx_set_option(humid_ctx *ctx, const char *key, int64_t value);
/* Optional: one slab of device memory for a run over about n_reads reads, so that the first run does
 * not pay ~35 separate allocations (the `humid` host calls it while pass 1 still parses).  Never
 * fails for lack of memory: without a slab the buffers are allocated one by one as before. */
int  humid_ctx_reserve(humid_ctx *ctx, uint64_t n_reads, uint32_t word_nt);
/* Page-locked host memory for the buffers of humid_dedup_run*: copies from / to it run at the full
 * PCIe rate (pageable buffers are staged by the runtime at a fraction of it: 5 GB/s measured).
 * NULL when the allocation fails -- ordinary memory works everywhere, only slower. */
void *humid_host_alloc(uint64_t bytes);
void  humid_host_free(void *p);

/* ---- the whole hot path ----------------------------------------------------
 * Replaces, between FastQ pass 1 and pass 2:
 *   trie.add(word.data)                 src/humid.cc:94-97   (exact counts)
 *   findHammingNeighbours(trie, d)      src/humid.cc:113-130 (lib/trie walk x asymmetricHamming)
 *     or, option "edit_distance",
 *   findEditNeighbours(trie, d)         src/humid.cc:140-158 (walk x asymmetricLevenshtein)
 *   findClusters(trie, maximum)         src/humid.cc:167-193 + src/cluster.cc:10-87
 *   trie.find(word)->leaf->cluster ...  src/humid.cc:223-231 (keep), :276-277 (cluster id)
 * words[N] (2N uint64 when word_nt > 32), filtered[N] in; cluster_id[N] (0 = filtered, ids 1.. in the order
 * src/humid.cc:177-180 hands them out) and keep[N] (1 = the record writeFiltered
 * emits: the first read, in input order, whose word is its cluster's maxLeaf) out.
 * Host buffers, caller-owned; summary may be NULL. */
int humid_dedup_run(humid_ctx *ctx, const uint64_t *words, const uint8_t *filtered,
                    uint64_t n_reads, uint32_t word_nt, uint32_t distance, uint32_t method,
                    uint32_t *cluster_id, uint8_t *keep, humid_summary *summary);

/* Same contract with DEVICE pointers (inputs already resident in HBM, outputs left
 * in HBM); work is queued on the context's stream and the call returns after the
 * stream has drained.  summary (host) may be NULL. */
int humid_dedup_run_device(humid_ctx *ctx, const uint64_t *d_words, const uint8_t *d_filtered,
                           uint64_t n_reads, uint32_t word_nt, uint32_t distance,
                           uint32_t method, uint32_t *d_cluster_id, uint8_t *d_keep,
                           humid_summary *summary);

/* The same with the word packing on the device (makeWord, src/fastq.cc:146-161): bases[n_reads *
 * word_nt] holds, per record, the word_nt symbols getNucleotides (src/fastq.cc:116-144) assembles --
 * header UMI, then the leading bases of every file's read, 'N' where a read or UMI was short -- as
 * the ASCII bytes of the FastQ.  A/C/G/T -> 0/1/2/3; any other byte counts as 'G' and filters the
 * word (src/fastq.cc:151-158).  humid_get_packed_words returns the words and flags the device made
 * (u64[n_reads], or u64[2 n_reads] above 32 nt; u8[n_reads]). */
int humid_dedup_run_bases(humid_ctx *ctx, const uint8_t *bases, uint64_t n_reads, uint32_t word_nt,
                          uint32_t distance, uint32_t method, uint32_t *cluster_id, uint8_t *keep,
                          humid_summary *summary);
int humid_get_packed_words(humid_ctx *ctx, uint64_t *words, uint8_t *filtered);

/* ---- results of the last run, per unique word in Trie::walk() order ---------
 * (what a caller would read through Result<NLeaf>{leaf,path}, src/humid.cc:117,178,307;
 * NLeaf src/leaf.h:6-9; Cluster src/cluster.h:12-18).  Host output buffers sized by
 * summary.unique / 2*summary.edges / summary.clusters; any pointer may be NULL. */
int humid_get_leaves(humid_ctx *ctx, uint64_t *word, uint32_t *count, uint32_t *first_read,
                     uint32_t *degree, uint32_t *cluster_id, uint8_t *is_max_leaf);
int humid_get_adjacency(humid_ctx *ctx, uint32_t *nbr_off /* unique+1 */,
                        uint32_t *nbr_idx /* 2*edges, each list ascending */);
int humid_get_clusters(humid_ctx *ctx, uint64_t *size, uint32_t *max_count,
                       uint32_t *max_leaf /* walk index of Cluster::maxLeaf */);

/* Histograms of the last run: runStatistics src/humid.cc:301-315 and clusterStats
 * src/cluster.cc:89-95 -> counts.dat / neigh.dat / clusters.dat.
 * which: 0 = leaf->count, 1 = neighbours.size(), 2 = Cluster::size.
 * Writes up to cap (key,value) pairs in ascending key order; *n_out = bins found
 * (call with cap = 0 to size the buffers). */
int humid_get_histogram(humid_ctx *ctx, uint32_t which, uint64_t *keys, uint64_t *values,
                        uint64_t cap, uint64_t *n_out);

/* ---- clustering over an explicit neighbour graph ----------------------------
 * Replaces findClusters (src/humid.cc:167-193) + assignDirectionalCluster /
 * assignMaxCluster (src/cluster.h:26-36) for a caller that built NLeaf::neighbours
 * itself (as tests/test_cluster.cc:11-14 does with link()): leaves are walked in
 * index order, neighbour lists are scanned in the order given.
 * count[U]; nbr_off[U+1], nbr_idx[nbr_off[U]] (CSR, host).  Out: leaf_cluster[U]
 * (ids 1..C), and per cluster id c at slot c-1: size, max_count, max_leaf.
 * cl_* buffers must hold U entries; *n_clusters = C.  Neighbour lists must be symmetric (b in
 * a's list <=> a in b's, as link() and src/humid.cc:121-122 produce) and two linked leaves may
 * not both have count 0 (the reference's maxNeighbour_ never terminates on that input):
 * HUMID_E_INVALID otherwise. */
int humid_cluster_graph(humid_ctx *ctx, const uint32_t *count, const uint32_t *nbr_off,
                        const uint32_t *nbr_idx, uint32_t n_leaves, uint32_t method,
                        uint32_t *leaf_cluster, uint64_t *cl_size, uint32_t *cl_max_count,
                        uint32_t *cl_max_leaf, uint32_t *n_clusters);

/* ---- stages of the same path, for the multi-GPU driver --------------------------
 * One process per GPU (humid_amd/sharded.py, torch.distributed over RCCL): the packed words of
 * all ranks are all-gathered, every rank counts the words of ONE value range
 * (humid_stage_count), the per-range unique arrays are all-gathered (ranges are disjoint and
 * ordered, so their concatenation is Trie::walk() order), neighbours + clusters run over that
 * array (humid_stage_graph) and every rank emits the per-read results of the words it owns
 * (humid_stage_map; 0 elsewhere, so a sum over ranks is the answer).  All pointers are DEVICE
 * pointers unless noted; every call returns after the context's stream has drained.  The
 * reference has no counterpart (it is single-process); semantics per read are those of
 * humid_dedup_run. */

/* usable reads per top-`bits` bin of the word -> d_hist[1 << bits] (u32), for range splitters */
int humid_stage_histogram(humid_ctx *ctx, const uint64_t *d_words, const uint8_t *d_filtered,
                          uint64_t n_reads, uint32_t word_nt, uint32_t bits, uint32_t *d_hist);
/* Trie::add for the reads whose word is in [range_lo, range_hi] (inclusive); expected_reads =
 * upper bound of such reads (0 = n_reads) sizes the table.  n_unique/n_usable: host outputs. */
int humid_stage_count(humid_ctx *ctx, const uint64_t *d_words, const uint8_t *d_filtered,
                      uint64_t n_reads, uint32_t word_nt, uint64_t range_lo, uint64_t range_hi,
                      uint64_t expected_reads, uint64_t *n_unique, uint64_t *n_usable);
/* device pointers (owned by ctx, valid until the next stage_count) of this range's unique words
 * in ascending order, their counts and first read indices */
int humid_stage_unique(humid_ctx *ctx, const uint64_t **d_word, const uint32_t **d_count,
                       const uint32_t **d_first);
/* findHammingNeighbours + findClusters over an ascending unique array; *d_cluster_id /
 * *d_is_max: device arrays (owned by ctx) in the same order.  summary: host. */
int humid_stage_graph(humid_ctx *ctx, const uint64_t *d_g_word, const uint32_t *d_g_count,
                      uint64_t n_unique, uint32_t word_nt, uint32_t distance, uint32_t method,
                      const uint32_t **d_cluster_id, const uint8_t **d_is_max,
                      humid_summary *summary);
/* per-read (cluster_id, keep) for the reads counted by the preceding humid_stage_count, given
 * the cluster ids / maxLeaf flags of ITS unique words (local order); other reads get 0 */
int humid_stage_map(humid_ctx *ctx, const uint32_t *d_local_cluster_id,
                    const uint8_t *d_local_is_max, uint64_t n_reads, uint32_t *d_cluster_id,
                    uint8_t *d_keep);

/* Dense variant of humid_stage_count for a rank of a multi-GPU run: the usable reads of the rank's
 * value range are compacted in read order and counted with the LDS-partitioned tables, exactly like
 * a single-GPU read set.  shard_begin[n_shards+1] (host): the home shards of the gathered reads;
 * counts[q] (host, out) = this rank's reads in shard q = split sizes of the result all-to-all.
 * humid_stage_unique / humid_stage_graph* follow as usual; humid_stage_map_dense then yields the
 * packed results (cluster_id | keep << 31) of those reads in the same dense order, i.e. already
 * laid out as the per-shard streams. */
int humid_stage_count_dense(humid_ctx *ctx, const uint64_t *d_words, const uint8_t *d_filtered,
                            uint64_t n_reads, uint32_t word_nt, uint64_t range_lo, uint64_t range_hi,
                            const uint64_t *shard_begin, uint32_t n_shards, uint64_t *counts,
                            uint64_t *n_unique, uint64_t *n_usable);
int humid_stage_map_dense(humid_ctx *ctx, const uint32_t *d_local_cluster_id,
                          const uint8_t *d_local_is_max, const uint32_t **d_packed,
                          uint64_t *n_packed);

/* Partitioned neighbour search.  Every rank holds the whole ascending unique array (after the
 * all-gather of the per-range arrays); rank part_rank of part_world finds the pairs whose first
 * element lies in its slice -- an equal slice of the positions for the prefix combination, the
 * words whose combination key falls into its part of the key space for the sorted combinations.
 * The union over the ranks is every neighbour pair exactly once.  *d_edges: device array (owned by
 * ctx) of (smaller index << 32 | larger index).  The shares are all-gathered and handed to
 * humid_stage_graph_edges, which is humid_stage_graph with the pairs given instead of searched. */
int humid_stage_pairs(humid_ctx *ctx, const uint64_t *d_g_word, uint64_t n_unique, uint32_t word_nt,
                      uint32_t distance, uint32_t part_rank, uint32_t part_world,
                      const uint64_t **d_edges, uint64_t *n_edges);
int humid_stage_graph_edges(humid_ctx *ctx, const uint64_t *d_g_word, const uint32_t *d_g_count,
                            uint64_t n_unique, const uint64_t *d_edges, uint64_t n_edges,
                            uint32_t word_nt, uint32_t distance, uint32_t method,
                            const uint32_t **d_cluster_id, const uint8_t **d_is_max,
                            humid_summary *summary);

/* Edit distance (-e) on several GPUs (all-gather mode): humid_stage_pairs_edit = this rank's share of
 * the Levenshtein neighbour search (every part_world-th join of the shifted-segment search) over the
 * replicated unique array; shares may repeat a pair, so the gathered list goes through
 * humid_stage_unique_edges (sorted, duplicate-free) before humid_stage_graph_edges. */
int humid_stage_pairs_edit(humid_ctx *ctx, const uint64_t *d_g_word, uint64_t n_unique, uint32_t word_nt,
                           uint32_t distance, uint32_t part_rank, uint32_t part_world,
                           const uint64_t **d_edges, uint64_t *n_edges);
int humid_stage_unique_edges(humid_ctx *ctx, const uint64_t *d_edges, uint64_t n_edges, uint64_t n_unique,
                             const uint64_t **d_unique_edges, uint64_t *n_unique_edges);

/* Result return without N-sized collectives.  The owner of a word computes the results of its
 * reads; the reads' home ranks need them.  Both sides know the same predicate (value ranges), so
 * the streams carry no indices:
 *   humid_stage_owned_results (owner):  packed results (cluster_id | keep << 31) of all reads this
 *     context counted, dense, in read order; counts[q] = how many fall into
 *     [shard_begin[q], shard_begin[q+1]) -- the split sizes of an all-to-all send.
 *   humid_stage_owner_perm (home rank): for its own reads, the owner of each (by range) and the
 *     stable owner-major order: *d_perm[k] = local read of the k-th received result,
 *     counts[o] = reads owned by rank o -- the split sizes of the all-to-all receive.
 *   humid_stage_scatter (home rank):    writes cluster_id/keep of the received stream (0 for
 *     filtered reads).
 * shard_begin, range_lo/hi, counts: host arrays; the rest device pointers (owned by ctx where
 * returned through **).  Requires the global-table count variant on the owner side. */
int humid_stage_owned_results(humid_ctx *ctx, const uint32_t *d_local_cluster_id,
                              const uint8_t *d_local_is_max, const uint64_t *shard_begin,
                              uint32_t n_shards, const uint32_t **d_packed, uint64_t *counts);
int humid_stage_owner_perm(humid_ctx *ctx, const uint64_t *d_words, const uint8_t *d_filtered,
                           uint64_t n_reads, const uint64_t *range_lo, const uint64_t *range_hi,
                           uint32_t n_ranks, const uint32_t **d_perm, uint64_t *counts);
/* the same for words of word_nt nucleotides (33 .. 64: two uint64 per read, ranges of heads; <= 32: as above) */
int humid_stage_owner_perm_wide(humid_ctx *ctx, const uint64_t *d_words, const uint8_t *d_filtered,
                                uint64_t n_reads, uint32_t word_nt, const uint64_t *range_lo, const uint64_t *range_hi,
                                uint32_t n_ranks, const uint32_t **d_perm, uint64_t *counts);
int humid_stage_scatter(humid_ctx *ctx, const uint32_t *d_perm, const uint32_t *d_packed,
                        uint64_t n_recv, uint64_t n_reads, uint32_t *d_cluster_id, uint8_t *d_keep);

/* ---- exchange mode: words travel to the owner of their VALUE range --------------------------
 * (humid_amd/sharded.py, mode "exchange".)  Instead of the all-gather of every word to every
 * rank, each usable read's word goes to the rank that owns its value range (all-to-all, ranges
 * from an all-reduced histogram), that rank counts it (humid_stage_count_dense over the received
 * array), and candidate neighbours meet per pigeonhole combination: the prefix combination is
 * local to a value range (ranges are cut at prefix boundaries: use at most *prefix_bits histogram
 * bits), for every other combination the unique words travel once more, to the rank that owns
 * their combination key (hash of the key; a bucket is never split).  Pairs carry GLOBAL unique
 * indices (rank offset + local walk index); only the pairs -- about 2 % of the reads -- and the
 * counts of their endpoints are replicated, and clustered as a compact graph.
 *   humid_stage_plan_info:   combinations of the pigeonhole plan for plan_unique words in total (all
 *     ranks pass the same number) and the bits of the shortest prefix combination.  Host arithmetic
 *     only: ctx may be NULL (then the automatic plan is reported and no GPU is needed).
 *   humid_stage_combo_route: (word, id | count << 32) items of this rank's ascending unique array
 *     (id = id_base + index) in destination-major order, *d_items[2k] = word, [2k+1] = id | count<<32;
 *     counts[q] = items for rank q.
 *   humid_stage_pairs_keyed: the neighbour pairs among n_items items that share the bucket of
 *     `combo` and were not already found by an earlier combination, as 16-byte records
 *     *d_records[2k] = (smaller id << 32 | larger id), [2k+1] = count(smaller) | count(larger) << 32.
 *     interleaved = 1: d_items as above (any order; d_count ignored); interleaved = 0: a plain
 *     ascending word array with ids id_base + index and counts d_count (combination 0 only).
 *   humid_stage_compact_nodes: the distinct endpoints of an edge list (ascending), the same edges
 *     over positions in that list and (record_stride 2: the records above) the endpoints' counts --
 *     the input of humid_stage_graph_edges for a graph that leaves out the singletons (every
 *     singleton is its own cluster and its own maxLeaf).  record_stride 1: plain (a << 32 | b) edges.
 *   humid_stage_route_words: the words of this rank's usable reads in the owner-major order of the
 *     preceding humid_stage_owner_perm (the all-to-all send buffer).
 *   humid_stage_exchange_ids: cluster id + maxLeaf flag of this rank's unique words (global walk
 *     indices id_base ..) from the compact graph's results; ids count the cluster-creating leaves
 *     before a leaf in the WHOLE walk (src/humid.cc:177-180): singletons + compact creators.
 * humid_stage_count_dense accepts d_filtered = NULL: every read is usable and lies in
 * [range_lo, range_hi] (checked); the array is counted as it stands and the range only shapes the
 * word-ordered LDS buckets. */
/* HIP-event durations of the dominant kernels of the last humid_stage_count_dense /
 * humid_stage_map_dense pair (k_dedup_lds or k_hash_insert; k_read_map_bucket, k_read_map_part or k_read_map_packed) */
int humid_stage_kernel_ms(humid_ctx *ctx, float *ms_k_insert, float *ms_k_map, uint32_t *count_mode_used);
int humid_stage_route_words(humid_ctx *ctx, const uint64_t *d_words, uint64_t n_send,
                            const uint64_t **d_routed);
/* The same routing without a host wait and without a sort: the caller already knows how many reads go
 * to every owner (send_counts[q]: from the all-gathered per-rank histograms, whose bins the value
 * ranges are cut at).  *d_routed = the usable words in owner-major order, INPUT ORDER inside every
 * owner's block (the owner derives "first read of a word" from it); *d_perm = routed position -> read
 * index, for humid_stage_scatter.  Queued on the context's stream.  humid_stage_route_check waits
 * for the stream and returns HUMID_E_INVALID if the counts did not match the reads. */
int humid_stage_route(humid_ctx *ctx, const uint64_t *d_words, const uint8_t *d_filtered, uint64_t n_reads,
                      const uint64_t *range_lo, const uint64_t *range_hi, uint32_t n_ranks,
                      const uint64_t *send_counts, const uint64_t **d_routed, const uint32_t **d_perm);
int humid_stage_route_check(humid_ctx *ctx);
int humid_stage_exchange_ids(humid_ctx *ctx, const uint32_t *d_nodes, const uint32_t *d_compact_cluster_id,
                             const uint8_t *d_compact_is_max, uint64_t n_nodes, uint64_t n_clusters,
                             uint64_t id_base, uint64_t u_local, const uint32_t **d_local_cluster_id,
                             const uint8_t **d_local_is_max);
int humid_stage_plan_info(humid_ctx *ctx, uint32_t word_nt, uint32_t distance, uint64_t plan_unique,
                          uint32_t *n_combos, uint32_t *prefix_bits);
int humid_stage_combo_route(humid_ctx *ctx, const uint64_t *d_word, const uint32_t *d_count,
                            uint64_t n_unique, uint64_t id_base, uint32_t word_nt, uint32_t distance,
                            uint64_t plan_unique, uint32_t combo, uint32_t n_ranks,
                            const uint64_t **d_items, uint64_t *counts);
int humid_stage_pairs_keyed(humid_ctx *ctx, const uint64_t *d_items, uint64_t n_items, int interleaved,
                            uint64_t id_base, const uint32_t *d_count, uint32_t word_nt,
                            uint32_t distance, uint64_t plan_unique, uint32_t combo,
                            const uint64_t **d_records, uint64_t *n_edges);
int humid_stage_compact_nodes(humid_ctx *ctx, const uint64_t *d_edges, uint64_t n_edges,
                              uint32_t record_stride, const uint32_t **d_nodes, uint64_t *n_nodes,
                              const uint64_t **d_compact_edges, const uint32_t **d_node_counts);

/* ---- the whole exchange-mode pass of one rank in ONE call --------------------------------------
 * humid_dedup_run_exchange runs, for this rank's shard of the reads (input order), the sequence the
 * stage entry points above make up -- histogram, value ranges, word exchange, counts, pairs per
 * combination, compact graph, result exchange, scatter -- with everything between the exchanges inside
 * the library (persistent buffers, no host language in between).  What only the caller can do, moving
 * bytes between ranks, it does through humid_comm:
 *   host_all_gather: every rank contributes `bytes` bytes of HOST memory; all[world * bytes] receives the
 *     contributions in rank order (blocking).
 *   exchange: DEVICE memory; rank q is sent d_send[send_off[q] .. + send_bytes[q]) and what rank q sends
 *     here arrives at d_recv[recv_off[q] ..), recv_bytes[q] long -- for every q including this rank
 *     itself.  Two shapes occur: an all-to-all (all_gather = 0: both sides laid out in rank order,
 *     send_off and recv_off the running sums of the sizes) and an all-gather (all_gather = 1: send_off
 *     all 0, send_bytes all equal: the same bytes go to everybody).  The transfer must be ordered after the work queued on `stream` (the
 *     context's stream) and either complete or be ordered before later work on that stream when the
 *     call returns (grouped ncclSend/ncclRecv on `stream` do exactly that).
 * Both return 0 or a negative value, which ends the run with HUMID_E_COMM.  With world == 1 neither is
 * called (comm may then be NULL).  Every rank must make the call with the same word_nt, distance and
 * method.  *summary receives the totals of the WHOLE read set (identical on all ranks; ms_* are this
 * rank's); *info what a caller needs for the statistics files.  word_nt 1 .. 64 (above 32: two uint64 per
 * read, as in humid_dedup_run).  Limits: at most 16 ranks, a pigeonhole plan with a prefix (distance <
 * word_nt): HUMID_E_UNSUPPORTED otherwise. */
#define HUMID_E_COMM        -7   /* a humid_comm callback failed                        */
typedef struct humid_comm {
  void *user;
  uint32_t rank, world;
  int (*host_all_gather)(void *user, const void *mine, uint64_t bytes, void *all);
  int (*exchange)(void *user, const void *d_send, const uint64_t *send_off, const uint64_t *send_bytes,
                  void *d_recv, const uint64_t *recv_off, const uint64_t *recv_bytes, int all_gather,
                  void *stream);
} humid_comm;
typedef struct humid_exchange_info {
  uint64_t unique_local;             /* unique words this rank owns (its value range)             */
  uint64_t id_base;                  /* walk index of the first of them                           */
  uint64_t n_nodes;                  /* unique words with neighbours, all ranks                   */
  uint64_t n_pairs;                  /* neighbour pairs, all ranks                                */
  const uint32_t *d_unique_count;    /* device: counts of this rank's unique words (counts.dat)   */
  const uint32_t *d_unique_degree;   /* device: neighbours of every one of them (neigh.dat); since ABI 4: every rank clusters
                                      * its own components (+ the replicated ones that cross value ranges), no rank holds all pairs */
} humid_exchange_info;
int humid_dedup_run_exchange(humid_ctx *ctx, const humid_comm *comm, const uint64_t *d_words,
                             const uint8_t *d_filtered, uint64_t n_local, uint32_t word_nt,
                             uint32_t distance, uint32_t method, uint32_t *d_cluster_id, uint8_t *d_keep,
                             humid_summary *summary, humid_exchange_info *info);

/* A ready-made humid_comm.host_all_gather for ranks that are PROCESSES OF ONE NODE: a POSIX shared-memory
 * segment with one slot and one arrival counter per rank (a gather is a store into the own slot, a
 * release of the counter and a spin on the others': a few microseconds, where a collective of the
 * process group costs ~100 us of launches and waits for a 16 KB table).  Rank 0 creates the segment
 * `name` (e.g. "/humid_<pid>"), the others attach to it (they wait up to ~30 s for it to appear); every
 * humid_shm_open is COLLECTIVE (rank 0 waits until every rank has attached to ITS segment: a rank that mapped a
 * stale segment of the same name, left by a crashed run, notices and maps again).  Every
 * rank then passes humid_shm_all_gather as host_all_gather and its handle as `user` -- or, when it needs
 * `user` for its own exchange callback, calls humid_shm_all_gather(handle, ...) from its own wrapper.
 * A gather is at most slot_bytes per rank (humid_dedup_run_exchange needs 16 KB: the histogram table). */
typedef struct humid_shm humid_shm;
int  humid_shm_open(humid_shm **out, const char *name, uint32_t rank, uint32_t world, uint64_t slot_bytes);
int  humid_shm_all_gather(void *shm, const void *mine, uint64_t bytes, void *all);
void humid_shm_abort(humid_shm *shm);    /* this rank gives the group up: every rank's gathers return -1 from now on */
void humid_shm_close(humid_shm *shm);

/* src/cluster.cc:31-33 atLeastDouble_, evaluated on the device (parity probe). */
int humid_at_least_double(humid_ctx *ctx, uint64_t a, uint64_t b, int *result);

#ifdef __cplusplus
}
#endif
#endif
