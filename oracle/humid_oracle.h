/*
 * humid_oracle.h -- CPU restatement of HUMID's neighbour-search-and-cluster path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is product code: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, load or call
 * it, and only as the checker / the reported CPU baseline.  The product path
 * (humid_amd/, include/humid_hip.h) never links or imports it.
 *
 * PINNING STATUS
 *  - clustering half (src/cluster.cc, src/leaf.h, src/humid.cc:167-193): pinned by
 *    every known answer in the reference's tests/test_cluster.cc:17-137
 *    (re-encoded as data in tests/golden/ref_test_cluster.json).
 *  - word extraction (src/fastq.cc:57-93,116-161,192-230): pinned by every known
 *    answer in tests/test_fastq.cc:9-202 (tests/golden/ref_test_fastq.json).
 *  - trie half (lib/trie = jfjlaros/trie, un-vendored git submodule, EMPTY in
 *    /root/reference, pinned SHA not recorded anywhere in the tree): PARITY
 *    UNPINNED.  The reference holds no test, fixture or golden vector that touches
 *    Trie::add/walk/asymmetricHamming/find.  Set membership (Hamming distance over
 *    nucleotides <= d, self excluded by src/humid.cc:120) and count semantics are
 *    fixed by the call sites; ORDER (walk = ascending lexicographic, asymmetric
 *    search = words >= query in ascending order) restates the library's published
 *    algorithm (4-ary trie, children visited in index order) and is isolated in
 *    orc_walk()/orc_asym_hamming() below.
 *  - edit distance (-e, src/humid.cc:140-158): the same un-vendored library; PARITY UNPINNED.
 *    Membership (Levenshtein distance between equal-length words <= d) is fixed by the call
 *    site; ORDER and MULTIPLICITY (hypothesis H3: every pair reported once, ascending) restate
 *    the textbook trie search, see orc_find_edit_neighbours().
 *  - The reference as a whole is UNBUILDABLE here (four empty submodules, -lisal
 *    absent); no stand-in headers were written, so there is no oracle/_ref.
 *
 * Packed-word layout (the C-ABI's, not the reference's): nucleotide i of an
 * n-symbol word (A0 C1 G2 T3, src/fastq.cc:12) sits in bits [2(n-1-i), 2(n-1-i)+1]
 * of a uint64 (n <= 32), so integer order == lexicographic order == trie order.
 * 33 <= n <= 64: two uint64 per word, [0] = the first n-32 nucleotides (right-aligned),
 * [1] = the last 32; every `words` array then holds 2 entries per read / leaf.
 */
#ifndef HUMID_ORACLE_H
#define HUMID_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- src/cluster.h:12-18, src/leaf.h:6-9 ---------------------------------- */
typedef struct OCluster {
  size_t id;
  size_t maxCount;
  struct OLeaf *maxLeaf;
  size_t size;
  int visited;
} OCluster;

typedef struct OLeaf {
  size_t count;            /* lib/trie Leaf::count                         */
  struct OLeaf **nbr;      /* NLeaf::neighbours (std::vector, push order)  */
  size_t nn, ncap;
  OCluster *cluster;       /* NLeaf::cluster                               */
  uint32_t rank;           /* oracle bookkeeping: index in walk order      */
} OLeaf;

/* ---- src/cluster.cc -------------------------------------------------------- */
int    orc_at_least_double(size_t a, size_t b);                 /* :31-33 */
OLeaf *orc_max_neighbour(OLeaf *leaf);                          /* :39-51 */
void   orc_assign_max_cluster(OLeaf *leaf, OCluster *cluster);  /* :72-80 */
void   orc_assign_directional_cluster(OLeaf *leaf, OCluster *c);/* :82-87 */

/* ---- hand-built graphs, as tests/test_cluster.cc:11-14 builds them --------- */
typedef struct orc_graph orc_graph;
orc_graph *orc_graph_create(size_t n_leaves);
void   orc_graph_destroy(orc_graph *g);
void   orc_graph_set_count(orc_graph *g, size_t leaf, size_t count);
void   orc_graph_link(orc_graph *g, size_t a, size_t b);        /* link()  */
void   orc_graph_preassign(orc_graph *g, size_t leaf, size_t cluster_id);
size_t orc_graph_max_neighbour(orc_graph *g, size_t leaf);
/* one explicit call of assign{Directional,Max}Cluster(leaf, new Cluster{id}) */
void   orc_graph_assign(orc_graph *g, size_t leaf, size_t cluster_id, int maximum);
/* the findClusters loop (src/humid.cc:176-189) over leaves in index order    */
size_t orc_graph_find_clusters(orc_graph *g, int maximum);
/* per leaf: cluster id (0 = none); per cluster id c (1-based) slot c-1:      */
void   orc_graph_export(orc_graph *g, uint32_t *leaf_cluster,
                        uint64_t *cl_size, uint64_t *cl_max_count,
                        int64_t *cl_max_leaf, size_t cl_cap);

/* ---- src/fastq.cc word extraction ------------------------------------------ */
size_t orc_make_string_size(const char *s, size_t size, char pad, char *out); /* :57-66 */
size_t orc_extract_last_field(const char *s, char sep, char *out);            /* :192-199 */
int    orc_valid_umi(const char *umi);                                        /* :201-214 */
size_t orc_extract_umi(const char *header, char *out);                        /* :72-93  */
void   orc_nt_from_file(size_t files, size_t length, size_t *out);            /* :220-230 */
/* getNucleotides :116-144; returns number of chars written to out */
size_t orc_get_nucleotides(const char *first_header, const char *const *seqs,
                           size_t n_files, const size_t *nt_to_take,
                           size_t header_umi_size, char *out);
/* makeWord :146-161; data[i] in 0..3; returns filtered flag */
int    orc_make_word(const char *nucleotides, size_t n, uint8_t *data);
uint64_t orc_pack_word(const uint8_t *data, size_t n);          /* C-ABI layout */
/* preCompute src/humid.cc:38-59 */
void   orc_pre_compute(size_t first_header_umi, size_t n_files, size_t word_length,
                       size_t *header_umi_size, size_t *nt_to_take);

/* ---- the trie + the src/humid.cc pipeline ---------------------------------- */
typedef struct orc_ctx orc_ctx;
orc_ctx *orc_create(uint32_t word_nt);
void     orc_destroy(orc_ctx *c);
/* readData loop src/humid.cc:92-99: trie.add for every non-filtered word */
void     orc_read_data(orc_ctx *c, const uint64_t *words, const uint8_t *filtered,
                       uint64_t n_reads);
uint64_t orc_find_hamming_neighbours(orc_ctx *c, uint32_t distance); /* :113-130 */
uint64_t orc_find_edit_neighbours(orc_ctx *c, uint32_t distance);    /* :140-158 (-e) */
uint64_t orc_find_clusters(orc_ctx *c, int maximum);                 /* :167-193 */
/* writeFiltered :220-234 (keep) + writeAnnotated :268-285 (cluster_id) */
void     orc_map_reads(orc_ctx *c, const uint64_t *words, const uint8_t *filtered,
                       uint64_t n_reads, uint32_t *cluster_id, uint8_t *keep);

uint64_t orc_total(const orc_ctx *c);
uint64_t orc_usable(const orc_ctx *c);
uint64_t orc_unique(const orc_ctx *c);
uint64_t orc_n_clusters(const orc_ctx *c);
uint64_t orc_n_edges(const orc_ctx *c);   /* undirected pairs */
/* per unique word, in walk order; any pointer may be NULL */
void     orc_export_leaves(const orc_ctx *c, uint64_t *word, uint64_t *count,
                           uint32_t *degree, uint32_t *cluster_id, uint8_t *is_max_leaf);
/* CSR in walk order; nbr_off has unique+1 entries, nbr_idx has 2*edges */
void     orc_export_adjacency(const orc_ctx *c, uint64_t *nbr_off, uint32_t *nbr_idx);
/* per cluster id (slot id-1) */
void     orc_export_clusters(const orc_ctx *c, uint64_t *size, uint64_t *max_count,
                             uint32_t *max_leaf_rank);

/* one-call convenience: read -> neighbours -> clusters -> map; returns 0.
 * method bit 0: maximum clustering (-x); bit 1: Levenshtein instead of Hamming neighbours (-e).
 * summary5 (may be NULL): total, usable, unique, clusters, neighbour pairs.
 * phase_seconds (may be NULL): [0] read+count, [1] neighbours, [2] clusters, [3] map */
int orc_dedup_run(const uint64_t *words, const uint8_t *filtered, uint64_t n_reads,
                  uint32_t word_nt, uint32_t distance, uint32_t method,
                  uint32_t *cluster_id, uint8_t *keep, uint64_t *summary5,
                  double *phase_seconds);

#ifdef __cplusplus
}
#endif
#endif
