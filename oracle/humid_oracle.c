/*
 * humid_oracle.c -- CPU restatement of HUMID's hot path.  TEST INFRASTRUCTURE ONLY;
 * see humid_oracle.h for who may use it and for the pinning status ("parity
 * unpinned" for the lib/trie half).  Plain C11, single thread like the reference.
 *
 * Every function cites the reference lines (relative to /root/reference) it follows.
 * Recursions of the reference are run on an explicit stack with the SAME visiting
 * order (the reference overflows the machine stack on deep graphs,
 * docs/troubleshooting.rst:6-18).
 */
#define _POSIX_C_SOURCE 200809L
#include "humid_oracle.h"

#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------------- */
/* small helpers                                                             */
/* ------------------------------------------------------------------------- */
static void *xmalloc(size_t n) {
  void *p = malloc(n ? n : 1);
  if (!p) abort();
  return p;
}
static void *xcalloc(size_t n, size_t m) {
  void *p = calloc(n ? n : 1, m ? m : 1);
  if (!p) abort();
  return p;
}
static void *xrealloc(void *q, size_t n) {
  void *p = realloc(q, n ? n : 1);
  if (!p) abort();
  return p;
}
static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* std::vector<NLeaf*>::push_back */
static void leaf_push(OLeaf *l, OLeaf *n) {
  if (l->nn == l->ncap) {
    l->ncap = l->ncap ? 2 * l->ncap : 2;
    l->nbr = (OLeaf **)xrealloc(l->nbr, l->ncap * sizeof(OLeaf *));
  }
  l->nbr[l->nn++] = n;
}

/* ------------------------------------------------------------------------- */
/* src/cluster.cc                                                            */
/* ------------------------------------------------------------------------- */

/* src/cluster.cc:10-13 assignLeaf_ */
static void assign_leaf_(OLeaf *leaf, OCluster *cluster) {
  leaf->cluster = cluster;
  leaf->cluster->size += leaf->count;
}

/* src/cluster.cc:20-25 updateMaxCount_ (strict >) */
static void update_max_count_(OLeaf *leaf, OCluster *cluster) {
  if (leaf->count > cluster->maxCount) {
    cluster->maxLeaf = leaf;
    cluster->maxCount = leaf->count;
  }
}

/* src/cluster.cc:31-33 atLeastDouble_ (size_t arithmetic) */
int orc_at_least_double(size_t a, size_t b) { return a >= 2 * b; }

/* src/cluster.cc:39-51 maxNeighbour_: first qualifying neighbour in list order,
 * then restart the scan at the new leaf. */
OLeaf *orc_max_neighbour(OLeaf *leaf) {
  size_t i = 0;
  while (i < leaf->nn) {
    OLeaf *neighbour = leaf->nbr[i++];
    if (!neighbour->cluster && orc_at_least_double(neighbour->count, leaf->count)) {
      leaf = neighbour;
      i = 0;
    }
  }
  return leaf;
}

typedef struct { OLeaf *leaf; size_t i; } Frame;
typedef struct { Frame *f; size_t n, cap; } FStack;
static void fs_push(FStack *s, OLeaf *l) {
  if (s->n == s->cap) {
    s->cap = s->cap ? 2 * s->cap : 64;
    s->f = (Frame *)xrealloc(s->f, s->cap * sizeof(Frame));
  }
  s->f[s->n].leaf = l;
  s->f[s->n].i = 0;
  s->n++;
}

/* src/cluster.cc:58-69 assignDirectionalCluster_ -- same pre-order, explicit stack */
static void assign_directional_cluster_(OLeaf *leaf, OCluster *cluster) {
  FStack st = {0, 0, 0};
  assign_leaf_(leaf, cluster);
  fs_push(&st, leaf);
  while (st.n) {
    Frame *top = &st.f[st.n - 1];
    OLeaf *cur = top->leaf;
    int descended = 0;
    while (top->i < cur->nn) {
      OLeaf *neighbour = cur->nbr[top->i++];
      if (!neighbour->cluster && orc_at_least_double(cur->count, neighbour->count)) {
        assign_leaf_(neighbour, cluster);
        fs_push(&st, neighbour);
        descended = 1;
        break;
      }
    }
    if (!descended) st.n--;
  }
  free(st.f);
}

/* src/cluster.cc:72-80 assignMaxCluster -- same pre-order, explicit stack */
void orc_assign_max_cluster(OLeaf *leaf, OCluster *cluster) {
  FStack st = {0, 0, 0};
  assign_leaf_(leaf, cluster);
  update_max_count_(leaf, cluster);
  fs_push(&st, leaf);
  while (st.n) {
    Frame *top = &st.f[st.n - 1];
    OLeaf *cur = top->leaf;
    int descended = 0;
    while (top->i < cur->nn) {
      OLeaf *neighbour = cur->nbr[top->i++];
      if (!neighbour->cluster) {
        assign_leaf_(neighbour, cluster);
        update_max_count_(neighbour, cluster);
        fs_push(&st, neighbour);
        descended = 1;
        break;
      }
    }
    if (!descended) st.n--;
  }
  free(st.f);
}

/* src/cluster.cc:82-87 assignDirectionalCluster */
void orc_assign_directional_cluster(OLeaf *leaf, OCluster *cluster) {
  OLeaf *node = orc_max_neighbour(leaf);
  update_max_count_(node, cluster);
  assign_directional_cluster_(node, cluster);
}

/* ------------------------------------------------------------------------- */
/* hand-built graphs (tests/test_cluster.cc style)                           */
/* ------------------------------------------------------------------------- */
struct orc_graph {
  size_t n;
  OLeaf *leaf;
  OCluster **cl;     /* cl[id] for ids handed out, sparse by id */
  size_t cl_cap;
  size_t max_id;
};

orc_graph *orc_graph_create(size_t n) {
  orc_graph *g = (orc_graph *)xcalloc(1, sizeof(*g));
  g->n = n;
  g->leaf = (OLeaf *)xcalloc(n, sizeof(OLeaf));
  for (size_t i = 0; i < n; i++) g->leaf[i].rank = (uint32_t)i;
  return g;
}
void orc_graph_destroy(orc_graph *g) {
  if (!g) return;
  for (size_t i = 0; i < g->n; i++) free(g->leaf[i].nbr);
  for (size_t i = 0; i < g->cl_cap; i++) free(g->cl[i]);
  free(g->cl);
  free(g->leaf);
  free(g);
}
void orc_graph_set_count(orc_graph *g, size_t l, size_t c) { g->leaf[l].count = c; }
/* tests/test_cluster.cc:11-14 link() */
void orc_graph_link(orc_graph *g, size_t a, size_t b) {
  leaf_push(&g->leaf[a], &g->leaf[b]);
  leaf_push(&g->leaf[b], &g->leaf[a]);
}
static OCluster *graph_cluster(orc_graph *g, size_t id) {
  if (id >= g->cl_cap) {
    size_t nc = g->cl_cap ? g->cl_cap : 8;
    while (nc <= id) nc *= 2;
    g->cl = (OCluster **)xrealloc(g->cl, nc * sizeof(OCluster *));
    for (size_t i = g->cl_cap; i < nc; i++) g->cl[i] = NULL;
    g->cl_cap = nc;
  }
  if (!g->cl[id]) {
    g->cl[id] = (OCluster *)xcalloc(1, sizeof(OCluster));
    g->cl[id]->id = id;
  }
  if (id > g->max_id) g->max_id = id;
  return g->cl[id];
}
/* tests/test_cluster.cc:36-38: a neighbour that already sits in a cluster */
void orc_graph_preassign(orc_graph *g, size_t l, size_t id) {
  g->leaf[l].cluster = graph_cluster(g, id);
}
size_t orc_graph_max_neighbour(orc_graph *g, size_t l) {
  return (size_t)(orc_max_neighbour(&g->leaf[l]) - g->leaf);
}
void orc_graph_assign(orc_graph *g, size_t l, size_t id, int maximum) {
  OCluster *c = graph_cluster(g, id);
  if (maximum) orc_assign_max_cluster(&g->leaf[l], c);
  else orc_assign_directional_cluster(&g->leaf[l], c);
}
/* src/humid.cc:176-189 with walk order == leaf index order */
size_t orc_graph_find_clusters(orc_graph *g, int maximum) {
  size_t id = 1;
  for (size_t i = 0; i < g->n; i++) {
    if (!g->leaf[i].cluster) orc_graph_assign(g, i, id++, maximum);
  }
  return id - 1;
}
void orc_graph_export(orc_graph *g, uint32_t *leaf_cluster, uint64_t *cl_size,
                      uint64_t *cl_max_count, int64_t *cl_max_leaf, size_t cl_cap) {
  if (leaf_cluster)
    for (size_t i = 0; i < g->n; i++)
      leaf_cluster[i] = g->leaf[i].cluster ? (uint32_t)g->leaf[i].cluster->id : 0;
  for (size_t id = 1; id <= cl_cap; id++) {
    OCluster *c = (id < g->cl_cap) ? g->cl[id] : NULL;
    if (cl_size) cl_size[id - 1] = c ? c->size : 0;
    if (cl_max_count) cl_max_count[id - 1] = c ? c->maxCount : 0;
    if (cl_max_leaf) cl_max_leaf[id - 1] = (c && c->maxLeaf) ? (int64_t)(c->maxLeaf - g->leaf) : -1;
  }
}

/* ------------------------------------------------------------------------- */
/* src/fastq.cc word extraction                                              */
/* ------------------------------------------------------------------------- */

/* src/fastq.cc:12 nuc map; -1 = not in map */
static int nuc_code(char c) {
  switch (c) {
    case 'A': return 0;
    case 'C': return 1;
    case 'G': return 2;
    case 'T': return 3;
    default: return -1;
  }
}

/* src/fastq.cc:57-66 makeStringSize_ */
size_t orc_make_string_size(const char *s, size_t size, char pad, char *out) {
  size_t len = strlen(s);
  size_t i = 0;
  for (; i < size && i < len; i++) out[i] = s[i];
  for (; i < size; i++) out[i] = pad;
  out[size] = 0;
  return size;
}

/* src/fastq.cc:192-199 extractLastField */
size_t orc_extract_last_field(const char *s, char sep, char *out) {
  const char *last = strrchr(s, sep);
  if (!last) { out[0] = 0; return 0; }
  size_t n = strlen(last + 1);
  memcpy(out, last + 1, n + 1);
  return n;
}

/* src/fastq.cc:201-214 validUMI: non-empty, only ACGT */
int orc_valid_umi(const char *umi) {
  if (!umi[0]) return 0;
  for (const char *p = umi; *p; p++)
    if (nuc_code(*p) < 0) return 0;
  return 1;
}

/* src/fastq.cc:72-93 extractUMI_ */
size_t orc_extract_umi(const char *header, char *out) {
  size_t first_space = strcspn(header, " ");
  char *sub = (char *)xmalloc(first_space + 1);
  memcpy(sub, header, first_space);
  sub[first_space] = 0;
  size_t n = orc_extract_last_field(sub, '_', out);
  if (orc_valid_umi(out)) { free(sub); return n; }
  n = orc_extract_last_field(sub, ':', out);
  if (orc_valid_umi(out)) { free(sub); return n; }
  free(sub);
  out[0] = 0;
  return 0;
}

/* src/fastq.cc:220-230 ntFromFile: equal split, remainder to the LAST file */
void orc_nt_from_file(size_t files, size_t length, size_t *out) {
  size_t div = length / files;
  for (size_t i = 0; i + 1 < files; i++) out[i] = div;
  out[files - 1] = div + length % files;
}

/* src/fastq.cc:116-144 getNucleotides */
size_t orc_get_nucleotides(const char *first_header, const char *const *seqs,
                           size_t n_files, const size_t *nt_to_take,
                           size_t header_umi_size, char *out) {
  size_t k = 0;
  if (header_umi_size > 0) {
    char *umi = (char *)xmalloc(strlen(first_header) + 1);
    orc_extract_umi(first_header, umi);
    char *sized = (char *)xmalloc(header_umi_size + 1);
    orc_make_string_size(umi, header_umi_size, 'N', sized);
    for (size_t i = 0; i < header_umi_size; i++) out[k++] = sized[i];
    free(sized);
    free(umi);
  }
  for (size_t f = 0; f < n_files; f++) {
    size_t length = nt_to_take[f];
    char *sized = (char *)xmalloc(length + 1);
    orc_make_string_size(seqs[f], length, 'N', sized);
    for (size_t p = 0; p < length; p++) out[k++] = sized[p];
    free(sized);
  }
  out[k] = 0;
  return k;
}

/* src/fastq.cc:146-161 makeWord: unknown char -> code of 'G' and filtered */
int orc_make_word(const char *nucleotides, size_t n, uint8_t *data) {
  int filtered = 0;
  for (size_t i = 0; i < n; i++) {
    int c = nuc_code(nucleotides[i]);
    if (c >= 0) data[i] = (uint8_t)c;
    else { data[i] = 2; filtered = 1; }
  }
  return filtered;
}

uint64_t orc_pack_word(const uint8_t *data, size_t n) {
  uint64_t w = 0;
  for (size_t i = 0; i < n; i++) w = (w << 2) | (uint64_t)(data[i] & 3);
  return w;
}

/* src/humid.cc:38-59 preCompute (peekUMI result passed in) */
void orc_pre_compute(size_t first_header_umi, size_t n_files, size_t word_length,
                     size_t *header_umi_size, size_t *nt_to_take) {
  size_t h = first_header_umi;
  size_t from_file = 0;
  if (word_length > h) from_file = word_length - h;
  orc_nt_from_file(n_files, from_file, nt_to_take);
  if (word_length < h) h = word_length;
  *header_umi_size = h;
}

/* ------------------------------------------------------------------------- */
/* lib/trie restated: Trie<4, NLeaf> (PARITY UNPINNED, see header)           */
/* ------------------------------------------------------------------------- */
typedef struct ONode {
  struct ONode *child[4];
  OLeaf *leaf;
} ONode;

#define ARENA_BLOCK (1u << 16)
typedef struct Arena {
  void **blocks;
  size_t nblocks, capblocks;
  size_t used;      /* items used in last block */
  size_t item;
} Arena;
static void *arena_new(Arena *a) {
  if (a->nblocks == 0 || a->used == ARENA_BLOCK) {
    if (a->nblocks == a->capblocks) {
      a->capblocks = a->capblocks ? 2 * a->capblocks : 16;
      a->blocks = (void **)xrealloc(a->blocks, a->capblocks * sizeof(void *));
    }
    a->blocks[a->nblocks++] = xcalloc(ARENA_BLOCK, a->item);
    a->used = 0;
  }
  return (char *)a->blocks[a->nblocks - 1] + (a->used++) * a->item;
}
static void arena_free(Arena *a) {
  for (size_t i = 0; i < a->nblocks; i++) free(a->blocks[i]);
  free(a->blocks);
}

struct orc_ctx {
  uint32_t n;          /* word length in nucleotides */
  uint32_t wpr;        /* uint64 per word: 1 (n <= 32) or 2 */
  ONode *root;
  Arena nodes, leaves;
  uint64_t total, usable, unique, edges;
  OLeaf **walk;        /* leaves in walk order (filled by find_hamming_neighbours) */
  OCluster **clusters; /* src/humid.cc:176 vector<Cluster*> */
  size_t n_clusters, cap_clusters;
};

/* symbol i (0 = first nucleotide) of a packed word.  n <= 32: one uint64.  33 <= n <= 64: two,
 * w[0] = the first n-32 nucleotides (right-aligned), w[1] = the last 32 (the C ABI's layout). */
static inline unsigned sym(const uint64_t *w, uint32_t n, uint32_t i) {
  if (n <= 32) return (unsigned)((w[0] >> (2u * (n - 1u - i))) & 3u);
  const uint32_t nh = n - 32;
  if (i < nh) return (unsigned)((w[0] >> (2u * (nh - 1u - i))) & 3u);
  return (unsigned)((w[1] >> (2u * (n - 1u - i))) & 3u);
}

orc_ctx *orc_create(uint32_t n) {
  if (n == 0 || n > 64) return NULL;
  orc_ctx *c = (orc_ctx *)xcalloc(1, sizeof(*c));
  c->n = n;
  c->wpr = n > 32 ? 2 : 1;
  c->nodes.item = sizeof(ONode);
  c->leaves.item = sizeof(OLeaf);
  c->root = (ONode *)arena_new(&c->nodes);
  return c;
}

static void walk_leaves(const orc_ctx *c, void (*fn)(OLeaf *, const uint64_t *, void *), void *arg);

static void free_leaf_cb(OLeaf *l, const uint64_t *w, void *arg) {
  (void)w; (void)arg;
  free(l->nbr);
}
void orc_destroy(orc_ctx *c) {
  if (!c) return;
  walk_leaves(c, free_leaf_cb, NULL);
  for (size_t i = 0; i < c->n_clusters; i++) free(c->clusters[i]);  /* freeClusters cluster.cc:97-101 */
  free(c->clusters);
  free(c->walk);
  arena_free(&c->nodes);
  arena_free(&c->leaves);
  free(c);
}

/* Trie::add (call site src/humid.cc:95): descend/create n nodes, leaf->count++ */
static OLeaf *trie_add(orc_ctx *c, const uint64_t *w) {
  ONode *node = c->root;
  for (uint32_t i = 0; i < c->n; i++) {
    unsigned s = sym(w, c->n, i);
    if (!node->child[s]) node->child[s] = (ONode *)arena_new(&c->nodes);
    node = node->child[s];
  }
  if (!node->leaf) {
    node->leaf = (OLeaf *)arena_new(&c->leaves);
    c->unique++;
  }
  node->leaf->count++;
  return node->leaf;
}

/* Trie::find (call sites src/humid.cc:223,276) */
static OLeaf *trie_find(const orc_ctx *c, const uint64_t *w) {
  const ONode *node = c->root;
  for (uint32_t i = 0; i < c->n; i++) {
    node = node->child[sym(w, c->n, i)];
    if (!node) return NULL;
  }
  return node->leaf;
}

/* Trie::walk (call sites src/humid.cc:117,178,307): depth-first, children in
 * index order 0..3 => ascending lexicographic == ascending packed word. */
static void walk_leaves(const orc_ctx *c, void (*fn)(OLeaf *, const uint64_t *, void *), void *arg) {
  const ONode *stack_node[66];
  unsigned stack_next[66];
  unsigned __int128 path = 0;
  int depth = 0;
  stack_node[0] = c->root;
  stack_next[0] = 0;
  while (depth >= 0) {
    const ONode *node = stack_node[depth];
    if ((uint32_t)depth == c->n) {
      if (node->leaf) {
        uint64_t w[2];
        if (c->wpr == 1) { w[0] = (uint64_t)path; w[1] = 0; }
        else { w[0] = (uint64_t)(path >> 64); w[1] = (uint64_t)path; }
        fn(node->leaf, w, arg);
      }
      depth--;
      path >>= 2;
      continue;
    }
    unsigned i = stack_next[depth];
    while (i < 4 && !node->child[i]) i++;
    if (i == 4) {
      depth--;
      path >>= 2;
      continue;
    }
    stack_next[depth] = i + 1;
    path = (path << 2) | i;
    depth++;
    stack_node[depth] = node->child[i];
    stack_next[depth] = 0;
  }
}

/* Trie::asymmetricHamming (call site src/humid.cc:118-119).  Published algorithm
 * of jfjlaros/trie: descend all children; a mismatch costs 1; in asymmetric mode a
 * child below word[position] is skipped until the path has once gone above the
 * query (`full`), so every unordered pair is reported from its smaller side only,
 * results in ascending order, the query itself included (the caller drops it,
 * src/humid.cc:120). */
typedef struct {
  const orc_ctx *c;
  const uint64_t *word;
  OLeaf *from;
  uint64_t pairs;
} HamArg;

static void asym_hamming_(HamArg *a, const ONode *node, uint32_t position, int distance, int full) {
  if (distance < 0) return;
  if (position == a->c->n) {
    OLeaf *h = node->leaf;
    if (h && h != a->from) {          /* src/humid.cc:120 */
      leaf_push(a->from, h);          /* :121 */
      leaf_push(h, a->from);          /* :122 */
      a->pairs++;
    }
    return;
  }
  unsigned q = sym(a->word, a->c->n, position);
  for (unsigned i = 0; i < 4; i++) {
    const ONode *ch = node->child[i];
    if (ch && (full || i >= q))
      asym_hamming_(a, ch, position + 1, distance - (i != q), full || i > q);
  }
}

void orc_read_data(orc_ctx *c, const uint64_t *words, const uint8_t *filtered, uint64_t n_reads) {
  for (uint64_t r = 0; r < n_reads; r++) {       /* src/humid.cc:92-99 */
    if (!(filtered && filtered[r])) {
      trie_add(c, words + r * c->wpr);
      c->usable++;
    }
    c->total++;
  }
}

typedef struct { orc_ctx *c; uint32_t distance; size_t k; } NbArg;
static void collect_cb(OLeaf *l, const uint64_t *w, void *arg) {
  (void)w;
  NbArg *a = (NbArg *)arg;
  l->rank = (uint32_t)a->k;
  a->c->walk[a->k++] = l;
}
static void neighbours_cb(OLeaf *l, const uint64_t *w, void *arg) {
  NbArg *a = (NbArg *)arg;
  HamArg h = {a->c, w, l, 0};
  asym_hamming_(&h, a->c->root, 0, (int)a->distance, 0);
  a->c->edges += h.pairs;
}

/* src/humid.cc:113-130 findHammingNeighbours */
uint64_t orc_find_hamming_neighbours(orc_ctx *c, uint32_t distance) {
  free(c->walk);
  c->walk = (OLeaf **)xmalloc((size_t)c->unique * sizeof(OLeaf *));
  NbArg a = {c, distance, 0};
  walk_leaves(c, collect_cb, &a);     /* rank bookkeeping (not in the reference) */
  a.k = 0;
  walk_leaves(c, neighbours_cb, &a);
  return c->unique;
}

/* Trie::asymmetricLevenshtein (call site src/humid.cc:146-147).  lib/trie is absent (PARITY
 * UNPINNED, see humid_oracle.h): restated as the textbook Levenshtein search of a trie -- depth
 * first over the children in index order with one dynamic-programming row per node (edit distances
 * between the path so far and every prefix of the query), pruned when the whole row exceeds the
 * bound -- with hypothesis H3: "asymmetric" = every unordered pair is reported once, from its
 * smaller word, in ascending order (the Hamming variant's H2). */
typedef struct { orc_ctx *c; const uint64_t *word; OLeaf *from; uint64_t pairs; int distance; } LevArg;
static void asym_lev_(LevArg *a, const ONode *node, uint32_t depth, const int *prev) {
  const uint32_t n = a->c->n;
  if (depth == n) {
    OLeaf *h = node->leaf;
    if (h && h != a->from && h->rank > a->from->rank && prev[n] <= a->distance) {   /* src/humid.cc:148 */
      leaf_push(a->from, h);          /* :149 */
      leaf_push(h, a->from);          /* :150 */
      a->pairs++;
    }
    return;
  }
  for (unsigned i = 0; i < 4; i++) {
    const ONode *ch = node->child[i];
    if (!ch) continue;
    int row[66];
    int best;
    row[0] = (int)depth + 1;
    best = row[0];
    for (uint32_t j = 1; j <= n; j++) {
      const int sub = prev[j - 1] + (sym(a->word, n, j - 1) != i);
      const int del = prev[j] + 1, ins = row[j - 1] + 1;
      int v = sub < del ? sub : del;
      if (ins < v) v = ins;
      row[j] = v;
      if (v < best) best = v;
    }
    if (best <= a->distance) asym_lev_(a, ch, depth + 1, row);
  }
}

static void edit_neighbours_cb(OLeaf *l, const uint64_t *w, void *arg) {
  NbArg *a = (NbArg *)arg;
  LevArg h = {a->c, w, l, 0, (int)a->distance};
  int row0[66];
  for (uint32_t j = 0; j <= a->c->n; j++) row0[j] = (int)j;
  asym_lev_(&h, a->c->root, 0, row0);
  a->c->edges += h.pairs;
}

/* src/humid.cc:140-158 findEditNeighbours */
uint64_t orc_find_edit_neighbours(orc_ctx *c, uint32_t distance) {
  free(c->walk);
  c->walk = (OLeaf **)xmalloc((size_t)c->unique * sizeof(OLeaf *));
  NbArg a = {c, distance, 0};
  walk_leaves(c, collect_cb, &a);
  a.k = 0;
  walk_leaves(c, edit_neighbours_cb, &a);
  return c->unique;
}

typedef struct { orc_ctx *c; int maximum; size_t id; } ClArg;
static void clusters_cb(OLeaf *l, const uint64_t *w, void *arg) {
  (void)w;
  ClArg *a = (ClArg *)arg;
  orc_ctx *c = a->c;
  if (!l->cluster) {                                   /* src/humid.cc:179 */
    OCluster *cl = (OCluster *)xcalloc(1, sizeof(OCluster));
    cl->id = a->id++;                                  /* :180 */
    if (a->maximum) orc_assign_max_cluster(l, cl);     /* :182 */
    else orc_assign_directional_cluster(l, cl);        /* :185 */
    if (c->n_clusters == c->cap_clusters) {
      c->cap_clusters = c->cap_clusters ? 2 * c->cap_clusters : 1024;
      c->clusters = (OCluster **)xrealloc(c->clusters, c->cap_clusters * sizeof(OCluster *));
    }
    c->clusters[c->n_clusters++] = cl;                 /* :187 */
  }
}

/* src/humid.cc:167-193 findClusters; ids start at 1 (:177) */
uint64_t orc_find_clusters(orc_ctx *c, int maximum) {
  ClArg a = {c, maximum, 1};
  walk_leaves(c, clusters_cb, &a);
  return c->n_clusters;
}

/* src/humid.cc:220-234 (keep) and :268-285 (cluster_id), one pass over the reads */
void orc_map_reads(orc_ctx *c, const uint64_t *words, const uint8_t *filtered,
                   uint64_t n_reads, uint32_t *cluster_id, uint8_t *keep) {
  for (size_t i = 0; i < c->n_clusters; i++) c->clusters[i]->visited = 0;
  for (uint64_t r = 0; r < n_reads; r++) {
    uint32_t id = 0;            /* :272 cluster 0 = could not be clustered */
    uint8_t k = 0;
    if (!(filtered && filtered[r])) {
      OLeaf *leaf = trie_find(c, words + r * c->wpr);
      if (!leaf->cluster->visited && leaf->cluster->maxLeaf == leaf) {  /* :224-226 */
        k = 1;
        leaf->cluster->visited = 1;                                     /* :231 */
      }
      id = (uint32_t)leaf->cluster->id;                                 /* :277 */
    }
    if (cluster_id) cluster_id[r] = id;
    if (keep) keep[r] = k;
  }
}

uint64_t orc_total(const orc_ctx *c) { return c->total; }
uint64_t orc_usable(const orc_ctx *c) { return c->usable; }
uint64_t orc_unique(const orc_ctx *c) { return c->unique; }
uint64_t orc_n_clusters(const orc_ctx *c) { return c->n_clusters; }
uint64_t orc_n_edges(const orc_ctx *c) { return c->edges; }

typedef struct {
  uint64_t *word, *count; uint32_t *degree, *cluster_id; uint8_t *is_max; size_t k; uint32_t wpr;
} ExArg;
static void export_cb(OLeaf *l, const uint64_t *w, void *arg) {
  ExArg *a = (ExArg *)arg;
  size_t k = a->k++;
  if (a->word) { for (uint32_t q = 0; q < a->wpr; q++) a->word[k * a->wpr + q] = w[q]; }
  if (a->count) a->count[k] = l->count;
  if (a->degree) a->degree[k] = (uint32_t)l->nn;
  if (a->cluster_id) a->cluster_id[k] = l->cluster ? (uint32_t)l->cluster->id : 0;
  if (a->is_max) a->is_max[k] = (l->cluster && l->cluster->maxLeaf == l) ? 1 : 0;
}
void orc_export_leaves(const orc_ctx *c, uint64_t *word, uint64_t *count,
                       uint32_t *degree, uint32_t *cluster_id, uint8_t *is_max_leaf) {
  ExArg a = {word, count, degree, cluster_id, is_max_leaf, 0, c->wpr};
  walk_leaves(c, export_cb, &a);
}

void orc_export_adjacency(const orc_ctx *c, uint64_t *nbr_off, uint32_t *nbr_idx) {
  uint64_t off = 0;
  for (uint64_t u = 0; u < c->unique; u++) {
    OLeaf *l = c->walk[u];
    nbr_off[u] = off;
    for (size_t j = 0; j < l->nn; j++) nbr_idx[off + j] = l->nbr[j]->rank;
    off += l->nn;
  }
  nbr_off[c->unique] = off;
}

void orc_export_clusters(const orc_ctx *c, uint64_t *size, uint64_t *max_count,
                         uint32_t *max_leaf_rank) {
  for (size_t i = 0; i < c->n_clusters; i++) {
    OCluster *cl = c->clusters[i];
    if (size) size[cl->id - 1] = cl->size;
    if (max_count) max_count[cl->id - 1] = cl->maxCount;
    if (max_leaf_rank) max_leaf_rank[cl->id - 1] = cl->maxLeaf ? cl->maxLeaf->rank : 0xffffffffu;
  }
}

int orc_dedup_run(const uint64_t *words, const uint8_t *filtered, uint64_t n_reads,
                  uint32_t word_nt, uint32_t distance, uint32_t method,
                  uint32_t *cluster_id, uint8_t *keep, uint64_t *summary5,
                  double *phase_seconds) {
  orc_ctx *c = orc_create(word_nt);
  if (!c) return -1;
  double t0 = now_s();
  orc_read_data(c, words, filtered, n_reads);
  double t1 = now_s();
  if (method & 2) orc_find_edit_neighbours(c, distance);       /* -e, src/humid.cc:381-382 */
  else orc_find_hamming_neighbours(c, distance);
  double t2 = now_s();
  orc_find_clusters(c, (method & 1) != 0);
  double t3 = now_s();
  orc_map_reads(c, words, filtered, n_reads, cluster_id, keep);
  double t4 = now_s();
  if (summary5) {
    summary5[0] = c->total;
    summary5[1] = c->usable;
    summary5[2] = c->unique;
    summary5[3] = c->n_clusters;
    summary5[4] = orc_n_edges(c);
  }
  if (phase_seconds) {
    phase_seconds[0] = t1 - t0;
    phase_seconds[1] = t2 - t1;
    phase_seconds[2] = t3 - t2;
    phase_seconds[3] = t4 - t3;
  }
  orc_destroy(c);
  return 0;
}
