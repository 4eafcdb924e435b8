// kernels_graph.hip.h -- neighbour search (generalised pigeonhole), CSR, union-find components
// Part of libhumid_hip.so (see humid_hip.hip for the pipeline and the C ABI).  Device code for
// gfx950 only; included once, in this order, by humid_hip.hip.
#ifndef HUMID_KERNELS_GRAPH_HIP_H
#define HUMID_KERNELS_GRAPH_HIP_H

#include "common.hip.h"

// --------------------------------------------------------------------------------
// 3. neighbour search: pigeonhole segments
// --------------------------------------------------------------------------------
// Generalised pigeonhole: the n nucleotides are cut into s segments; two words within
// Hamming distance d agree exactly on at least s-d of them, so every pair is found in the bucket
// of some COMBINATION of s-d segments.  d=1: s=2, 2 combos of 12 nt (n=24).  d=2: s=4, 6 combos
// of 12 nt -- not 3 segments of 8 nt, whose 65 536 buckets hold hundreds of words each.
// Combo 0 is always the top s-d segments, i.e. a prefix: its buckets are runs of the sorted
// unique array and need no sort.  mask[c] = bits of combo c; a pair is emitted from the FIRST
// combo it agrees on.
#define MAX_COMBOS 20
#define MAX_FIELDS 8
struct ComboPlan {
  u32 ncombo;
  u32 key_bits;                       // bits of a combo key (sum of its field widths)
  W2 mask[MAX_COMBOS];                // hi = 0 for one-word plans; at most 64 bits set (= the key)
  u8 nfield[MAX_COMBOS];
  u8 shift[MAX_COMBOS][MAX_FIELDS];   // fields from most to least significant
  u8 width[MAX_COMBOS][MAX_FIELDS];
};

// bucket key of combo `cb` for every unique word (fields concatenated, most significant first)
// Kernel arguments derived from the plan are passed BY VALUE in small structs and indexed
// STATICALLY (unrolled loops with a predicate): they then live in scalar registers.  History
// (DESIGN.md section 3a): with the whole plan passed as one struct and its u8 field tables indexed
// dynamically, hipcc emitted vector byte loads and kept the loaded shift in the LAST allocated VGPR
// (v7 of 8) of k_combo_keys -- the one register that this platform occasionally overwrites with the
// lane number (see HUMID_GUARD_LAST_VGPR in common.hip.h): single waves computed wrong bucket keys
// and 5-60 of 219 k neighbour pairs were lost per 10 M-read run.  Both defences are kept: no plan
// value sits in a vector register, and no kernel's last register holds anything.
template <class WT>
struct EarlierMasksT {
  WT m[MAX_COMBOS];
};

// fields of ONE combo
struct ComboFields {
  u32 nf;
  u8 shift[MAX_FIELDS];
  u8 width[MAX_FIELDS];
};

template <class KeyT, class WT>
__global__ void k_combo_keys(const WT *__restrict__ s_word, u32 n, ComboFields cf,
                             KeyT *__restrict__ key, u32 *__restrict__ val) {
  HUMID_GUARD_LAST_VGPR();
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const WT w = s_word[i];
  u64 k = 0;
#pragma unroll
  for (u32 f = 0; f < MAX_FIELDS; f++) {
    if (f < cf.nf) {
      const u32 wd = cf.width[f];
      k = ((wd >= 64) ? 0ull : (k << wd)) | w_field(w, cf.shift[f], wd);
    }
  }
  key[i] = (KeyT)k;
  val[i] = i;
}

// Source of the two-level grouping (kernels_part.hip.h: tile partition + k_group_fine) for ANY combination
// of one-word words: payload = the word, key = its combination key (fields concatenated, first field most
// significant) moved to the top of 64 bits.
struct FieldsSrc {
  const u64 *words;
  ComboFields cf;
  u32 kb;                          // key bits, 1 .. 63
  __device__ __forceinline__ bool load(u32 j, u64 &payload) const { payload = words[j]; return true; }
  __device__ __forceinline__ u64 key(u64 w) const {
    u64 k = 0;
#pragma unroll
    for (u32 f = 0; f < MAX_FIELDS; f++) {
      if (f < cf.nf) {
        const u32 wd = cf.width[f];
        k = ((wd >= 64) ? 0ull : (k << wd)) | w_field(w, cf.shift[f], wd);
      }
    }
    return k << (64 - kb);
  }
};

// The same for two-word words: a 16-byte word does not fit the 8-byte payload, so the payload IS the key
// (computed once from the word); the grouped payloads are scratch and the words are gathered through the
// grouped positions afterwards (k_gather_bucket_words).
struct FieldsSrcW2 {
  const W2 *words;
  ComboFields cf;
  u32 kb;
  __device__ __forceinline__ bool load(u32 j, u64 &payload) const {
    const W2 w = words[j];
    u64 k = 0;
#pragma unroll
    for (u32 f = 0; f < MAX_FIELDS; f++) {
      if (f < cf.nf) {
        const u32 wd = cf.width[f];
        k = ((wd >= 64) ? 0ull : (k << wd)) | w_field(w, cf.shift[f], wd);
      }
    }
    payload = k << (64 - kb);
    return true;
  }
  __device__ __forceinline__ u64 key(u64 payload) const { return payload; }
};

// --------------------------------------------------------------------------------
// 4. connected components (lock-free union-find, smaller index wins => root = min rank)
// --------------------------------------------------------------------------------
__device__ __forceinline__ u32 uf_find(const u32 *P, u32 x) {
  u32 p = ld_agent(&P[x]);
  while (p != x) { x = p; p = ld_agent(&P[x]); }
  return x;
}

__device__ __forceinline__ void uf_union(u32 *P, u32 a, u32 b) {
  while (true) {
    a = uf_find(P, a);
    b = uf_find(P, b);
    if (a == b) return;
    if (a > b) { u32 t = a; a = b; b = t; }
    if (atomicCAS(&P[b], b, a) == b) return;
  }
}

// One thread per position i of the bucket-sorted order (V = ranks in that order; combo 0 uses
// the sorted unique array itself); compares with the following elements of its bucket: the
// bucket ends at the first j whose word differs inside the combo mask.  Ranks ascend inside a
// bucket, so (ri < rj) always.  A pair is emitted only from the FIRST combo it agrees on.
// Two phases with identical control flow and no shared append counter:
//   FILL = false: deg[] += 1 per endpoint, union(ri, rj) in the component forest
//   FILL = true : writes rj into ri's CSR row and ri into rj's (per-row cursors; the rows are
//                 put in ascending order afterwards by k_sort_lists)
// MODE: what happens to a found pair
//   PM_COUNT      deg[] += 1 per endpoint, union(ri, rj)            (single-GPU phase A)
//   PM_FILL       both directions into the CSR rows via cursors      (single-GPU phase B)
//   PM_EMIT_COUNT pc[t] = pairs found by this thread                 (multi-GPU share, phase A)
//   PM_EMIT_FILL  edge (min << 32 | max) at poff[t] + k              (multi-GPU share, phase B)
// i0/n_i: the thread block covers positions [i0, i0 + n_i) as the first element of a pair; the
// second runs on to the end of the bucket anywhere in [0, n).
enum { PM_COUNT = 0, PM_FILL = 1, PM_EMIT_COUNT = 2, PM_EMIT_FILL = 3 };

// The forest of `parent` only has to keep together what the clustering can move between.  The
// directional method climbs to and floods along neighbours whose counts differ by a factor two
// (src/cluster.cc:39-69: every step tests atLeastDouble_); a neighbour pair with similar counts is
// never crossed, so it need not join two components.  cnt == nullptr (maximum method, which floods
// every edge, src/cluster.cc:72-80): every pair joins.  Finer components = more of them in flight
// and no order to respect between them: 10^6 single-read words that are all neighbours of
// neighbours are 10^6 independent one-leaf components instead of one chain walked leaf by leaf.
__device__ __forceinline__ bool joins_for_clustering(const u32 *__restrict__ cnt, u32 a, u32 b) {
  if (!cnt) return true;
  const u64 ca = cnt[a], cb = cnt[b];
  return at_least_double(ca, cb) || at_least_double(cb, ca);
}

template <bool PASS0, int MODE, class WT>
__global__ void __launch_bounds__(256)
k_pairs(const WT *__restrict__ W, const u32 *__restrict__ V, u32 n, u32 i0, u32 n_i, WT mask,
        EarlierMasksT<WT> em, u32 cb, u32 distance, u32 *deg, u32 *parent,
        const u32 *__restrict__ nbr_off, u32 *cur, u32 *nbr_idx, u32 *__restrict__ pc,
        const u32 *__restrict__ poff, u64 *__restrict__ edges, u32 *__restrict__ had,
        u32 walk_max = 0, ull *big = nullptr, const u32 *__restrict__ cnt = nullptr) {
  HUMID_GUARD_LAST_VGPR();
  // W: the words IN THE ORDER WALKED (the sorted unique array for the prefix combo, a gathered
  // copy in bucket order for the sorted combos), so the inner loop is one sequential, coalesced
  // stream; the ranks V[] are only loaded for the pairs that are found.
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_i) return;
  // second phase: only the few positions that found a pair in the first walk their bucket again
  // (had[] / pc[] were written by the matching first-phase launch: same grid, same t)
  // had[t] = pairs found (capped, << 24) | distance to the first one: a position with ONE pair -- nearly all
  // of them -- writes its two CSR entries without walking its bucket a second time
  const u32 h = (MODE == PM_FILL && had) ? had[t] : 0u;
  if (MODE == PM_FILL && had && !h) return;
  if (MODE == PM_EMIT_FILL && pc[t] == 0) return;
  const u32 i = i0 + t;
  if (MODE == PM_FILL && (h >> 24) == 1u && (h & 0xffffffu)) {
    const u32 j = i + (h & 0xffffffu);
    const u32 ri = PASS0 ? i : V[i], rj = PASS0 ? j : V[j];
    nbr_idx[nbr_off[ri] + atomicAdd(&cur[ri], 1u)] = rj;
    nbr_idx[nbr_off[rj] + atomicAdd(&cur[rj], 1u)] = ri;
    return;
  }
  const WT wi = W[i];
  const u32 ri = PASS0 ? i : V[i];
  u32 found = 0, first_off = 0;
  u64 e = (MODE == PM_EMIT_FILL) ? (u64)poff[t] : 0;
  // walk_max > 0: a position compares itself with at most the next walk_max words of its bucket;
  // a bucket that goes on beyond that is reported in *big (bit cb) and its remaining pairs
  // (j - i > walk_max) belong to k_pairs_tiles -- a thread per position is quadratic in a lane
  // on a bucket of 10^5 words, the tiles spread the same comparisons over the whole device
  const u32 jend = (walk_max && n - i > walk_max + 1) ? i + walk_max + 1 : n;
  u32 j = i + 1;
  for (; j < jend; j++) {
    const WT x = w_xor(wi, W[j]);
    if (w_hits(x, mask)) break;                    // left the bucket
    if (w_mismatch(x) > distance) continue;
    bool first = true;
#pragma unroll
    for (u32 q = 0; q < MAX_COMBOS; q++)
      first = first && !(q < cb && !w_hits(x, em.m[q]));
    if (!first) continue;
    const u32 rj = PASS0 ? j : V[j];
    if (MODE == PM_FILL) {
      nbr_idx[nbr_off[ri] + atomicAdd(&cur[ri], 1u)] = rj;
      nbr_idx[nbr_off[rj] + atomicAdd(&cur[rj], 1u)] = ri;
    } else if (MODE == PM_COUNT) {
      if (!found) first_off = j - i;
      found++;
      atomicAdd(&deg[rj], 1u);
      if (joins_for_clustering(cnt, ri, rj)) uf_union(parent, ri, rj);
    } else if (MODE == PM_EMIT_COUNT) {
      found++;
    } else {
      edges[e++] = ri < rj ? (((u64)ri << 32) | rj) : (((u64)rj << 32) | ri);
    }
  }
  if (MODE == PM_COUNT && found) atomicAdd(&deg[ri], found);
  if (MODE == PM_COUNT && had) had[t] = found ? (((found < 255u ? found : 255u) << 24) | (first_off < (1u << 24) ? first_off : 0u)) : 0u;
  if (MODE == PM_EMIT_COUNT) pc[t] = found;
  if ((MODE == PM_COUNT || MODE == PM_EMIT_COUNT) && big && j == jend && jend < n && !w_hits(w_xor(wi, W[jend]), mask))
    atomicOr(big, 1ull << cb);
}

// ---- large buckets: the pairs k_pairs leaves out, as tiles ------------------------------------
// A run is a bucket (maximal stretch of equal key in the walked order) longer than walk_max + 1.
// Its pairs (i, j) with j - i > walk_max are cut into PT2_TILE x PT2_TILE squares of the upper
// triangle; one workgroup compares one square: the 'b' side staged in LDS and read by every lane
// at the same address (a broadcast, no bank conflict), PT2_ROWS words of the 'a' side per thread in
// registers, so one LDS read feeds PT2_ROWS xor/popcount comparisons.  Found pairs are as rare
// here as anywhere and take the same actions as in k_pairs.
#define PT2_THREADS 256
#define PT2_ROWS 4
#define PT2_TILE (PT2_THREADS * PT2_ROWS)      // 1024 words a side; also the walk_max of k_pairs
struct BigRun { u32 start, len; ull tile0; };  // tile0: running sum of nt (nt + 1) / 2 over the runs before

// heads of the runs: position i starts a bucket and the bucket still holds position i + walk_max + 1.
// The end is found by bisection: inside the walked order equal keys are contiguous.
template <class WT>
__global__ void k_big_runs(const WT *__restrict__ W, u32 n, WT mask, u32 walk_max, BigRun *runs, u32 cap,
                           u32 *n_runs) {
  HUMID_GUARD_LAST_VGPR();
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || n - i <= walk_max + 1) return;
  const WT wi = W[i];
  if (w_hits(w_xor(wi, W[i + walk_max + 1]), mask)) return;
  if (i > 0 && !w_hits(w_xor(wi, W[i - 1]), mask)) return;       // not the first of its bucket
  u32 lo = i + walk_max + 1, hi = n;                              // W[lo] in the bucket, W[hi] not (or hi == n)
  while (hi - lo > 1) {
    const u32 mid = lo + (hi - lo) / 2;
    if (w_hits(w_xor(wi, W[mid]), mask)) hi = mid; else lo = mid;
  }
  const u32 k = atomicAdd(n_runs, 1u);
  if (k < cap) { runs[k].start = i; runs[k].len = hi - i; runs[k].tile0 = 0; }
}

template <bool PASS0, int MODE, class WT>
__global__ void __launch_bounds__(PT2_THREADS)
k_pairs_tiles(const WT *__restrict__ W, const u32 *__restrict__ V, const BigRun *__restrict__ runs, u32 n_runs,
              ull total_tiles, EarlierMasksT<WT> em, u32 cb, u32 distance, u32 walk_max, u32 *deg, u32 *parent,
              const u32 *__restrict__ nbr_off, u32 *cur, u32 *nbr_idx, const u32 *__restrict__ cnt,
              u64 *__restrict__ edges = nullptr, ull *ecount = nullptr, u32 i_lo = 0, u32 i_hi = 0xffffffffu) {
  HUMID_GUARD_LAST_VGPR();
  // [i_lo, i_hi): only pairs whose FIRST position lies in that range (a rank's share of the positions in the
  // all-gather mode's pair search); the default is every pair
  // PM_EMIT_COUNT: *ecount += pairs found (one add per wave and square); PM_EMIT_FILL: the pairs are
  // appended to edges[] at *ecount (which the caller set to the pairs already there)
  __shared__ WT sb[PT2_TILE];
  u32 emitted = 0;
  for (ull q = blockIdx.x; q < total_tiles; q += gridDim.x) {
    u32 lo = 0, hi = n_runs;                                      // the run whose tiles hold q
    while (hi - lo > 1) {
      const u32 mid = (lo + hi) / 2;
      if (runs[mid].tile0 <= q) lo = mid; else hi = mid;
    }
    const u32 s = runs[lo].start, L = runs[lo].len;
    const ull r = q - runs[lo].tile0;                             // r = b (b + 1) / 2 + a, a <= b
    u32 b = (u32)((sqrt(8.0 * (double)r + 1.0) - 1.0) * 0.5);
    while ((ull)b * (b + 1) / 2 > r) b--;
    while ((ull)(b + 1) * (b + 2) / 2 <= r) b++;
    const u32 a = (u32)(r - (ull)b * (b + 1) / 2);
    if ((b - a) * PT2_TILE + PT2_TILE - 1 <= walk_max) continue;  // every pair of the square is k_pairs'
    const u32 b0 = b * PT2_TILE, nb = L - b0 < PT2_TILE ? L - b0 : PT2_TILE;
    __syncthreads();                                              // the previous square is done with sb
    for (u32 u = threadIdx.x; u < nb; u += PT2_THREADS) sb[u] = W[s + b0 + u];
    WT wa[PT2_ROWS];
    u32 lim[PT2_ROWS];                                            // j > lim: the pair is this kernel's
#pragma unroll
    for (u32 k = 0; k < PT2_ROWS; k++) {
      const u32 ia = a * PT2_TILE + k * PT2_THREADS + threadIdx.x;
      const bool ok = ia < L;
      wa[k] = W[s + (ok ? ia : 0)];
      lim[k] = ok ? s + ia + walk_max : 0xffffffffu;
    }
    __syncthreads();
    for (u32 u = 0; u < nb; u++) {
      const WT wb = sb[u];
      const u32 j = s + b0 + u;
#pragma unroll
      for (u32 k = 0; k < PT2_ROWS; k++) {
        const WT x = w_xor(wa[k], wb);
        if (w_mismatch(x) > distance || j <= lim[k]) continue;
        bool first = true;
#pragma unroll
        for (u32 qq = 0; qq < MAX_COMBOS; qq++)
          first = first && !(qq < cb && !w_hits(x, em.m[qq]));
        if (!first) continue;
        const u32 i = lim[k] - walk_max;
        if (i < i_lo || i >= i_hi) continue;
        const u32 ri = PASS0 ? i : V[i], rj = PASS0 ? j : V[j];
        if (MODE == PM_FILL) {
          nbr_idx[nbr_off[ri] + atomicAdd(&cur[ri], 1u)] = rj;
          nbr_idx[nbr_off[rj] + atomicAdd(&cur[rj], 1u)] = ri;
        } else if (MODE == PM_COUNT) {
          atomicAdd(&deg[ri], 1u);
          atomicAdd(&deg[rj], 1u);
          if (joins_for_clustering(cnt, ri, rj)) uf_union(parent, ri, rj);
        } else if (MODE == PM_EMIT_COUNT) {
          emitted++;
        } else {
          edges[atomicAdd(ecount, 1ull)] = ri < rj ? (((u64)ri << 32) | rj) : (((u64)rj << 32) | ri);
        }
      }
    }
  }
  if (MODE == PM_EMIT_COUNT) {
#pragma unroll
    for (u32 dd = 32; dd >= 1; dd >>= 1) emitted += __shfl_xor(emitted, dd);
    if ((threadIdx.x & 63) == 0 && emitted) atomicAdd(ecount, (ull)emitted);
  }
}

// words of a sorted combo in bucket order (one gather per combo instead of one per comparison)
template <class WT>
__global__ void k_gather_bucket_words(const WT *__restrict__ s_word, const u32 *__restrict__ V, u32 n,
                                      WT *__restrict__ wv, const u32 *__restrict__ n_valid = nullptr) {
  HUMID_GUARD_LAST_VGPR();
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (n_valid && *n_valid < n) n = *n_valid;           // (a padded grouping that dropped words: V ends there)
  if (i < n) wv[i] = s_word[V[i]];
}

// the same two phases driven by an explicit edge list (multi-GPU: the ranks' shares, all-gathered)
template <bool FILL>
__global__ void __launch_bounds__(256)
k_edges_apply(const u64 *__restrict__ edges, u64 n_edges, u32 n_nodes, u32 *deg, u32 *parent,
              const u32 *__restrict__ nbr_off, u32 *cur, u32 *nbr_idx, ull *ctr,
              const u32 *__restrict__ cnt = nullptr) {
  HUMID_GUARD_LAST_VGPR();
  for (u64 k = (u64)blockIdx.x * blockDim.x + threadIdx.x; k < n_edges; k += (u64)gridDim.x * blockDim.x) {
    const u64 ed = edges[k];
    const u32 a = (u32)(ed >> 32), b = (u32)ed;
    if (a >= n_nodes || b >= n_nodes || a == b) { ctr[CTR_OVERFULL] = 1; continue; }   // malformed edge
    if (FILL) {
      nbr_idx[nbr_off[a] + atomicAdd(&cur[a], 1u)] = b;
      nbr_idx[nbr_off[b] + atomicAdd(&cur[b], 1u)] = a;
    } else {
      atomicAdd(&deg[a], 1u);
      atomicAdd(&deg[b], 1u);
      if (joins_for_clustering(cnt, a, b)) uf_union(parent, a, b);
    }
  }
}

// multi-GPU share of a sorted combo: the unique words whose combo key lies in [klo, khi]
// (key, rank) appended in arbitrary order; fixed grid, one global atomic per block
template <class KeyT>
__global__ void __launch_bounds__(256)
k_select_keyrange(const u64 *__restrict__ s_word, u32 n, ComboFields cf, u64 klo, u64 khi,
                  KeyT *__restrict__ key_out, u32 *__restrict__ val_out, ull *ctr) {
  HUMID_GUARD_LAST_VGPR();
  __shared__ u32 lds[8];
  const u32 chunk = (n + gridDim.x - 1) / gridDim.x;
  const u32 lo = blockIdx.x * chunk;
  const u32 hi = (lo + chunk < n) ? lo + chunk : n;
  auto key_of = [&](u32 i) {
    const u64 w = s_word[i];
    u64 k = 0;
#pragma unroll
    for (u32 f = 0; f < MAX_FIELDS; f++) {
      if (f < cf.nf) {
        const u32 wd = cf.width[f];
        k = ((wd >= 64) ? 0ull : (k << wd)) | w_field(w, cf.shift[f], wd);
      }
    }
    return k;
  };
  u32 mine = 0;
  for (u32 i = lo + threadIdx.x; i < hi; i += 256) {
    const u64 k = key_of(i);
    mine += (k >= klo && k <= khi) ? 1u : 0u;
  }
  const u32 total = block_sum(mine, lds);
  if (threadIdx.x == 0) lds[4] = total ? (u32)atomicAdd(&ctr[CTR_SPECIAL], (ull)total) : 0u;
  __syncthreads();
  u32 base = lds[4];
  if (total == 0) return;
  for (u32 i0 = lo; i0 < hi; i0 += 256) {
    const u32 i = i0 + threadIdx.x;
    u64 k = 0;
    bool sel = false;
    if (i < hi) { k = key_of(i); sel = (k >= klo && k <= khi); }
    u32 tot;
    const u32 r = block_rank(sel, lds, &tot);
    if (sel) { key_out[base + r] = (KeyT)k; val_out[base + r] = i; }
    base += tot;
  }
}

// --------------------------------------------------------------------------------
// 3b. edit distance (-e, /root/reference/src/humid.cc:140-158)
// --------------------------------------------------------------------------------
// Levenshtein distance between two n-nucleotide words, exact up to 3 (returns >= 4 otherwise;
// lev_band2 below: exact up to 5).
// Equal lengths: d edits hold at most d/2 insertions and as many deletions, so for d <= 3 an
// optimal alignment never leaves the diagonals -1, 0, +1.  Row i keeps D[i][i-1], D[i][i],
// D[i][i+1] (L, M, R).
template <class WT>
__device__ __forceinline__ u32 lev_band1(WT x, WT y, u32 n) {
  const u32 INF = 64;
  u32 L = INF, M = 0, R = 1;
  u32 y0 = 4u, y1 = w_sym(y, n, 0), y2 = n >= 2 ? w_sym(y, n, 1) : 4u;    // y_i, y_{i+1}, y_{i+2}
  for (u32 i = 0; i < n; i++) {                          // row i -> row i + 1
    const u32 xs = w_sym(x, n, i);                       // x_{i+1}
    const u32 nl = min(L + (xs != y0), M + 1u);
    const u32 nm = min(min(M + (xs != y1), R + 1u), nl + 1u);
    const u32 nr = i + 2 <= n ? min(R + (xs != y2), nm + 1u) : INF;
    L = nl; M = nm; R = nr;
    y0 = y1;
    y1 = y2;
    y2 = i + 3 <= n ? w_sym(y, n, i + 2) : 4u;           // y_{(i+1)+2}
  }
  return M;
}

// The same with FIVE diagonals (-2 .. +2): exact up to distance 5 (at most two insertions and two
// deletions between equal-length words).  v[t] = D[i][i + t - 2]; cells outside the matrix are INF.
template <class WT>
__device__ __forceinline__ u32 lev_band2(WT x, WT y, u32 n) {
  const u32 INF = 64;
  u32 v[5] = {INF, INF, 0, 1, 2};                        // row 0: D[0][j] = j
  if (n < 1) v[3] = INF;
  if (n < 2) v[4] = INF;
  for (u32 i = 0; i < n; i++) {                          // row i -> row i + 1
    const u32 xs = w_sym(x, n, i);                       // x_{i+1}
    u32 nv[5];
#pragma unroll
    for (u32 t = 0; t < 5; t++) {
      const int j = (int)i + 1 + (int)t - 2;             // column of the new cell, 0 .. n
      u32 best = INF;
      if (j >= 0 && j <= (int)n) {
        if (j >= 1) best = v[t] + (xs != w_sym(y, n, (u32)j - 1) ? 1u : 0u);      // diagonal
        if (t + 1 < 5) best = min(best, v[t + 1] + 1u);                            // x_{i+1} deleted
        if (t >= 1 && j >= 1) best = min(best, nv[t - 1] + 1u);                    // y_j inserted
      }
      nv[t] = best < INF ? best : INF;
    }
#pragma unroll
    for (u32 t = 0; t < 5; t++) v[t] = nv[t];
  }
  return v[2];
}

template <u32 BAND, class WT>
__device__ __forceinline__ u32 lev_band(WT x, WT y, u32 n) {
  return BAND == 1 ? lev_band1(x, y, n) : lev_band2(x, y, n);
}

// The x side of a verification, prepared once per thread (a thread verifies ONE word against a run of candidates).
// BAND 1, 2: the banded programmes above (d <= 3, d <= 5).  BAND 0: ANY distance (round 3: -e beyond 5 edits) --
// the whole dynamic programme of the n x n matrix, one COLUMN per step kept as the bit vectors of its vertical
// differences (Myers 1999 in Hyyro's edit-distance form: the row above the matrix grows by one per column).  n <= 64:
// one 64-bit vector; peq[s] = the positions of x that hold nucleotide s.
template <u32 BAND, class WT>
struct LevX {
  WT x;
  u32 n;
  __device__ __forceinline__ LevX(WT x_, u32 n_) : x(x_), n(n_) {}
  __device__ __forceinline__ u32 dist(WT y) const { return lev_band<BAND>(x, y, n); }
};
template <class WT>
struct LevX<0, WT> {
  u64 peq0, peq1, peq2, peq3;
  u32 n;
  __device__ __forceinline__ LevX(WT x, u32 n_) : peq0(0), peq1(0), peq2(0), peq3(0), n(n_) {
    for (u32 i = 0; i < n; i++) {
      const u32 sx = w_sym(x, n, i);
      const u64 b = 1ull << i;
      peq0 |= sx == 0 ? b : 0ull;
      peq1 |= sx == 1 ? b : 0ull;
      peq2 |= sx == 2 ? b : 0ull;
      peq3 |= sx == 3 ? b : 0ull;
    }
  }
  __device__ __forceinline__ u32 dist(WT y) const {
    u64 pv = ~0ull, mv = 0;
    u32 score = n;                                       // D[n][0]
    const u64 last = 1ull << (n - 1);
    for (u32 j = 0; j < n; j++) {                        // column j -> j + 1
      const u32 sy = w_sym(y, n, j);
      const u64 eq = (sy & 2u) ? ((sy & 1u) ? peq3 : peq2) : ((sy & 1u) ? peq1 : peq0);
      const u64 xv = eq | mv;
      const u64 xh = (((eq & pv) + pv) ^ pv) | eq;
      u64 ph = mv | ~(xh | pv);
      u64 mh = pv & xh;
      score += (ph & last) ? 1u : 0u;
      score -= (mh & last) ? 1u : 0u;
      ph = (ph << 1) | 1ull;                             // D[0][j + 1] = D[0][j] + 1
      mh <<= 1;
      pv = mh | ~(xv | ph);
      mv = ph & xv;
    }
    return score;                                        // D[n][n]
  }
};

// Candidate join of one combination and one shift pattern: X = the words' own segments (sorted by
// key), Y = the same segments read at shifted positions (sorted by key).  Thread per X entry: the
// run of equal keys in Y is found by binary search, every candidate is verified by the dynamic
// programme.  COUNT: pc[t] = pairs found; FILL: (smaller rank << 32 | larger rank) from poff[t].
// A pair may come out several times (both roles, several combinations): the list is made unique
// afterwards.
// walk (0: no bound) / big: a run of equal keys in Y longer than `walk` is not walked by one lane -- the COUNT pass
// stops there and raises *big; the caller then takes this join through k_edit_chunks / k_edit_join_chunks, which
// cut every long run into pieces of `walk` candidates (round 3: a 10^5-word bucket made every one of its 10^5 lanes
// verify 10^5 candidates)
template <bool FILL, class KeyT, class WT, u32 BAND>
__global__ void __launch_bounds__(256)
k_edit_join(const KeyT *__restrict__ KX, const u32 *__restrict__ VX, const KeyT *__restrict__ KY,
            const u32 *__restrict__ VY, u32 n, const WT *__restrict__ words, u32 word_nt, u32 distance,
            u32 *__restrict__ pc, const u32 *__restrict__ poff, u64 *__restrict__ edges, u32 walk = 0, ull *big = nullptr) {
  HUMID_GUARD_LAST_VGPR();
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const KeyT key = KX[t];
  const u32 rx = VX[t];
  const WT wx = words[rx];
  u32 lo = 0, hi = n;
  while (lo < hi) {
    const u32 mid = lo + ((hi - lo) >> 1);
    if (KY[mid] < key) lo = mid + 1; else hi = mid;
  }
  u32 found = 0;
  u64 e = FILL ? (u64)poff[t] : 0;
  if (!FILL && walk && big && lo + walk < n && KY[lo + walk] == key) { atomicOr(big, 1ull); pc[t] = 0; return; }
  const LevX<BAND, WT> lx(wx, word_nt);
  for (u32 j = lo; j < n && KY[j] == key; j++) {
    const u32 ry = VY[j];
    if (ry == rx) continue;
    if (lx.dist(words[ry]) > distance) continue;
    if (FILL) edges[e++] = rx < ry ? (((u64)rx << 32) | ry) : (((u64)ry << 32) | rx);
    else found++;
  }
  if (!FILL) pc[t] = found;
}

// ---- the same join with every run of equal keys cut into pieces of `walk` candidates ----
// per X entry: start of its run in Y and the number of pieces (>= 1: the scan's input)
template <class KeyT>
__global__ void __launch_bounds__(256)
k_edit_chunks(const KeyT *__restrict__ KX, const KeyT *__restrict__ KY, u32 n, u32 walk, u32 *__restrict__ run_lo,
              u32 *__restrict__ n_chunks) {
  HUMID_GUARD_LAST_VGPR();
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t > n) return;
  if (t == n) { n_chunks[n] = 0; return; }             // scan sentinel
  const KeyT key = KX[t];
  u32 lo = 0, hi = n;
  while (lo < hi) {
    const u32 mid = lo + ((hi - lo) >> 1);
    if (KY[mid] < key) lo = mid + 1; else hi = mid;
  }
  const u32 first = lo;
  hi = n;
  while (lo < hi) {                                    // first j with KY[j] > key
    const u32 mid = lo + ((hi - lo) >> 1);
    if (KY[mid] <= key) lo = mid + 1; else hi = mid;
  }
  const u32 len = lo - first;
  run_lo[t] = first;
  n_chunks[t] = len ? (len + walk - 1) / walk : 1u;
}
// one thread per piece: chunk_off = exclusive scan of n_chunks (n + 1 entries)
template <bool FILL, class KeyT, class WT, u32 BAND>
__global__ void __launch_bounds__(256)
k_edit_join_chunks(const KeyT *__restrict__ KX, const u32 *__restrict__ VX, const KeyT *__restrict__ KY,
                   const u32 *__restrict__ VY, u32 n, const u32 *__restrict__ run_lo, const u32 *__restrict__ chunk_off,
                   u32 n_pieces, u32 walk, const WT *__restrict__ words, u32 word_nt, u32 distance, u32 *__restrict__ pc,
                   const u32 *__restrict__ poff, u64 *__restrict__ edges) {
  HUMID_GUARD_LAST_VGPR();
  const u32 p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_pieces) return;
  u32 lo = 0, hi = n;                                  // the X entry of this piece: largest t with chunk_off[t] <= p
  while (hi - lo > 1) {
    const u32 mid = lo + ((hi - lo) >> 1);
    if (chunk_off[mid] <= p) lo = mid; else hi = mid;
  }
  const u32 t = lo;
  const KeyT key = KX[t];
  const u32 rx = VX[t];
  const WT wx = words[rx];
  const u32 j0 = run_lo[t] + (p - chunk_off[t]) * walk;
  u32 found = 0;
  u64 e = FILL ? (u64)poff[p] : 0;
  const LevX<BAND, WT> lx(wx, word_nt);
  for (u32 j = j0; j < n && j - j0 < walk && KY[j] == key; j++) {
    const u32 ry = VY[j];
    if (ry == rx) continue;
    if (lx.dist(words[ry]) > distance) continue;
    if (FILL) edges[e++] = rx < ry ? (((u64)rx << 32) | ry) : (((u64)ry << 32) | rx);
    else found++;
  }
  if (!FILL) pc[p] = found;
}

// head[i] = 1 where a new value starts in the sorted 64-bit array; head[n] = 0 (scan sentinel)
static __global__ void k_heads_u64(const u64 *__restrict__ sorted, u32 n, u32 *__restrict__ head) {
  HUMID_GUARD_LAST_VGPR();
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i > n) return;
  head[i] = (i < n && (i == 0 || sorted[i] != sorted[i - 1])) ? 1u : 0u;
}

static __global__ void k_compact_heads_u64(const u64 *__restrict__ sorted, const u32 *__restrict__ head,
                                    const u32 *__restrict__ hpos, u32 n, u64 *__restrict__ out) {
  HUMID_GUARD_LAST_VGPR();
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && head[i]) out[hpos[i]] = sorted[i];
}

// every CSR row ascending (the order NLeaf::neighbours has under the trie hypotheses H1+H2)
static __global__ void __launch_bounds__(256)
k_sort_lists(const u32 *__restrict__ off, u32 n, u32 *idx) {
  HUMID_GUARD_LAST_VGPR();
  u32 u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= n) return;
  const u32 b = off[u], d = off[u + 1] - b;
  if (d < 2) return;
  u32 *a = idx + b;
  if (d <= 32) {
    u32 v[32];
    for (u32 k = 0; k < d; k++) v[k] = a[k];
    for (u32 k = 1; k < d; k++) {          // insertion sort
      u32 x = v[k];
      u32 m = k;
      while (m > 0 && v[m - 1] > x) { v[m] = v[m - 1]; m--; }
      v[m] = x;
    }
    for (u32 k = 0; k < d; k++) a[k] = v[k];
  } else {                                 // heap sort in place
    for (u32 start = d / 2; start-- > 0;) {
      u32 r = start;
      while (true) {
        u32 ch = 2 * r + 1;
        if (ch >= d) break;
        if (ch + 1 < d && a[ch + 1] > a[ch]) ch++;
        if (a[r] >= a[ch]) break;
        u32 t = a[r]; a[r] = a[ch]; a[ch] = t;
        r = ch;
      }
    }
    for (u32 end = d - 1; end > 0; end--) {
      u32 t = a[0]; a[0] = a[end]; a[end] = t;
      u32 r = 0;
      while (true) {
        u32 ch = 2 * r + 1;
        if (ch >= end) break;
        if (ch + 1 < end && a[ch + 1] > a[ch]) ch++;
        if (a[r] >= a[ch]) break;
        u32 t2 = a[r]; a[r] = a[ch]; a[ch] = t2;
        r = ch;
      }
    }
  }
}

// --------------------------------------------------------------------------------
// 4. connected components: sizes, and the split small / big
// --------------------------------------------------------------------------------
#define SMALL_COMP 32u      // components up to this many leaves are clustered by one lane, in registers

// flatten the forest and count the leaves of every component at its root
// n_dev (may be null): the number of nodes as the device knows it, when the launch is sized by a bound
static __global__ void k_comp_stats(const u32 *__restrict__ deg, u32 *P, u32 n, u32 *csize, const u32 *__restrict__ n_dev = nullptr) {
  HUMID_GUARD_LAST_VGPR();
  u32 u = blockIdx.x * blockDim.x + threadIdx.x;
  if (n_dev && *n_dev < n) n = *n_dev;
  if (u >= n || deg[u] == 0) return;
  const u32 root = uf_find(P, u);
  P[u] = root;
  atomicAdd(&csize[root], 1u);
}

// M = leaves with >= 1 neighbour, Mbig = those in components larger than SMALL_COMP, and the sum of
// the degrees (= 2E) in 64 bits: the CSR offsets are 32-bit, so the host must see an overflow
// BEFORE it sizes nbr_idx from a wrapped scan
#define CC_ROOTS 2048u      // listed roots a workgroup of k_comp_count holds before it flushes them
static __global__ void __launch_bounds__(256)
k_comp_count(const u32 *__restrict__ deg, const u32 *__restrict__ P, const u32 *__restrict__ csize, u32 n,
             ull *ctr, u32 *__restrict__ small_roots, const u32 *__restrict__ n_dev = nullptr) {
  HUMID_GUARD_LAST_VGPR();
  if (n_dev && *n_dev < n) n = *n_dev;
  __shared__ u32 lds[4];
  __shared__ ull ldeg;
  __shared__ u32 lroots[CC_ROOTS], lroots_n, lroots_base;
  if (threadIdx.x == 0) { ldeg = 0; lroots_n = 0; }
  __syncthreads();
  u32 m = 0, mb = 0;
  ull ds = 0;
  const u32 lane = threadIdx.x & 63;
  u32 round = 0;
  for (u32 u0 = blockIdx.x * blockDim.x; u0 < n; u0 += gridDim.x * blockDim.x) {     // (whole waves stay in the loop: ballot below)
    const u32 u = u0 + threadIdx.x;
    bool small_root = false;
    if (u < n) {
      const u32 dg = deg[u];
      if (dg) {
        m++;
        ds += dg;
        const u32 root = P[u];                           // P was flattened by k_comp_stats
        const u32 cs = csize[root];
        if (cs > SMALL_COMP) mb++;
        small_root = root == u && cs > 2 && cs <= SMALL_COMP;
      }
    }
    // the roots of the components k_cluster_small takes, as a dense list: a thread per LISTED root there
    // instead of a thread per leaf that mostly finds nothing to do.  Collected per workgroup in LDS
    // and flushed with ONE global atomic when the buffer fills or the loop ends (a single-address
    // atomic per wave cost 100 us here: ~40 k of them serialise at ~12 ns each).
    if (small_roots) {
      const u64 bm = __ballot(small_root);
      if (bm) {
        u32 base = 0;
        if (lane == 0) base = atomicAdd(&lroots_n, (u32)__popcll(bm));
        base = __shfl(base, 0);
        if (small_root) lroots[base + (u32)__popcll(bm & ((1ull << lane) - 1ull))] = u;
      }
      // CC_ROOTS / 256 rounds cannot overfill the buffer: only every eighth round meets at a barrier and flushes
      if ((++round & (CC_ROOTS / 256u - 1u)) == 0) {
        __syncthreads();
        const u32 cntl = lroots_n;
        if (cntl) {                                      // (uniform: read after the barrier)
          if (threadIdx.x == 0) lroots_base = (u32)atomicAdd(&ctr[CTR_SMALLROOTS], (ull)cntl);
          __syncthreads();
          for (u32 k = threadIdx.x; k < cntl; k += blockDim.x) small_roots[lroots_base + k] = lroots[k];
          __syncthreads();
          if (threadIdx.x == 0) lroots_n = 0;
        }
        __syncthreads();
      }
    }
  }
  if (small_roots) {
    __syncthreads();
    const u32 cntl = lroots_n;
    if (cntl) {
      if (threadIdx.x == 0) lroots_base = (u32)atomicAdd(&ctr[CTR_SMALLROOTS], (ull)cntl);
      __syncthreads();
      for (u32 k = threadIdx.x; k < cntl; k += blockDim.x) small_roots[lroots_base + k] = lroots[k];
    }
  }
  const u32 tm = block_sum(m, lds);                    // (its barriers also publish ldeg = 0)
  const u32 tb = block_sum(mb, lds);
#pragma unroll
  for (u32 d = 32; d >= 1; d >>= 1) ds += __shfl_xor(ds, d);
  if ((threadIdx.x & 63) == 0 && ds) atomicAdd(&ldeg, ds);
  __syncthreads();
  if (threadIdx.x == 0) {
    if (tm) atomicAdd(&ctr[CTR_NONSINGLE], (ull)tm);
    if (tb) atomicAdd(&ctr[CTR_MEMBERS], (ull)tb);
    if (ldeg) atomicAdd(&ctr[CTR_EDGES], ldeg);
  }
}

// one pass instead of three memsets and an iota: parent[i] = i, deg = csize = cur = 0 (deg has
// n + 1 entries: the extra one is the sentinel of the exclusive scan)
static __global__ void k_graph_init(u32 *__restrict__ parent, u32 *__restrict__ deg, u32 *__restrict__ csize,
                             u32 *__restrict__ cur, u32 n) {
  HUMID_GUARD_LAST_VGPR();
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { parent[i] = i; deg[i] = 0; csize[i] = 0; cur[i] = 0; }
  if (i == n) deg[i] = 0;
}

static __global__ void k_iota(u32 *p, u32 n) {
  HUMID_GUARD_LAST_VGPR();
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = i;
}

// explicit-graph entry point: union every CSR entry (u, nbr) the clustering can cross
static __global__ void k_union_csr(const u32 *__restrict__ off, const u32 *__restrict__ idx, u32 n, u32 *P,
                            const u32 *__restrict__ cnt) {
  HUMID_GUARD_LAST_VGPR();
  u32 u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= n) return;
  for (u32 k = off[u]; k < off[u + 1]; k++)
    if (idx[k] > u && joins_for_clustering(cnt, u, idx[k])) uf_union(P, u, idx[k]);   // lists are symmetric (checked)
}

// members of the BIG components, keyed (root << 32 | rank); unordered, sorted afterwards
static __global__ void __launch_bounds__(256)
k_member_keys(const u32 *__restrict__ deg, u32 *P, const u32 *__restrict__ csize, u32 n, u64 *mkeys,
              ull *ctr) {
  HUMID_GUARD_LAST_VGPR();
  __shared__ u32 lds[8];
  const u32 chunk = (n + gridDim.x - 1) / gridDim.x;
  const u32 lo = blockIdx.x * chunk;
  const u32 hi = (lo + chunk < n) ? lo + chunk : n;
  u32 mine = 0;
  for (u32 u = lo + threadIdx.x; u < hi; u += 256)
    mine += (deg[u] && csize[uf_find(P, u)] > SMALL_COMP) ? 1u : 0u;
  const u32 total = block_sum(mine, lds);
  if (threadIdx.x == 0) lds[4] = total ? (u32)atomicAdd(&ctr[CTR_SPECIAL], (ull)total) : 0u;
  __syncthreads();
  u32 base = lds[4];
  if (total == 0) return;
  for (u32 u0 = lo; u0 < hi; u0 += 256) {
    const u32 u = u0 + threadIdx.x;
    u32 root = 0;
    bool mem = (u < hi) && deg[u] != 0;
    if (mem) { root = uf_find(P, u); mem = csize[root] > SMALL_COMP; }
    u32 tot;
    const u32 r = block_rank(mem, lds, &tot);
    if (mem) mkeys[base + r] = ((u64)root << 32) | u;
    base += tot;
  }
}


#endif  // HUMID_KERNELS_GRAPH_HIP_H
