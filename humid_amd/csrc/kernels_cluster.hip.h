// kernels_cluster.hip.h -- order-exact directional / maximum clustering per component (src/cluster.cc)
// Part of libhumid_hip.so (see humid_hip.hip for the pipeline and the C ABI).  Device code for
// gfx950 only; included once, in this order, by humid_hip.hip.
#ifndef HUMID_KERNELS_CLUSTER_HIP_H
#define HUMID_KERNELS_CLUSTER_HIP_H

#include "common.hip.h"
#include "kernels_graph.hip.h"
#include "kernels_count.hip.h"

// --------------------------------------------------------------------------------
// 5. clustering
// --------------------------------------------------------------------------------
#define MAX_CLIMB 40u   // hops of maxNeighbour_: counts >= 1 double per hop, so 32 suffice; see below

// The findClusters loop over the leaves of ONE connected component, ascending.  Literal
// restatement of
//   findClusters loop            /root/reference/src/humid.cc:176-189  (members ascending)
//   maxNeighbour_                src/cluster.cc:39-51  (first qualifying neighbour, restart)
//   assignDirectionalCluster_    src/cluster.cc:58-69  (pre-order flood, explicit stack)
//   assignMaxCluster             src/cluster.cc:72-80
// A cluster is named by its creating leaf (cl_of = creator rank + 1); ids come later from a
// prefix sum over creators, which reproduces `id++` in walk order.  `st` holds 2 words per member.
// St: anything indexable as st[i] -> u32& (a pointer into scratch, or a strided view of LDS)
template <bool MAXIMUM, class MemberAt, class St>
__device__ __forceinline__ void cluster_one_component(MemberAt member_at, u32 n_members,
                                                      const u32 *__restrict__ cnt,
                                                      const u32 *__restrict__ off,
                                                      const u32 *__restrict__ idx, u32 *cl_of,
                                                      u32 *maxleaf, u64 *cl_size, St st) {
  for (u32 m = 0; m < n_members; m++) {
    const u32 u = member_at(m);
    if (cl_of[u] != 0) continue;                  // src/humid.cc:179
    const u32 label = u + 1;                      // new Cluster, creator u
    u32 start = u;
    u32 best = u;
    u32 bestc = 0;
    if (!MAXIMUM) {
      // maxNeighbour_
      u32 leaf = u;
      u32 k = off[leaf], kend = off[leaf + 1];
      u64 lc = cnt[leaf];
      u32 hops = 0;                               // every hop at least doubles a count >= 1: a
      while (k < kend) {                          // 32-bit count allows 32; the cap only matters
        u32 nb = idx[k++];                        // for a caller's zero counts (never spin)
        if (cl_of[nb] == 0 && at_least_double(cnt[nb], lc) && hops < MAX_CLIMB) {
          leaf = nb; lc = cnt[leaf];
          k = off[leaf]; kend = off[leaf + 1];
          hops++;
        }
      }
      start = leaf;
      best = leaf;                                // updateMaxCount_ once, cluster.cc:85
    }
    u64 size = 0;
    u32 depth = 0;
    // assignLeaf_(start)
    cl_of[start] = label;
    size += cnt[start];
    if (MAXIMUM) { bestc = cnt[start]; best = start; }
    st[0] = start; st[1] = off[start]; depth = 1;
    while (depth) {
      const u32 cur = st[2 * (depth - 1)];
      u32 k = st[2 * (depth - 1) + 1];
      const u32 kend = off[cur + 1];
      const u64 cc = cnt[cur];
      bool descended = false;
      while (k < kend) {
        const u32 nb = idx[k++];
        if (cl_of[nb] != 0) continue;
        if (!MAXIMUM && !at_least_double(cc, cnt[nb])) continue;
        cl_of[nb] = label;
        const u32 nc = cnt[nb];
        size += nc;
        if (MAXIMUM && nc > bestc) { bestc = nc; best = nb; }   // updateMaxCount_ strict >
        st[2 * (depth - 1) + 1] = k;
        st[2 * depth] = nb; st[2 * depth + 1] = off[nb];
        depth++;
        descended = true;
        break;
      }
      if (!descended) depth--;
    }
    maxleaf[u] = best;
    cl_size[u] = size;
  }
}

// BIG components: one lane per component = the head of its run in the sorted member keys;
// stack in HBM scratch (2 words per member of the run).
template <bool MAXIMUM>
__global__ void __launch_bounds__(64)
k_cluster_components(const u64 *__restrict__ mkeys, u32 n_members, const u32 *__restrict__ cnt,
                     const u32 *__restrict__ off, const u32 *__restrict__ idx, u32 *cl_of,
                     u32 *maxleaf, u64 *cl_size, u32 *stk) {
  HUMID_GUARD_LAST_VGPR();
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_members) return;
  const u32 root = (u32)(mkeys[i] >> 32);
  if (i > 0 && (u32)(mkeys[i - 1] >> 32) == root) return;   // not a component head
  u32 len = 1;
  while (i + len < n_members && (u32)(mkeys[i + len] >> 32) == root) len++;
  cluster_one_component<MAXIMUM>([&](u32 m) { return (u32)mkeys[i + m]; }, len, cnt, off, idx, cl_of,
                                 maxleaf, cl_size, stk + 2 * (u64)i);
}

// BIG components, directional method, one WORKGROUP per component.  The sequential loop of the
// reference is kept where order matters (members ascending, the climb hop by hop, first
// qualifying neighbour in list order) and parallelised where it does not: the flood is the
// closure of "cnt[cur] >= 2 cnt[nb]" over unassigned leaves -- the same set in any visiting order
// (src/cluster.cc:58-69), its size a commutative sum -- so it runs as a level-synchronous BFS
// with the neighbour lists scanned by whole waves.  A dense 100 k-leaf component costs
// O(edges / 256) here instead of O(edges) dependent loads on one lane.
// heads[h] = position in mkeys of the first member of component h; fr = 2 words of scratch per
// member.  cl_of is claimed with atomicCAS and read with agent-scope loads.
static __global__ void __launch_bounds__(256)
k_cluster_big_coop(const u64 *__restrict__ mkeys, u32 n_members, const u32 *__restrict__ heads,
                   const ull *__restrict__ ctr, const u32 *__restrict__ cnt, const u32 *__restrict__ off,
                   const u32 *__restrict__ idx, u32 *cl_of, u32 *maxleaf, u64 *cl_size, u32 *fr) {
  HUMID_GUARD_LAST_VGPR();
  __shared__ u32 s_min;         // block-wide minimum (next unassigned member / first qualifying neighbour)
  __shared__ u32 s_next;        // size of the next BFS frontier
  __shared__ ull s_size;        // reads in the cluster being flooded
  const u32 n_heads = (u32)ctr[CTR_SPECIAL];
  const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (u32 h = blockIdx.x; h < n_heads; h += gridDim.x) {
    const u32 i0 = heads[h];
    const u32 root = (u32)(mkeys[i0] >> 32);
    // length of the run of this component's members
    u32 len = 0;
    for (u32 base = i0;; base += 256) {
      if (threadIdx.x == 0) s_min = NONE32;
      __syncthreads();
      const u32 p = base + threadIdx.x;
      if (p >= n_members || (u32)(mkeys[p] >> 32) != root) atomicMin(&s_min, p);
      __syncthreads();
      const u32 e = s_min;
      __syncthreads();
      if (e != NONE32) { len = e - i0; break; }
    }
    u32 *fa = fr + 2 * (u64)i0, *fb = fa + len;
    u32 pos = 0;
    while (true) {
      // next unassigned member at or after pos (src/humid.cc:178-179)
      u32 hit = NONE32;
      for (; pos < len; pos += 256) {
        if (threadIdx.x == 0) s_min = NONE32;
        __syncthreads();
        const u32 m = pos + threadIdx.x;
        if (m < len && ld_agent(&cl_of[(u32)mkeys[i0 + m]]) == 0) atomicMin(&s_min, m);
        __syncthreads();
        hit = s_min;
        __syncthreads();
        if (hit != NONE32) break;
      }
      if (hit == NONE32) break;
      pos = hit;
      const u32 u = (u32)mkeys[i0 + pos];
      const u32 label = u + 1;
      // maxNeighbour_: hop to the FIRST unassigned neighbour (list order) with >= 2x the count
      u32 leaf = u;
      for (u32 hops = 0; hops < MAX_CLIMB; hops++) {
        if (threadIdx.x == 0) s_min = NONE32;
        __syncthreads();
        const u32 b = off[leaf], e = off[leaf + 1];
        const u64 lc = cnt[leaf];
        for (u32 k = b + threadIdx.x; k < e; k += 256) {
          const u32 nb = idx[k];
          if (ld_agent(&cl_of[nb]) == 0 && at_least_double(cnt[nb], lc)) { atomicMin(&s_min, k - b); break; }
        }
        __syncthreads();
        const u32 kmin = s_min;
        __syncthreads();
        if (kmin == NONE32) break;
        leaf = idx[b + kmin];
      }
      // flood from `leaf`
      if (threadIdx.x == 0) {
        atomicExch(&cl_of[leaf], label);
        fa[0] = leaf;
        s_size = cnt[leaf];
        s_next = 0;
      }
      __syncthreads();
      u32 nfr = 1;
      u32 *cur_f = fa, *nxt_f = fb;
      while (nfr) {
        for (u32 f = wave; f < nfr; f += 4) {
          const u32 cu = cur_f[f];
          const u64 cc = cnt[cu];
          for (u32 k = off[cu] + lane; k < off[cu + 1]; k += 64) {
            const u32 nb = idx[k];
            const u32 nc = cnt[nb];
            if (at_least_double(cc, nc) && ld_agent(&cl_of[nb]) == 0 && atomicCAS(&cl_of[nb], 0u, label) == 0u) {
              nxt_f[atomicAdd(&s_next, 1u)] = nb;
              atomicAdd(&s_size, (ull)nc);
            }
          }
        }
        __syncthreads();
        nfr = s_next;
        __syncthreads();
        if (threadIdx.x == 0) s_next = 0;
        u32 *t = cur_f; cur_f = nxt_f; nxt_f = t;
        __syncthreads();
      }
      if (threadIdx.x == 0) { maxleaf[u] = leaf; cl_size[u] = s_size; }
      __syncthreads();
      pos++;
    }
  }
}

// heads of the runs of equal root in the sorted member keys (fixed grid, one atomic per block)
static __global__ void __launch_bounds__(256)
k_comp_heads(const u64 *__restrict__ mkeys, u32 n_members, u32 *__restrict__ heads, ull *ctr) {
  HUMID_GUARD_LAST_VGPR();
  __shared__ u32 lds[8];
  const u32 chunk = (n_members + gridDim.x - 1) / gridDim.x;
  const u32 lo = blockIdx.x * chunk;
  const u32 hi = (lo + chunk < n_members) ? lo + chunk : n_members;
  auto is_head = [&](u32 i) { return i == 0 || (u32)(mkeys[i] >> 32) != (u32)(mkeys[i - 1] >> 32); };
  u32 mine = 0;
  for (u32 i = lo + threadIdx.x; i < hi; i += 256) mine += is_head(i) ? 1u : 0u;
  const u32 total = block_sum(mine, lds);
  if (threadIdx.x == 0) lds[4] = total ? (u32)atomicAdd(&ctr[CTR_SPECIAL], (ull)total) : 0u;
  __syncthreads();
  u32 base = lds[4];
  if (total == 0) return;
  for (u32 i0 = lo; i0 < hi; i0 += 256) {
    const u32 i = i0 + threadIdx.x;
    const bool hd = (i < hi) && is_head(i);
    u32 tot;
    const u32 r = block_rank(hd, lds, &tot);
    if (hd) heads[base + r] = i;
    base += tot;
  }
}

// The two trivial cases in one pass over the leaves, every cl_of entry written by exactly one
// lane:
//  * no neighbours: the leaf creates its own cluster (src/humid.cc:179-187 with an empty list:
//    maxNeighbour_ returns the leaf, cluster.cc:39-51);
//  * components of exactly two leaves a < b (one centre + one satellite: the bulk of the
//    non-trivial components on UMI data): closed form of the same loop, done by a's lane (the
//    root: uf_union keeps the smaller index on top);
//  * every other leaf is marked unassigned for k_cluster_small / k_cluster_components.
template <bool MAXIMUM>
__global__ void __launch_bounds__(256)
k_cluster_trivial(const u32 *__restrict__ deg, const u32 *__restrict__ P, const u32 *__restrict__ csize, u32 n,
                  const u32 *__restrict__ cnt, const u32 *__restrict__ off, const u32 *__restrict__ idx,
                  u32 *cl_of, u32 *maxleaf, u64 *cl_size) {
  HUMID_GUARD_LAST_VGPR();
  u32 a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= n) return;
  if (deg[a] == 0) { cl_of[a] = a + 1; maxleaf[a] = a; cl_size[a] = cnt[a]; return; }
  const u32 root = P[a];                           // flattened by k_comp_stats
  const u32 members = csize[root];
  // alone in its component although it has neighbours (joins_for_clustering: none of them within
  // reach of a climb or a flood): the same as no neighbours
  if (members == 1) { cl_of[a] = a + 1; maxleaf[a] = a; cl_size[a] = cnt[a]; return; }
  if (members != 2) { cl_of[a] = 0; return; }
  if (root != a) return;                           // b: written by a's lane
  u32 b = a;                                       // the other member: the neighbour with the same root
  for (u32 k = off[a]; k < off[a + 1]; k++) {
    const u32 nb = idx[k];
    if (P[nb] == root) { b = nb; break; }
  }
  const u64 ca = cnt[a], cb = cnt[b];
  if (MAXIMUM) {                                   // whole component, maxLeaf = first strict maximum
    cl_of[a] = a + 1; cl_of[b] = a + 1;
    maxleaf[a] = (cb > ca) ? b : a;
    cl_size[a] = ca + cb;
    return;
  }
  if (at_least_double(cb, ca)) {                   // a climbs to b, b floods back to a
    cl_of[a] = a + 1; cl_of[b] = a + 1;
    maxleaf[a] = b;
    cl_size[a] = ca + cb;
  } else if (at_least_double(ca, cb)) {            // a stays, absorbs b
    cl_of[a] = a + 1; cl_of[b] = a + 1;
    maxleaf[a] = a;
    cl_size[a] = ca + cb;
  } else {                                         // two clusters; b finds a already assigned
    cl_of[a] = a + 1; maxleaf[a] = a; cl_size[a] = ca;
    cl_of[b] = b + 1; maxleaf[b] = b; cl_size[b] = cb;
  }
}

// SMALL components (<= SMALL_COMP leaves): one lane per component root collects the members by
// a breadth-first walk, orders them, and runs the same loop with member list and stack in
// private memory.  No sort, no scratch.
template <bool MAXIMUM>
__global__ void __launch_bounds__(128)
k_cluster_small(const u32 *__restrict__ roots, const ull *__restrict__ ctr, const u32 *__restrict__ P,
                const u32 *__restrict__ csize, u32 n, const u32 *__restrict__ cnt, const u32 *__restrict__ off,
                const u32 *__restrict__ idx, u32 *cl_of, u32 *maxleaf, u64 *cl_size) {
  HUMID_GUARD_LAST_VGPR();
  // one lane per LISTED root (k_comp_count: components of 3 .. SMALL_COMP leaves; one and two leaves:
  // k_cluster_trivial; more: k_cluster_components / k_cluster_big_coop)
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (u32)ctr[CTR_SMALLROOTS]) return;
  const u32 u = roots[t];
  if (u >= n) return;
  const u32 target = csize[u];
  if (target > SMALL_COMP || target <= 2) return;
  u32 mem[SMALL_COMP];
  u32 st[2 * SMALL_COMP];
  u32 nm = 1;
  mem[0] = u;
  for (u32 q = 0; q < nm && nm < target; q++) {
    const u32 v = mem[q];
    for (u32 k = off[v]; k < off[v + 1] && nm < target; k++) {
      const u32 nb = idx[k];
      bool seen = P[nb] != u;                         // a neighbour outside this component (joins_for_clustering)
      for (u32 t = 0; t < nm; t++) seen |= (mem[t] == nb);
      if (!seen) mem[nm++] = nb;
    }
  }
  for (u32 k = 1; k < nm; k++) {              // ascending = walk order inside the component
    u32 x = mem[k];
    u32 m = k;
    while (m > 0 && mem[m - 1] > x) { mem[m] = mem[m - 1]; m--; }
    mem[m] = x;
  }
  cluster_one_component<MAXIMUM>([&](u32 m) { return mem[m]; }, nm, cnt, off, idx, cl_of, maxleaf, cl_size, st);
}

// The same with the member list and the stack in LDS instead of private (scratch) memory: column
// `lane` of two [depth][64] tables, so that the lanes of a wave hit 64 different banks.  64 threads per
// workgroup, 24 KB of LDS.
struct LdsColumn {
  u32 *base;                                            // &table[0][lane]
  __device__ __forceinline__ u32 &operator[](u32 i) const { return base[i * 64u]; }
};
template <bool MAXIMUM>
__global__ void __launch_bounds__(64)
k_cluster_small_lds(const u32 *__restrict__ roots, const ull *__restrict__ ctr, const u32 *__restrict__ P,
                    const u32 *__restrict__ csize, u32 n, const u32 *__restrict__ cnt, const u32 *__restrict__ off,
                    const u32 *__restrict__ idx, u32 *cl_of, u32 *maxleaf, u64 *cl_size) {
  HUMID_GUARD_LAST_VGPR();
  __shared__ u32 lmem[SMALL_COMP][64];
  __shared__ u32 lst[2 * SMALL_COMP][64];
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (u32)ctr[CTR_SMALLROOTS]) return;
  const u32 u = roots[t];
  if (u >= n) return;
  const u32 target = csize[u];
  if (target > SMALL_COMP || target <= 2) return;
  const LdsColumn mem{&lmem[0][threadIdx.x]}, st{&lst[0][threadIdx.x]};
  u32 nm = 1;
  mem[0] = u;
  for (u32 q = 0; q < nm && nm < target; q++) {
    const u32 v = mem[q];
    for (u32 k = off[v]; k < off[v + 1] && nm < target; k++) {
      const u32 nb = idx[k];
      bool seen = P[nb] != u;                         // a neighbour outside this component (joins_for_clustering)
      for (u32 j = 0; j < nm; j++) seen |= (mem[j] == nb);
      if (!seen) { mem[nm] = nb; nm++; }
    }
  }
  for (u32 k = 1; k < nm; k++) {              // ascending = walk order inside the component
    const u32 x = mem[k];
    u32 m = k;
    while (m > 0 && mem[m - 1] > x) { mem[m] = mem[m - 1]; m--; }
    mem[m] = x;
  }
  cluster_one_component<MAXIMUM>([&](u32 m) { return mem[m]; }, nm, cnt, off, idx, cl_of, maxleaf, cl_size, st);
}

static __global__ void k_creator_flags(const u32 *__restrict__ cl_of, u32 n, u32 *flag) {
  HUMID_GUARD_LAST_VGPR();
  u32 u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u < n) flag[u] = (cl_of[u] == u + 1) ? 1u : 0u;
}

// per node: final cluster id (creators numbered in walk order) and maxLeaf flag; on one GPU also
// the per-slot result word (slot_out != null), which saves the separate k_slot_results pass
static __global__ void k_finalize_nodes(const u32 *__restrict__ cl_of, const u32 *__restrict__ pos,
                                 const u32 *__restrict__ maxleaf, u32 n, u32 *__restrict__ cid,
                                 u8 *__restrict__ ismax, const u32 *__restrict__ s_first,
                                 const u32 *__restrict__ s_slot, u64 *__restrict__ slot_out) {
  HUMID_GUARD_LAST_VGPR();
  u32 u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= n) return;
  const u32 creator = cl_of[u] - 1;
  const u32 c = pos[creator] + 1;
  const bool mx = maxleaf[creator] == u;
  cid[u] = c;
  ismax[u] = mx ? 1 : 0;
  if (slot_out) slot_out[s_slot[u]] = ((u64)(mx ? s_first[u] : NONE32) << 32) | c;
}

// per hash slot: (cluster id, read to keep) of the word it holds
static __global__ void k_slot_results(const u32 *__restrict__ l_cid, const u8 *__restrict__ l_ismax,
                               const u32 *__restrict__ s_first, const u32 *__restrict__ s_slot, u32 n,
                               u64 *__restrict__ slot_out) {
  HUMID_GUARD_LAST_VGPR();
  u32 u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= n) return;
  slot_out[s_slot[u]] = ((u64)(l_ismax[u] ? s_first[u] : NONE32) << 32) | l_cid[u];
}

static __global__ void k_export_clusters(const u32 *__restrict__ flag, const u32 *__restrict__ pos,
                                  const u32 *__restrict__ maxleaf, const u64 *__restrict__ cl_size,
                                  const u32 *__restrict__ cnt, u32 n, u64 *o_size, u32 *o_maxcount,
                                  u32 *o_maxleaf) {
  HUMID_GUARD_LAST_VGPR();
  u32 u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= n || !flag[u]) return;
  const u32 c = pos[u];
  if (o_size) o_size[c] = cl_size[u];
  if (o_maxleaf) o_maxleaf[c] = maxleaf[u];
  if (o_maxcount) o_maxcount[c] = cnt[maxleaf[u]];
}


#endif  // HUMID_KERNELS_CLUSTER_HIP_H
