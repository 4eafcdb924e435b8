// sharded.cpp -- see sharded.hpp.  Host C++ above the C ABI: HIP runtime calls for memory, streams and
// peer copies, RCCL (loaded on demand) for the exchanges, humid_stage_* for all the compute.
#include "sharded.hpp"

#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <thread>

namespace humid_host {
namespace {

constexpr unsigned HIST_BITS = 12;      // top word bits of the range histogram (humid_amd/sharded.py)
constexpr unsigned MAX_RANKS = 16;      // humid_stage_route / humid_stage_combo_route (kernels_map.hip.h)

// ---- RCCL, loaded when a run wants it (the single-GPU start-up never pays for the library) ----
struct Rccl {
  void *lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  bool load() {
    lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!lib) return false;
    auto sym = [&](const char *n) { return dlsym(lib, n); };
    GetUniqueId = (decltype(GetUniqueId))sym("ncclGetUniqueId");
    CommInitRank = (decltype(CommInitRank))sym("ncclCommInitRank");
    CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
    GroupStart = (decltype(GroupStart))sym("ncclGroupStart");
    GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd");
    Send = (decltype(Send))sym("ncclSend");
    Recv = (decltype(Recv))sym("ncclRecv");
    GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
    return GetUniqueId && CommInitRank && CommDestroy && GroupStart && GroupEnd && Send && Recv && GetErrorString;
  }
};

// ---- what the ranks (threads of this process) share ----
struct Group {
  unsigned P = 1;
  std::vector<int> device;                 // rank -> HIP device
  bool use_rccl = false;
  Rccl rccl;
  ncclUniqueId nccl_id{};
  // barrier (C++17: no std::barrier) that a failed rank releases for everyone
  std::mutex mu;
  std::condition_variable cv;
  unsigned waiting = 0;
  uint64_t generation = 0;
  std::atomic<bool> failed{false};
  int fail_code = HUMID_OK;
  std::string fail_text;
  // host numbers: slot[r] = what rank r published for the current step
  std::vector<std::vector<uint8_t>> slot;
  // device buffers published for a peer-copy exchange
  struct Pub { const uint8_t *ptr = nullptr; std::vector<uint64_t> off, cnt; };
  std::vector<Pub> pub;
  // -s histograms, merged under mu
  std::map<uint64_t, uint64_t> hist_counts, hist_neigh;
  // the ranks come up (context, stream, communicator) while the host still parses; then they wait here
  int job_state = 0;                       // 0: not yet, 1: go, 2: there will be none
  struct Job *job = nullptr;
  bool wait_for_job() {
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [&] { return job_state != 0 || failed.load(); });
    return job_state == 1 && !failed.load();
  }
  void post_job(struct Job *j) {
    std::lock_guard<std::mutex> lk(mu);
    job = j;
    job_state = j ? 1 : 2;
    cv.notify_all();
  }

  void fail(int code, const std::string &text) {
    std::lock_guard<std::mutex> lk(mu);
    if (!failed.load()) { fail_code = code; fail_text = text; failed.store(true); }
    cv.notify_all();
  }
  // false: some rank failed (nobody waits any longer)
  bool barrier() {
    std::unique_lock<std::mutex> lk(mu);
    if (failed.load()) return false;
    const uint64_t gen = generation;
    if (++waiting == P) { waiting = 0; generation++; cv.notify_all(); return true; }
    cv.wait(lk, [&] { return generation != gen || failed.load(); });
    return !failed.load();
  }
};

struct DevBuf {                            // device memory of one rank, grown on demand
  void *p = nullptr;
  size_t cap = 0;
  ~DevBuf() { if (p) (void)hipFree(p); }
  hipError_t ensure(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
    const size_t want = bytes + bytes / 8 + 256;
    hipError_t e = hipMalloc(&p, want);
    if (e == hipSuccess) cap = want;
    return e;
  }
  template <class T> T *as() const { return (T *)p; }
};

struct Range { uint64_t lo = 1, hi = 0, expected = 0; };   // lo > hi: empty

// P ordered, disjoint, covering value ranges with balanced usable-read counts, cut at histogram bins
// (humid_amd/sharded.py splitters_from_hist; identical on every rank)
std::vector<Range> splitters_from_hist(const std::vector<uint64_t> &hist, unsigned P, unsigned word_nt, unsigned bits) {
  const unsigned shift = 2 * word_nt - bits;
  const size_t n_bins = hist.size();
  std::vector<uint64_t> cum(n_bins);
  uint64_t total = 0;
  for (size_t i = 0; i < n_bins; i++) { total += hist[i]; cum[i] = total; }
  std::vector<size_t> bounds{0};
  for (unsigned k = 1; k < P; k++) {
    const uint64_t target = (total * k + P - 1) / P;
    size_t b = (size_t)(std::lower_bound(cum.begin(), cum.end(), target) - cum.begin()) + 1;
    b = std::min(std::max(b, bounds.back()), n_bins);
    bounds.push_back(b);
  }
  bounds.push_back(n_bins);
  std::vector<Range> out(P);
  for (unsigned r = 0; r < P; r++) {
    const size_t b0 = bounds[r], b1 = bounds[r + 1];
    if (b1 <= b0) continue;
    out[r].lo = (uint64_t)b0 << shift;
    out[r].hi = (r == P - 1 || ((uint64_t)b1 << shift) == 0) ? ~0ull : ((uint64_t)b1 << shift) - 1;
    out[r].expected = cum[b1 - 1] - (b0 ? cum[b0 - 1] : 0);
  }
  return out;
}

// count_order for humid_stage_count_dense from the global histogram (sharded.py _order_hint)
int order_hint(const std::vector<uint64_t> &hist, const Range &rg, unsigned word_nt, unsigned bits) {
  if (rg.lo > rg.hi) return -1;
  const unsigned shift = 2 * word_nt - bits;
  const size_t b0 = (size_t)(rg.lo >> shift), b1 = std::min<size_t>((size_t)(rg.hi >> shift), hist.size() - 1);
  if (b1 + 1 - b0 < 4) return -1;
  double sum = 0, mx = 0;
  for (size_t b = b0; b <= b1; b++) { sum += (double)hist[b]; mx = std::max(mx, (double)hist[b]); }
  if (sum < 65536) return -1;
  const double ratio = mx / (sum / (double)(b1 + 1 - b0));
  return ratio <= 1.25 ? 1 : (ratio > 2.5 ? 0 : -1);
}

struct Rank {
  Group &g;
  const unsigned r;
  humid_ctx *ctx = nullptr;
  hipStream_t st = nullptr;
  ncclComm_t comm = nullptr;
  std::string err;
  int code = HUMID_OK;

  Rank(Group &grp, unsigned rank) : g(grp), r(rank) {}
  ~Rank() {
    if (comm) g.rccl.CommDestroy(comm);
    if (ctx) humid_ctx_destroy(ctx);
    if (st) (void)hipStreamDestroy(st);
  }

  bool hip_ok(hipError_t e, const char *what) {
    if (e == hipSuccess) return true;
    code = e == hipErrorOutOfMemory ? HUMID_E_NOMEM : HUMID_E_HIP;
    err = std::string(what) + ": " + hipGetErrorString(e);
    return false;
  }
  bool lib_ok(int rc) {
    if (rc == HUMID_OK) return true;
    code = rc;
    err = humid_last_error(ctx);
    return false;
  }
  bool nccl_ok(ncclResult_t e, const char *what) {
    if (e == ncclSuccess) return true;
    code = HUMID_E_HIP;
    err = std::string(what) + ": " + g.rccl.GetErrorString(e);
    return false;
  }
  bool together() {                       // a barrier every rank reaches, or nobody goes on
    if (g.barrier()) return true;
    if (code == HUMID_OK) { code = g.fail_code; err = g.fail_text; }
    return false;
  }

  // host numbers of all ranks: every rank publishes n values, gets the P x n table
  template <class T>
  bool host_all_gather(const T *mine, size_t n, std::vector<T> &all) {
    g.slot[r].assign((const uint8_t *)mine, (const uint8_t *)mine + n * sizeof(T));
    if (!together()) return false;
    all.resize((size_t)g.P * n);
    for (unsigned q = 0; q < g.P; q++) std::memcpy(all.data() + (size_t)q * n, g.slot[q].data(), n * sizeof(T));
    return together();                    // nobody overwrites its slot before all have read it
  }

  // bytes between device buffers: rank q gets send[send_off[q] .. + send_cnt[q]) of every rank, laid out
  // at recv_off[src].  (send_off may alias: an all-gather sends the same bytes to everyone.)
  bool exchange(const void *send, const std::vector<uint64_t> &send_off, const std::vector<uint64_t> &send_cnt,
                void *recv, const std::vector<uint64_t> &recv_off, const std::vector<uint64_t> &recv_cnt) {
    if (!hip_ok(hipStreamSynchronize(st), "hipStreamSynchronize")) return false;   // queued stage work (route) done
    if (g.use_rccl) {
      if (!together()) return false;      // every rank is alive and about to enter the same group call
      if (!nccl_ok(g.rccl.GroupStart(), "ncclGroupStart")) return false;
      for (unsigned q = 0; q < g.P; q++) {
        if (send_cnt[q] &&
            !nccl_ok(g.rccl.Send((const uint8_t *)send + send_off[q], send_cnt[q], ncclUint8, (int)q, comm, st), "ncclSend"))
          return false;
        if (recv_cnt[q] &&
            !nccl_ok(g.rccl.Recv((uint8_t *)recv + recv_off[q], recv_cnt[q], ncclUint8, (int)q, comm, st), "ncclRecv"))
          return false;
      }
      if (!nccl_ok(g.rccl.GroupEnd(), "ncclGroupEnd")) return false;
      return hip_ok(hipStreamSynchronize(st), "hipStreamSynchronize (exchange)");
    }
    Group::Pub &mine = g.pub[r];
    mine.ptr = (const uint8_t *)send;
    mine.off = send_off;
    mine.cnt = send_cnt;
    if (!together()) return false;        // all send buffers are complete and published
    for (unsigned q = 0; q < g.P; q++) {
      const Group::Pub &src = g.pub[q];
      if (src.cnt[r] != recv_cnt[q]) { code = HUMID_E_INVALID; err = "exchange: split sizes of sender and receiver differ"; return false; }
      if (recv_cnt[q] &&
          !hip_ok(hipMemcpyAsync((uint8_t *)recv + recv_off[q], src.ptr + src.off[r], recv_cnt[q], hipMemcpyDefault, st),
                  "hipMemcpyAsync (peer copy)"))
        return false;
    }
    if (!hip_ok(hipStreamSynchronize(st), "hipStreamSynchronize (exchange)")) return false;
    return together();                    // the senders may reuse their buffers
  }
  // all-to-all of `elem`-byte items with split sizes in items
  bool all_to_all(const void *send, const std::vector<uint64_t> &send_items, void *recv,
                  const std::vector<uint64_t> &recv_items, size_t elem) {
    std::vector<uint64_t> so(g.P), sc(g.P), ro(g.P), rc(g.P);
    uint64_t a = 0, b = 0;
    for (unsigned q = 0; q < g.P; q++) {
      so[q] = a; sc[q] = send_items[q] * elem; a += sc[q];
      ro[q] = b; rc[q] = recv_items[q] * elem; b += rc[q];
    }
    return exchange(send, so, sc, recv, ro, rc);
  }
  // all-gather of n_all[r] items per rank into rank order
  bool all_gather_v(const void *send, const std::vector<uint64_t> &n_all, void *recv, size_t elem) {
    std::vector<uint64_t> so(g.P, 0), sc(g.P, n_all[r] * elem), ro(g.P), rc(g.P);
    uint64_t b = 0;
    for (unsigned q = 0; q < g.P; q++) { ro[q] = b; rc[q] = n_all[q] * elem; b += rc[q]; }
    return exchange(send, so, sc, recv, ro, rc);
  }
};

struct Job {
  const uint64_t *words;
  const uint8_t *filtered;
  uint64_t n_reads;
  uint32_t word_nt, distance, method;
  bool want_hist;
  uint32_t *cluster_id;
  uint8_t *keep;
  humid_summary sum{};                     // written by rank 0
};

#define STEP(expr) do { if (!(expr)) return false; } while (0)

bool run_rank(Rank &k) {
  Group &g = k.g;
  const unsigned P = g.P, r = k.r;
  STEP(k.hip_ok(hipSetDevice(g.device[r]), "hipSetDevice"));
  STEP(k.hip_ok(hipStreamCreateWithFlags(&k.st, hipStreamNonBlocking), "hipStreamCreate"));
  if (humid_ctx_create(&k.ctx, g.device[r], (void *)k.st) != HUMID_OK) {
    k.code = HUMID_E_HIP;
    k.err = humid_last_error(nullptr);
    return false;
  }
  if (g.use_rccl) {
    STEP(k.together());                                           // rank 0 made the id before the threads started
    STEP(k.nccl_ok(g.rccl.CommInitRank(&k.comm, (int)P, g.nccl_id, (int)r), "ncclCommInitRank"));
  }
  if (!g.wait_for_job()) return g.job_state == 2 && !g.failed.load();       // no job: a clean end
  Job &job = *g.job;
  hipStream_t st = k.st;
  const uint32_t n = job.word_nt, d = job.distance;
  const uint64_t r0 = job.n_reads * r / P, r1 = job.n_reads * (r + 1) / P, n_local = r1 - r0;

  // this rank's shard of the reads, in input order
  DevBuf d_w, d_f, d_cid, d_keep, d_hist, recv_w, e_loc, got, e_all, ret;
  STEP(k.hip_ok(d_w.ensure(n_local * 8 + 8), "hipMalloc"));
  STEP(k.hip_ok(d_f.ensure(n_local + 8), "hipMalloc"));
  STEP(k.hip_ok(d_cid.ensure(n_local * 4 + 8), "hipMalloc"));
  STEP(k.hip_ok(d_keep.ensure(n_local + 8), "hipMalloc"));
  if (n_local) {
    STEP(k.hip_ok(hipMemcpyAsync(d_w.p, job.words + r0, n_local * 8, hipMemcpyHostToDevice, st), "hipMemcpyAsync (words)"));
    STEP(k.hip_ok(hipMemcpyAsync(d_f.p, job.filtered + r0, n_local, hipMemcpyHostToDevice, st), "hipMemcpyAsync (flags)"));
  }

  // ---- 1. histograms of all ranks -> value ranges and every split size of the word exchange ----
  uint32_t n_combos1 = 0, pbits = 0;
  STEP(k.lib_ok(humid_stage_plan_info(k.ctx, n, d, 1, &n_combos1, &pbits)));
  if (pbits < 1) {
    k.code = HUMID_E_UNSUPPORTED;
    k.err = "-g: a distance this close to the word length leaves no prefix to cut value ranges at; use one GPU";
    return false;
  }
  const unsigned bits = std::min<unsigned>(std::min<unsigned>(HIST_BITS, 2 * n), pbits);
  const size_t n_bins = (size_t)1 << bits;
  STEP(k.hip_ok(d_hist.ensure(n_bins * 4), "hipMalloc"));
  STEP(k.hip_ok(hipMemsetAsync(d_hist.p, 0, n_bins * 4, st), "hipMemsetAsync"));
  STEP(k.lib_ok(humid_stage_histogram(k.ctx, d_w.as<uint64_t>(), d_f.as<uint8_t>(), n_local, n, bits, d_hist.as<uint32_t>())));
  std::vector<uint32_t> h_hist(n_bins), all_hist;
  STEP(k.hip_ok(hipMemcpyAsync(h_hist.data(), d_hist.p, n_bins * 4, hipMemcpyDeviceToHost, st), "hipMemcpyAsync (histogram)"));
  STEP(k.hip_ok(hipStreamSynchronize(st), "hipStreamSynchronize"));
  STEP(k.host_all_gather(h_hist.data(), n_bins, all_hist));
  std::vector<uint64_t> hist_sum(n_bins, 0);
  std::vector<std::vector<uint64_t>> cum(P, std::vector<uint64_t>(n_bins + 1, 0));     // per rank, cumulative
  for (unsigned q = 0; q < P; q++)
    for (size_t b = 0; b < n_bins; b++) {
      const uint64_t v = all_hist[(size_t)q * n_bins + b];
      hist_sum[b] += v;
      cum[q][b + 1] = cum[q][b] + v;
    }
  const std::vector<Range> ranges = splitters_from_hist(hist_sum, P, n, bits);
  const unsigned shift = 2 * n - bits;
  auto in_range = [&](unsigned src, unsigned owner) -> uint64_t {      // usable reads of rank src in owner's range
    const Range &rg = ranges[owner];
    if (rg.lo > rg.hi) return 0;
    const size_t b0 = (size_t)(rg.lo >> shift), b1 = std::min<size_t>((size_t)(rg.hi >> shift), n_bins - 1) + 1;
    return cum[src][b1] - cum[src][b0];
  };
  std::vector<uint64_t> send_counts(P), recv_counts(P), lo(P), hi(P);
  uint64_t n_send = 0, n_recv = 0;
  for (unsigned q = 0; q < P; q++) {
    send_counts[q] = in_range(r, q);
    recv_counts[q] = in_range(q, r);
    n_send += send_counts[q];
    n_recv += recv_counts[q];
    lo[q] = ranges[q].lo;
    hi[q] = ranges[q].hi;
  }
  uint64_t lo_r = ranges[r].lo, hi_r = ranges[r].hi;
  if (lo_r > hi_r) { lo_r = 0; hi_r = ~0ull; }                                           // empty range: nothing arrives
  STEP(k.lib_ok(humid_ctx_set_option(k.ctx, "count_order", order_hint(hist_sum, ranges[r], n, bits))));
  STEP(k.lib_ok(humid_ctx_set_option(k.ctx, "count_mode", 0)));

  // ---- 2. usable words -> owner of their range (stable: input order inside every block) ----
  const uint64_t *d_routed = nullptr;
  const uint32_t *d_perm = nullptr;
  STEP(k.lib_ok(humid_stage_route(k.ctx, d_w.as<uint64_t>(), d_f.as<uint8_t>(), n_local, lo.data(), hi.data(), P,
                                  send_counts.data(), &d_routed, &d_perm)));
  STEP(k.hip_ok(recv_w.ensure(n_recv * 8 + 8), "hipMalloc"));
  STEP(k.all_to_all(d_routed, send_counts, recv_w.p, recv_counts, 8));
  STEP(k.lib_ok(humid_stage_route_check(k.ctx)));

  // ---- 3. exact counts of the received words (all usable, all in this rank's range) ----
  const uint64_t shard_begin[2] = {0, n_recv};
  uint64_t cnt_one = 0, u_local = 0, usable_local = 0;
  STEP(k.lib_ok(humid_stage_count_dense(k.ctx, recv_w.as<uint64_t>(), nullptr, n_recv, n, lo_r, hi_r, shard_begin, 1,
                                        &cnt_one, &u_local, &usable_local)));
  const uint64_t meta[3] = {u_local, usable_local, n_local};
  std::vector<uint64_t> metas;
  STEP(k.host_all_gather(meta, 3, metas));
  uint64_t u_total = 0, goff = 0, usable = 0, total = 0;
  for (unsigned q = 0; q < P; q++) {
    if (q < r) goff += metas[3 * q];
    u_total += metas[3 * q];
    usable += metas[3 * q + 1];
    total += metas[3 * q + 2];
  }
  if (u_total >= 0xffffffffull) { k.code = HUMID_E_OVERFLOW; k.err = "more than 2^32-2 unique words in total"; return false; }
  const uint64_t *lw = nullptr;
  const uint32_t *lc = nullptr, *lfirst = nullptr;
  if (u_local) STEP(k.lib_ok(humid_stage_unique(k.ctx, &lw, &lc, &lfirst)));
  if (job.want_hist && u_local) {                                                       // counts.dat: leaf -> count
    std::vector<uint32_t> h(u_local);
    STEP(k.hip_ok(hipMemcpy(h.data(), lc, u_local * 4, hipMemcpyDeviceToHost), "hipMemcpy (counts)"));
    std::map<uint64_t, uint64_t> m;
    for (uint32_t v : h) m[v]++;
    std::lock_guard<std::mutex> lk(g.mu);
    for (auto &kv : m) g.hist_counts[kv.first] += kv.second;
  }

  // ---- 4. neighbour pairs in global unique indices, each with the counts of its endpoints ----
  uint64_t e_mine = 0;                                                                  // 16-byte records in e_loc
  auto append_pairs = [&](const uint64_t *rec, uint64_t n_rec) -> bool {
    if (!n_rec) return true;
    if ((e_mine + n_rec) * 16 > e_loc.cap) {                                            // grow, keeping what is there
      DevBuf bigger;
      STEP(k.hip_ok(bigger.ensure((e_mine + n_rec) * 32), "hipMalloc"));
      if (e_mine) STEP(k.hip_ok(hipMemcpyAsync(bigger.p, e_loc.p, e_mine * 16, hipMemcpyDeviceToDevice, st), "hipMemcpyAsync"));
      STEP(k.hip_ok(hipStreamSynchronize(st), "hipStreamSynchronize"));
      std::swap(bigger.p, e_loc.p);
      std::swap(bigger.cap, e_loc.cap);
    }
    STEP(k.hip_ok(hipMemcpyAsync(e_loc.as<uint8_t>() + e_mine * 16, rec, n_rec * 16, hipMemcpyDeviceToDevice, st), "hipMemcpyAsync"));
    STEP(k.hip_ok(hipStreamSynchronize(st), "hipStreamSynchronize"));                     // rec is a view the next call overwrites
    e_mine += n_rec;
    return true;
  };
  if (d > 0 && u_total > 1) {
    uint32_t n_combos = 0, pb2 = 0;
    STEP(k.lib_ok(humid_stage_plan_info(k.ctx, n, d, u_total, &n_combos, &pb2)));
    const uint64_t *rec = nullptr;
    uint64_t n_rec = 0;
    if (u_local > 1) {
      STEP(k.lib_ok(humid_stage_pairs_keyed(k.ctx, lw, u_local, 0, goff, lc, n, d, u_total, 0, &rec, &n_rec)));
      STEP(append_pairs(rec, n_rec));
    }
    for (uint32_t cb = 1; cb < n_combos; cb++) {
      const uint64_t *items = nullptr;
      std::vector<uint64_t> sc(P, 0), all_sc, rc(P);
      STEP(k.lib_ok(humid_stage_combo_route(k.ctx, lw, lc, u_local, goff, n, d, u_total, cb, P, &items, sc.data())));
      STEP(k.host_all_gather(sc.data(), P, all_sc));
      uint64_t n_got = 0;
      for (unsigned q = 0; q < P; q++) { rc[q] = all_sc[(size_t)q * P + r]; n_got += rc[q]; }
      STEP(k.hip_ok(got.ensure(n_got * 16 + 16), "hipMalloc"));
      STEP(k.all_to_all(items, sc, got.p, rc, 16));
      if (n_got > 1) {
        STEP(k.lib_ok(humid_stage_pairs_keyed(k.ctx, got.as<uint64_t>(), n_got, 1, 0, nullptr, n, d, u_total, cb, &rec, &n_rec)));
        STEP(append_pairs(rec, n_rec));
      }
    }
  }
  std::vector<uint64_t> e_counts;
  STEP(k.host_all_gather(&e_mine, 1, e_counts));
  uint64_t E = 0;
  for (uint64_t v : e_counts) E += v;
  STEP(k.hip_ok(e_all.ensure(E * 16 + 16), "hipMalloc"));
  STEP(k.hip_ok(e_loc.ensure(16), "hipMalloc"));
  STEP(k.all_gather_v(e_loc.p, e_counts, e_all.p, 16));

  // ---- 5. compact graph over the pairs' endpoints (replicated); ids by closed-form prefix counts ----
  const uint32_t *nodes = nullptr, *node_cnt = nullptr, *ccid = nullptr;
  const uint64_t *cedges = nullptr;
  const uint8_t *cismax = nullptr;
  uint64_t M = 0, C_c = 0;
  humid_summary gs;
  std::memset(&gs, 0, sizeof gs);
  if (E) {
    STEP(k.lib_ok(humid_stage_compact_nodes(k.ctx, e_all.as<uint64_t>(), E, 2, &nodes, &M, &cedges, &node_cnt)));
    if (job.want_hist && r == 0) {                                                       // neigh.dat: degree of every leaf
      std::vector<uint64_t> he(E);
      STEP(k.hip_ok(hipMemcpy(he.data(), cedges, E * 8, hipMemcpyDeviceToHost), "hipMemcpy (edges)"));
      std::vector<uint32_t> deg(M, 0);
      for (uint64_t e : he) { deg[e >> 32]++; deg[e & 0xffffffffull]++; }
      std::lock_guard<std::mutex> lk(g.mu);
      for (uint32_t v : deg) g.hist_neigh[v]++;
    }
    // (the graph runs over the compact node list: its "words" are only carried for the accessors)
    STEP(k.lib_ok(humid_stage_graph_edges(k.ctx, (const uint64_t *)nodes, node_cnt, M, cedges, E, n, d, job.method, &ccid,
                                          &cismax, &gs)));
    C_c = gs.clusters;
  }
  if (job.want_hist && r == 0 && u_total > M) {
    std::lock_guard<std::mutex> lk(g.mu);
    g.hist_neigh[0] += u_total - M;
  }
  const uint64_t clusters = u_total - M + C_c;
  if (clusters >= (1ull << 31)) { k.code = HUMID_E_OVERFLOW; k.err = "cluster ids exceed 31 bits"; return false; }
  const uint32_t *l_cid = nullptr;
  const uint8_t *l_ismax = nullptr;
  STEP(k.lib_ok(humid_stage_exchange_ids(k.ctx, nodes, ccid, cismax, M, C_c, goff, u_local, &l_cid, &l_ismax)));

  // ---- 6. per-read results at the owner, back to the home shards ----
  const uint32_t *packed = nullptr;
  uint64_t n_packed = 0;
  STEP(k.lib_ok(humid_stage_map_dense(k.ctx, l_cid, l_ismax, &packed, &n_packed)));
  if (n_packed != n_recv) { k.code = HUMID_E_INVALID; k.err = "map_dense returned a different number of reads than were counted"; return false; }
  STEP(k.hip_ok(ret.ensure(n_send * 4 + 8), "hipMalloc"));
  STEP(k.all_to_all(packed, recv_counts, ret.p, send_counts, 4));
  STEP(k.lib_ok(humid_stage_scatter(k.ctx, d_perm, ret.as<uint32_t>(), n_send, n_local, d_cid.as<uint32_t>(), d_keep.as<uint8_t>())));
  if (n_local) {
    STEP(k.hip_ok(hipMemcpyAsync(job.cluster_id + r0, d_cid.p, n_local * 4, hipMemcpyDeviceToHost, st), "hipMemcpyAsync (ids)"));
    STEP(k.hip_ok(hipMemcpyAsync(job.keep + r0, d_keep.p, n_local, hipMemcpyDeviceToHost, st), "hipMemcpyAsync (keep)"));
  }
  STEP(k.hip_ok(hipStreamSynchronize(st), "hipStreamSynchronize"));
  if (r == 0) {
    job.sum = gs;                                                                       // the kernel times of the graph stage
    job.sum.total = total;
    job.sum.usable = usable;
    job.sum.unique = u_total;
    job.sum.clusters = clusters;
    job.sum.edges = E;
    job.sum.nonsingle = M;
  }
  return k.together();                                                                  // peers may still be copying from this rank's buffers
}

}  // namespace

struct ShardedSession::Impl {
  Group g;
  std::vector<std::thread> threads;
  std::thread starter;
  std::string early_error;
  int early_code = HUMID_OK;
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  double ms_init = 0;
  std::atomic<unsigned> ready{0};
};

static double ms_since(std::chrono::steady_clock::time_point t) {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count();
}

ShardedSession::ShardedSession(unsigned n_ranks) : p_(new Impl) {
  Impl &m = *p_;
  Group &g = m.g;
  g.P = n_ranks;
  if (n_ranks < 1 || n_ranks > MAX_RANKS) { m.early_error = "-g takes 1 .. 16 ranks"; m.early_code = HUMID_E_INVALID; return; }
  g.device.resize(n_ranks);
  g.slot.resize(n_ranks);
  g.pub.resize(n_ranks);
  // everything that touches the HIP runtime happens on the starter thread: the caller goes on parsing
  m.starter = std::thread([&m, n_ranks] {
    Group &g = m.g;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev < 1) { g.fail(HUMID_E_HIP, "no HIP device"); return; }
    for (unsigned r = 0; r < n_ranks; r++) g.device[r] = (int)(r % (unsigned)n_dev);
    // RCCL needs a GPU per rank; HUMID_COMM=copy|rccl overrides the choice
    const char *want = getenv("HUMID_COMM");
    const bool distinct = n_ranks <= (unsigned)n_dev;
    g.use_rccl = want ? std::strcmp(want, "rccl") == 0 : distinct;
    if (g.use_rccl && !distinct) { g.fail(HUMID_E_INVALID, "HUMID_COMM=rccl needs one GPU per rank"); return; }
    if (g.use_rccl) {
      if (!g.rccl.load()) {
        std::fprintf(stderr, "humid: librccl.so not usable (%s): ranks exchange through peer copies\n", dlerror());
        g.use_rccl = false;
      } else if (g.rccl.GetUniqueId(&g.nccl_id) != ncclSuccess) {
        std::fprintf(stderr, "humid: ncclGetUniqueId failed: ranks exchange through peer copies\n");
        g.use_rccl = false;
      }
    }
    if (!g.use_rccl)                                   // peer copies: let the devices see each other's memory
      for (int a = 0; a < n_dev && a < (int)n_ranks; a++)
        for (int b = 0; b < n_dev && b < (int)n_ranks; b++)
          if (a != b && hipSetDevice(a) == hipSuccess) (void)hipDeviceEnablePeerAccess(b, 0);
    (void)hipGetLastError();
    m.ms_init = ms_since(m.t0);
    for (unsigned r = 0; r < n_ranks; r++)
      m.threads.emplace_back([&m, r] {
        Group &g = m.g;
        Rank k(g, r);
        if (!run_rank(k)) g.fail(k.code != HUMID_OK ? k.code : HUMID_E_HIP, "rank " + std::to_string(r) + ": " + k.err);
      });
  });
}

ShardedSession::~ShardedSession() {
  if (p_->starter.joinable()) p_->starter.join();
  p_->g.post_job(nullptr);                             // ranks still waiting for a job leave
  for (auto &t : p_->threads) if (t.joinable()) t.join();
  delete p_;
}

int ShardedSession::run(const uint64_t *words, const uint8_t *filtered, uint64_t n_reads, uint32_t word_nt,
                        uint32_t distance, uint32_t method, bool want_hist, uint32_t *cluster_id, uint8_t *keep,
                        ShardedResult &out) {
  Impl &m = *p_;
  Group &g = m.g;
  if (m.early_code != HUMID_OK) { out.error = m.early_error; return m.early_code; }
  if (word_nt == 0 || word_nt > 32) { out.error = "-g: words longer than 32 nt run on one GPU only"; return HUMID_E_UNSUPPORTED; }
  if (n_reads >= 0x7fffffffull * g.P) { out.error = "-g: more than 2^31-1 reads per rank"; return HUMID_E_OVERFLOW; }
  if (m.starter.joinable()) m.starter.join();
  out.ms_init = m.ms_init;
  Job job{words, filtered, n_reads, word_nt, distance, method, want_hist, cluster_id, keep, {}};
  const auto t1 = std::chrono::steady_clock::now();
  g.post_job(&job);
  for (auto &t : m.threads) t.join();
  out.ms_run = ms_since(t1);
  out.comm = g.use_rccl ? "rccl" : "copy";
  if (g.failed.load()) {
    out.error = g.fail_text;
    return g.fail_code != HUMID_OK ? g.fail_code : HUMID_E_HIP;
  }
  out.sum = job.sum;
  if (want_hist) {
    out.hist[0].assign(g.hist_counts.begin(), g.hist_counts.end());
    out.hist[1].assign(g.hist_neigh.begin(), g.hist_neigh.end());
    // clusters.dat: Cluster::size = reads of the cluster = usable reads that carry its id
    std::vector<uint32_t> size(out.sum.clusters + 1, 0);
    for (uint64_t i = 0; i < n_reads; i++)
      if (!filtered[i] && cluster_id[i] <= out.sum.clusters) size[cluster_id[i]]++;
    std::map<uint64_t, uint64_t> hm;
    for (uint64_t c = 1; c <= out.sum.clusters; c++) hm[size[c]]++;
    out.hist[2].assign(hm.begin(), hm.end());
  }
  return HUMID_OK;
}

}  // namespace humid_host
