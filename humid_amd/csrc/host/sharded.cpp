// sharded.cpp -- see sharded.hpp.  Host C++ above the C ABI: HIP runtime calls for memory, streams and
// peer copies, RCCL (loaded on demand) for the exchanges; the pass itself is the library's
// humid_dedup_run_exchange.
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

constexpr unsigned MAX_RANKS = 16;      // humid_stage_route / humid_stage_combo_route (kernels_map.hip.h)

// ---- RCCL, loaded when a run wants it (the single-GPU start-up never pays for the library) ----
struct Rccl {
  void *lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;      // optional: ends the collectives a failed pass left in flight
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
    CommAbort = (decltype(CommAbort))sym("ncclCommAbort");
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
  std::atomic<bool> rccl_failed{false};   // some rank's communicator did not come up: all ranks take the peer copies
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
  // every rank's communicator (null until it is up), so that a rank that fails can abort them ALL: a peer that
  // already waits in hipStreamSynchronize for an RCCL kernel would wait for ever
  std::vector<std::atomic<ncclComm_t>> comms;
  std::atomic<bool> aborted{false};
  void abort_comms() {
    if (!use_rccl || !rccl.CommAbort || aborted.exchange(true)) return;
    for (auto &cm : comms) {
      ncclComm_t x = cm.exchange(nullptr);
      if (x) (void)rccl.CommAbort(x);
    }
  }
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
    {
      std::lock_guard<std::mutex> lk(mu);
      if (!failed.load()) { fail_code = code; fail_text = text; failed.store(true); }
      cv.notify_all();
    }
    abort_comms();
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
    // (an aborted communicator was taken out of g.comms by abort_comms and is gone already)
    if (comm && r < g.comms.size() && g.comms[r].exchange(nullptr) == comm) g.rccl.CommDestroy(comm);
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

  // host numbers of all ranks: every rank publishes `bytes` bytes, gets the P x bytes table
  bool host_all_gather_bytes(const void *mine, uint64_t bytes, void *all) {
    g.slot[r].assign((const uint8_t *)mine, (const uint8_t *)mine + bytes);
    if (!together()) return false;
    for (unsigned q = 0; q < g.P; q++) std::memcpy((uint8_t *)all + (size_t)q * bytes, g.slot[q].data(), bytes);
    return together();                    // nobody overwrites its slot before all have read it
  }

  // bytes between device buffers: rank q gets send[send_off[q] .. + send_cnt[q]) of every rank, laid out
  // at recv_off[src].  (send_off may alias: an all-gather sends the same bytes to everyone.)
  bool exchange(const void *send, const std::vector<uint64_t> &send_off, const std::vector<uint64_t> &send_cnt,
                void *recv, const std::vector<uint64_t> &recv_off, const std::vector<uint64_t> &recv_cnt) {
    if (!hip_ok(hipStreamSynchronize(st), "hipStreamSynchronize")) return false;   // queued stage work (route) done
    if (g.use_rccl) {
      if (!together()) return false;      // every rank is alive and about to enter the same group call
      if (!nccl_ok(g.rccl.GroupStart(), "ncclGroupStart")) { g.fail(code, "rank " + std::to_string(r) + ": " + err); return false; }
      bool ok = true;
      for (unsigned q = 0; q < g.P && ok; q++) {
        if (send_cnt[q])
          ok = nccl_ok(g.rccl.Send((const uint8_t *)send + send_off[q], send_cnt[q], ncclUint8, (int)q, comm, st), "ncclSend");
        if (ok && recv_cnt[q])
          ok = nccl_ok(g.rccl.Recv((uint8_t *)recv + recv_off[q], recv_cnt[q], ncclUint8, (int)q, comm, st), "ncclRecv");
      }
      // the group is ALWAYS closed (an open group would swallow every later call of this thread); a failed
      // send / receive then makes the whole group fail: all communicators are aborted, so that no peer keeps
      // waiting in its stream for this rank's half of the exchange
      const ncclResult_t ge = g.rccl.GroupEnd();
      if (ok) ok = nccl_ok(ge, "ncclGroupEnd");
      if (!ok) { g.fail(code, "rank " + std::to_string(r) + ": " + err); return false; }
      if (!hip_ok(hipStreamSynchronize(st), "hipStreamSynchronize (exchange)")) { g.fail(code, "rank " + std::to_string(r) + ": " + err); return false; }
      if (g.failed.load()) { if (code == HUMID_OK) { code = g.fail_code; err = g.fail_text; } return false; }
      return true;
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
};

struct Job {
  const uint64_t *words;
  const uint8_t *filtered;
  uint64_t n_reads;
  uint32_t word_nt, distance, method;
  bool want_hist;
  bool edit;
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
    // a communicator that does not come up on some rank sends EVERY rank to the peer copies (the decision is
    // taken together: nobody may wait in a group call for a rank that left)
    const ncclResult_t ir = g.rccl.CommInitRank(&k.comm, (int)P, g.nccl_id, (int)r);
    if (ir != ncclSuccess) {
      std::fprintf(stderr, "humid: rank %u: ncclCommInitRank: %s\n", r, g.rccl.GetErrorString(ir));
      k.comm = nullptr;
      g.rccl_failed.store(true);
    } else
      g.comms[r].store(k.comm);
    STEP(k.together());
    if (g.rccl_failed.load()) {
      if (k.comm) { g.comms[r].store(nullptr); g.rccl.CommDestroy(k.comm); k.comm = nullptr; }
      STEP(k.together());                                         // every rank has read the flag
      if (r == 0) { g.use_rccl = false; std::fprintf(stderr, "humid: ranks exchange through peer copies\n"); }
      STEP(k.together());
    }
  }
  if (!g.wait_for_job()) return g.job_state == 2 && !g.failed.load();       // no job: a clean end
  Job &job = *g.job;
  hipStream_t st = k.st;
  const uint64_t r0 = job.n_reads * r / P, r1 = job.n_reads * (r + 1) / P, n_local = r1 - r0;

  // one slab for the library's buffers of this rank (about what it will count: its share of the reads; the
  // buffers that do not fit are allocated one by one as before)
  if (n_local) humid_ctx_reserve(k.ctx, n_local + n_local / 4, job.word_nt);
  // this rank's shard of the reads, in input order
  DevBuf d_w, d_f, d_cid, d_keep;
  const uint64_t wpr = job.word_nt > 32 ? 2 : 1;                                        // uint64 per word (include/humid_hip.h)
  STEP(k.hip_ok(d_w.ensure(n_local * 8 * wpr + 8), "hipMalloc"));
  STEP(k.hip_ok(d_f.ensure(n_local + 8), "hipMalloc"));
  STEP(k.hip_ok(d_cid.ensure(n_local * 4 + 8), "hipMalloc"));
  STEP(k.hip_ok(d_keep.ensure(n_local + 8), "hipMalloc"));
  if (n_local) {
    STEP(k.hip_ok(hipMemcpyAsync(d_w.p, job.words + r0 * wpr, n_local * 8 * wpr, hipMemcpyHostToDevice, st), "hipMemcpyAsync (words)"));
    STEP(k.hip_ok(hipMemcpyAsync(d_f.p, job.filtered + r0, n_local, hipMemcpyHostToDevice, st), "hipMemcpyAsync (flags)"));
  }
  // the pass itself is the library's (humid_dedup_run_exchange); this file moves the bytes
  if (job.edit) humid_ctx_set_option(k.ctx, "edit_distance", 1);
  if (P == 1) humid_ctx_set_option(k.ctx, "force_comm", 1);    // (HUMID_FORCE_SHARDED: the point is to run the transport)
  humid_comm cm;
  cm.user = &k;
  cm.rank = r;
  cm.world = P;
  cm.host_all_gather = [](void *user, const void *mine, uint64_t bytes, void *all) -> int {
    return ((Rank *)user)->host_all_gather_bytes(mine, bytes, all) ? 0 : -1;
  };
  cm.exchange = [](void *user, const void *d_send, const uint64_t *so, const uint64_t *sb, void *d_recv,
                   const uint64_t *ro, const uint64_t *rb, int, void *) -> int {
    Rank &k = *(Rank *)user;
    const unsigned P = k.g.P;
    return k.exchange(d_send, std::vector<uint64_t>(so, so + P), std::vector<uint64_t>(sb, sb + P), d_recv,
                      std::vector<uint64_t>(ro, ro + P), std::vector<uint64_t>(rb, rb + P)) ? 0 : -1;
  };
  humid_summary sum;
  humid_exchange_info info;
  std::memset(&sum, 0, sizeof sum);
  std::memset(&info, 0, sizeof info);
  const int rc = humid_dedup_run_exchange(k.ctx, &cm, d_w.as<uint64_t>(), d_f.as<uint8_t>(), n_local, job.word_nt, job.distance,
                                          job.method, d_cid.as<uint32_t>(), d_keep.as<uint8_t>(), &sum, &info);
  if (rc != HUMID_OK) {
    if (k.code == HUMID_OK) { k.code = rc; k.err = humid_last_error(k.ctx); }           // (a transport error keeps its own text)
    return false;
  }
  if (n_local) {
    STEP(k.hip_ok(hipMemcpyAsync(job.cluster_id + r0, d_cid.p, n_local * 4, hipMemcpyDeviceToHost, st), "hipMemcpyAsync (ids)"));
    STEP(k.hip_ok(hipMemcpyAsync(job.keep + r0, d_keep.p, n_local, hipMemcpyDeviceToHost, st), "hipMemcpyAsync (keep)"));
  }
  if (job.want_hist) {
    if (info.unique_local) {                                                            // counts.dat: leaf -> count
      std::vector<uint32_t> h(info.unique_local);
      STEP(k.hip_ok(hipMemcpyAsync(h.data(), info.d_unique_count, info.unique_local * 4, hipMemcpyDeviceToHost, st), "hipMemcpy (counts)"));
      STEP(k.hip_ok(hipStreamSynchronize(st), "hipStreamSynchronize"));
      std::map<uint64_t, uint64_t> m;
      for (uint32_t v : h) m[v]++;
      std::lock_guard<std::mutex> lk(g.mu);
      for (auto &kv : m) g.hist_counts[kv.first] += kv.second;
    }
    if (info.unique_local) {                                                            // neigh.dat: degree of every own leaf
      std::vector<uint32_t> dg(info.unique_local);
      STEP(k.hip_ok(hipMemcpyAsync(dg.data(), info.d_unique_degree, info.unique_local * 4, hipMemcpyDeviceToHost, st), "hipMemcpy (degrees)"));
      STEP(k.hip_ok(hipStreamSynchronize(st), "hipStreamSynchronize"));
      std::map<uint64_t, uint64_t> m;
      for (uint32_t v : dg) m[v]++;
      std::lock_guard<std::mutex> lk(g.mu);
      for (auto &kv : m) g.hist_neigh[kv.first] += kv.second;
    }
  }
  STEP(k.hip_ok(hipStreamSynchronize(st), "hipStreamSynchronize"));
  if (r == 0) job.sum = sum;
  return k.together();                                                                  // peers may still be copying from this rank's buffers
}

}  // namespace

struct ShardedSession::Impl {
  Group g;
  std::vector<std::thread> threads;
  std::thread starter;
  std::string early_error;
  int early_code = HUMID_OK;
  bool ran = false;
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
  g.comms = std::vector<std::atomic<ncclComm_t>>(n_ranks);
  for (auto &cm : g.comms) cm.store(nullptr);
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
    // (also when RCCL is the plan: the ranks fall back to peer copies if a communicator does not come up)
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
                        ShardedResult &out, bool edit) {
  Impl &m = *p_;
  Group &g = m.g;
  if (m.early_code != HUMID_OK) { out.error = m.early_error; return m.early_code; }
  if (m.ran) { out.error = "-g: a sharded session runs one job"; return HUMID_E_INVALID; }     // (its rank threads end with the job)
  m.ran = true;
  if (word_nt == 0 || word_nt > 64) { out.error = "-g: word length 1 .. 64"; return HUMID_E_UNSUPPORTED; }
  if (n_reads >= 0x7fffffffull * g.P) { out.error = "-g: more than 2^31-1 reads per rank"; return HUMID_E_OVERFLOW; }
  if (m.starter.joinable()) m.starter.join();
  out.ms_init = m.ms_init;
  Job job{words, filtered, n_reads, word_nt, distance, method, want_hist, edit, cluster_id, keep, {}};
  const auto t1 = std::chrono::steady_clock::now();
  g.post_job(&job);
  for (auto &t : m.threads) if (t.joinable()) t.join();
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
