// humid -- command-line host of the MI355X-native HUMID hot path.
//
// Keeps the reference's CLI and FastQ-in / FastQ-out contract
// (/root/reference/src/humid.cc:369-429): same flags, same log lines, same output names
// (<name>_dedup.<ext>, <name>_annotated.<ext>), same .dat statistics.  FastQ is streamed twice:
// pass 1 builds the packed words, libhumid_hip.so (include/humid_hip.h) does counts ->
// neighbours -> clusters on the GPU, pass 2 writes with the per-read (cluster_id, keep) arrays
// instead of trie.find() (src/humid.cc:223-231,276-277).
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>

#include <unistd.h>

#include <atomic>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <fstream>
#include <functional>
#include <iostream>
#include <string>
#include <vector>

#include "../../../include/humid_hip.h"
#include "fastq_io.hpp"
#include "fastq_mmap.hpp"
#include <iterator>
#include <memory>
#include <thread>

#include "fast_inflate.hpp"
#include "sharded.hpp"
#include "words.hpp"

using namespace humid_host;

namespace {

struct Args {
  size_t word_length = 24;     // -n
  size_t distance = 1;         // -m
  std::string log_name = "/dev/stderr";   // -l
  std::string dir_name = ".";  // -d
  bool stats = false;          // -s
  bool filter = true;          // -q flips
  bool annotate = false;       // -a
  bool edit = false;           // -e
  bool maximum = false;        // -x
  std::vector<std::string> files;
  std::string dump_words;      // --dump-words (development: stop after pass 1, no GPU)
  unsigned gpus = 1;           // -g (not in the reference): ranks the read set is sharded over, one GPU each
};

void usage(const char *argv0) {
  std::fprintf(stderr,
               "usage: %s [-n 24] [-m 1] [-l /dev/stderr] [-d .] [-s] [-q] [-a] [-e] [-x] [-g 1] files...\n"
               "Deduplicate a dataset.\n"
               "  -n  word length\n  -m  allowed mismatches\n  -l  log file name\n  -d  output directory\n"
               "  -s  calculate statistics\n  -q  write deduplicated FastQ files (flag turns it OFF)\n"
               "  -a  write annotated FastQ files\n  -e  use edit distance (Levenshtein neighbours)\n"
               "  -x  use maximum clustering method\n"
               "  -g  GPUs to shard the read set over (1..16; default 1 or $HUMID_GPUS; no -e beyond -m 1)\n",
               argv0);
}

bool parse(int argc, char **argv, Args &a) {
  for (int i = 1; i < argc; i++) {
    std::string t = argv[i];
    auto need = [&](const char *what) -> const char * {
      if (i + 1 >= argc) { std::fprintf(stderr, "humid: %s needs a value\n", what); return nullptr; }
      return argv[++i];
    };
    if (t == "-h" || t == "--help") { usage(argv[0]); std::exit(0); }
    else if (t == "-n") { const char *v = need("-n"); if (!v) return false; a.word_length = std::strtoull(v, nullptr, 10); }
    else if (t == "-m") { const char *v = need("-m"); if (!v) return false; a.distance = std::strtoull(v, nullptr, 10); }
    else if (t == "-l") { const char *v = need("-l"); if (!v) return false; a.log_name = v; }
    else if (t == "-d") { const char *v = need("-d"); if (!v) return false; a.dir_name = v; }
    else if (t == "-g") { const char *v = need("-g"); if (!v) return false; a.gpus = (unsigned)std::strtoul(v, nullptr, 10); }
    else if (t == "--dump-words") { const char *v = need("--dump-words"); if (!v) return false; a.dump_words = v; }
    else if (t == "-s") a.stats = !a.stats;
    else if (t == "-q") a.filter = !a.filter;
    else if (t == "-a") a.annotate = !a.annotate;
    else if (t == "-e") a.edit = !a.edit;
    else if (t == "-x") a.maximum = !a.maximum;
    else if (t.size() > 1 && t[0] == '-') { std::fprintf(stderr, "humid: unknown option %s\n", t.c_str()); return false; }
    else a.files.push_back(t);
  }
  if (a.files.empty()) { std::fprintf(stderr, "humid: no input files\n"); return false; }
  return true;
}

// src/log.cc:4-15
time_t start_message(std::ofstream &log, const char *msg) {
  log << msg << "... ";
  log.flush();
  return time(nullptr);
}
void end_message(std::ofstream &log, time_t start) {
  time_t seconds = (time_t)difftime(time(nullptr), start);
  log << "done. (" << seconds / 60 << 'm' << seconds % 60 << "s)\n";
  log.flush();
}

void make_dirs(const std::string &path) {   // std::filesystem::create_directories
  std::string cur;
  for (size_t i = 0; i <= path.size(); i++) {
    if (i == path.size() || path[i] == '/') {
      if (!cur.empty()) mkdir(cur.c_str(), 0777);
    }
    if (i < path.size()) cur.push_back(path[i]);
  }
}

// 1 = written, 0 = the library failed (humid_last_error says why), -1 = the file could not be written
int write_hist(humid_ctx *ctx, uint32_t which, const std::string &path) {
  uint64_t n = 0;
  if (humid_get_histogram(ctx, which, nullptr, nullptr, 0, &n) != HUMID_OK) return 0;
  std::vector<uint64_t> k(n ? n : 1), v(n ? n : 1);
  if (n && humid_get_histogram(ctx, which, k.data(), v.data(), n, &n) != HUMID_OK) return 0;
  std::ofstream out(path, std::ios::out | std::ios::binary);
  for (uint64_t i = 0; i < n; i++) out << k[i] << ' ' << v[i] << '\n';   // src/humid.cc:333-349
  out.close();
  return out.fail() ? -1 : 1;
}

// Plain (uncompressed) output of the fast path, written through a shared mapping of the output file:
// every worker copies its share of the records straight to their final place, so the page cache is
// filled by all cores at once (one writer through pwrite() is bound by a single core's copy rate,
// parallel pwrite()s into one file serialise on the inode lock).  size_of(i) = bytes record i
// contributes (0: not written), emit(i, q) writes them at q and returns the end.
// Returns 1 = written, 0 = not possible on this file (the caller takes the buffered writer), -1 = failed.
int write_mapped(const std::string &path, size_t n, unsigned threads, const std::function<size_t(size_t)> &size_of,
                 const std::function<char *(size_t, char *)> &emit) {
  if (threads == 0) threads = 1;
  const bool timing = getenv("HUMID_TIMING") != nullptr;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
    return std::chrono::duration<double, std::milli>(b - a).count();
  };
  const auto t_begin = now();
  std::vector<uint64_t> part(threads + 1, 0);
  parallel_ranges(n, threads, [&](size_t b, size_t e, unsigned w) {
    uint64_t t = 0;
    for (size_t i = b; i < e; i++) t += size_of(i);
    part[w + 1] = t;
  });
  const auto t_sized = now();
  const bool one = threads <= 1 || n < 4096;            // parallel_ranges ran everything as worker 0
  for (unsigned w = 0; w < threads; w++) part[w + 1] += part[w];
  const uint64_t total = part[threads];
  const int fd = ::open(path.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0666);
  if (fd < 0) return -1;
  if (total == 0) return ::close(fd) == 0 ? 1 : -1;
  // the blocks are reserved first, so a full disk is an error code here and not a SIGBUS later.
  // (Measured and rejected: extending the file sparsely and letting the workers' page faults
  // allocate it -- on tmpfs parallel faults into one file contend so badly that pass 2 takes 3x
  // longer than with the single-threaded allocation of posix_fallocate.)
  // (Also measured and rejected, end of round 3: reserving in 32 MB chunks on a thread of its own while the workers
  // copy behind it -- the allocation and the faults on the same file slow each other down: 100-190 ms for both
  // together against 45 + 50-120 one after the other.)
  if (posix_fallocate(fd, 0, (off_t)total) != 0) { ::close(fd); return 0; }      // e.g. a device: buffered writer
  const auto t_alloc = now();
  char *m = (char *)mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  if (m == (char *)MAP_FAILED) { ::close(fd); return 0; }
  parallel_ranges(n, threads, [&](size_t b, size_t e, unsigned w) {
    char *const q0 = m + (one ? 0 : part[w]);
    char *q = q0;
    // (measured and rejected, end of round 3: MADV_POPULATE_WRITE over the worker's range in front of the copy -- the
    // copy phase is bound by the 200 k write faults of a 0.8 GB file, 50-120 ms against ~45 ms of preallocation, and
    // populating the same pages in one call per worker was no faster: 115-130 ms)
    for (size_t i = b; i < e; i++) q = emit(i, q);
    // this worker's pages leave the address space here, in parallel (the data stays in the page cache:
    // the mapping is shared); the munmap below then has next to nothing to walk
    const uintptr_t lo = ((uintptr_t)q0 + 4095) & ~(uintptr_t)4095, hi = (uintptr_t)q & ~(uintptr_t)4095;
    if (hi > lo) madvise((void *)lo, hi - lo, MADV_DONTNEED);
  });
  const auto t_copied = now();
  const bool ok = munmap(m, total) == 0;
  const bool closed = ::close(fd) == 0;
  if (timing)
    std::fprintf(stderr, "[humid]   mapped write of %.2f GB with %u workers: sizes %.1f ms, preallocation %.1f ms, copy %.1f ms, unmap + close %.1f ms\n",
                 (double)total / 1e9, threads, ms(t_begin, t_sized), ms(t_sized, t_alloc), ms(t_alloc, t_copied), ms(t_copied, now()));
  return (closed && ok) ? 1 : -1;
}

}  // namespace

int main(int argc, char **argv) {
  // development aid (no GPU): --gunzip IN.gz OUT inflates through the host's own decoder only
  // (fast_inflate.hpp); exit code 3 = the decoder declined the file
  if (argc == 4 && std::string(argv[1]) == "--gunzip") {
    std::ifstream in(argv[2], std::ios::binary);
    if (!in) return 1;
    std::string z((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    struct Buf { std::vector<char> v; size_t limit; } buf;
    buf.limit = z.size() * 1100 + (1u << 20);            // DEFLATE cannot expand more than 1032 : 1
    if (buf.limit > ((size_t)8 << 30)) buf.limit = (size_t)8 << 30;
    auto grow = [](void *user, size_t min_cap, size_t *cap) -> char * {
      Buf *b = (Buf *)user;
      if (min_cap > b->limit) return nullptr;
      if (b->v.size() < min_cap) b->v.resize(min_cap);
      *cap = b->v.size();
      return b->v.data();
    };
    size_t n = 0;
    const auto t0 = std::chrono::steady_clock::now();
    const bool ok = humid_host::fast_gunzip((const uint8_t *)z.data(), z.size(), grow, &buf, &n, host_threads());
    if (getenv("HUMID_TIMING"))
      std::fprintf(stderr, "fast_gunzip: %zu -> %zu bytes in %.3f s\n", z.size(), n,
                   std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    if (!ok) return 3;
    std::ofstream out(argv[3], std::ios::binary);
    out.write(buf.v.data(), (std::streamsize)n);
    return out ? 0 : 1;
  }
  if (getenv("HUMID_TIMING")) {
    timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    std::fprintf(stderr, "[humid] main() entered at %.6f (epoch s)\n", (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec);
  }
  Args a;
  if (!parse(argc, argv, a)) { usage(argv[0]); return 2; }
  if (a.word_length == 0 || a.word_length > 64) {
    std::fprintf(stderr, "humid: word length %zu is not supported by the HIP path (1..64)\n", a.word_length);
    return 2;
  }
  if (a.gpus == 1 && getenv("HUMID_GPUS")) a.gpus = (unsigned)std::strtoul(getenv("HUMID_GPUS"), nullptr, 10);
  if (a.gpus < 1 || a.gpus > 16) {
    std::fprintf(stderr, "humid: -g takes 1 .. 16 GPUs\n");
    return 2;
  }
  // HUMID_FORCE_SHARDED=1: the rank orchestration also for -g 1 (one rank; exercises the transport)
  const bool sharded = a.gpus > 1 || getenv("HUMID_FORCE_SHARDED") != nullptr;
  std::ofstream log(a.log_name.c_str(), std::ios::out | std::ios::binary);

  // The HIP runtime and the context come up (a few hundred ms) while pass 1 parses the files.
  humid_ctx *ctx = nullptr;
  int ctx_rc = HUMID_OK;
  std::string ctx_err;
  struct Joiner {
    std::thread th;
    ~Joiner() { if (th.joinable()) th.join(); }
  } ctx_init;
  // ... and as soon as the number of reads is known (files indexed) it takes ONE slab of device
  // memory for the whole run (humid_ctx_reserve) while pass 1 is still gathering words.
  std::atomic<long long> n_known{-1};                   // -1: not yet; 0: no slab wanted
  uint8_t *pinned = nullptr;                            // written by the init thread, read after its join
  uint64_t pin_bytes = 0;
  // -g: the ranks (threads with a context, a stream and a communicator each) come up the same way
  std::unique_ptr<ShardedSession> ranks;
  if (a.dump_words.empty() && sharded) ranks.reset(new ShardedSession(a.gpus));
  if (a.dump_words.empty() && !sharded)
    ctx_init.th = std::thread([&] {
      const auto ti = std::chrono::steady_clock::now();
      auto since = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - ti).count(); };
      ctx_rc = humid_ctx_create(&ctx, -1, nullptr);
      if (ctx_rc != HUMID_OK) { ctx_err = humid_last_error(nullptr); return; }
      const double t_ctx = since();
      long long n;
      while ((n = n_known.load(std::memory_order_acquire)) < 0) std::this_thread::yield();
      const double t_wait = since();
      double t_pin = t_wait, t_slab = t_wait;
      if (n > 0) {
        // page-locked staging for what crosses PCIe: packed words + flags in, cluster ids + keep flags out
        const uint64_t wb = (uint64_t)n * (a.word_length > 32 ? 16 : 8);
        pin_bytes = wb + (uint64_t)n + (uint64_t)n * 4 + (uint64_t)n + 64;
        std::thread pin_thread;                            // the two take ~30 ms and ~40 ms: side by side
        if (getenv("HUMID_NO_PINNED") == nullptr)
          pin_thread = std::thread([&] { pinned = (uint8_t *)humid_host_alloc(pin_bytes); t_pin = since(); });
        if (getenv("HUMID_NO_SLAB") == nullptr)
          humid_ctx_reserve(ctx, (uint64_t)n, (uint32_t)a.word_length);   // one slab + the code object loaded
        t_slab = since();
        if (pin_thread.joinable()) pin_thread.join();
      }
      if (getenv("HUMID_TIMING"))
        std::fprintf(stderr, "[humid]   init thread: context %.3f s, read count known %.3f s, page-locked staging %.3f s, slab + code object %.3f s\n",
                     t_ctx, t_wait, t_pin, t_slab);
    });
  struct Release {                                      // every early return lets the thread go on
    std::atomic<long long> &n;
    ~Release() { long long e = -1; n.compare_exchange_strong(e, 0); }
  } release_init{n_known};
  const auto t_start = std::chrono::steady_clock::now();
  auto phase = [&](const char *what) {                 // HUMID_TIMING=1: wall time of the host phases
    if (getenv("HUMID_TIMING"))
      std::fprintf(stderr, "[humid] %-28s %.3f s\n", what,
                   std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count());
  };

  // ---- preCompute (src/humid.cc:38-59): UMI size from the first header of the first file ----
  size_t first_umi = 0;
  {
    FastqReader peek(a.files.front());
    if (!peek.ok()) { std::fprintf(stderr, "humid: cannot open %s\n", a.files.front().c_str()); return 1; }
    FastqRecord r;
    if (peek.read(r)) first_umi = header_umi(r.name).size();   // empty file: no UMI (the reference crashes)
  }
  WordPlan plan = make_plan(first_umi, a.files.size(), a.word_length);
  time_t t = start_message(log, "Determing nucleotides to take");
  end_message(log, t);
  log << "  header: " << plan.header_umi;
  for (size_t i = 0; i < a.files.size(); i++) log << "\n  " << a.files[i] << ": " << plan.take[i];
  log << "\n";

  // ---- fast path: plain canonical FastQ files are mapped and indexed on all cores ----
  const unsigned threads = host_threads();
  std::vector<MappedFastq> maps(a.files.size());
  bool fast = a.files.size() <= 64 && getenv("HUMID_HOST_SLOW") == nullptr;
  if (fast) {
    // one opener per file (a gzip file is inflated by one zlib stream); the indexing inside uses
    // the remaining workers
    const size_t nf = a.files.size();
    const size_t budget = retain_budget_bytes() / (nf ? nf : 1);
    const unsigned per = threads / (unsigned)nf ? threads / (unsigned)nf : 1;
    std::vector<char> okf(nf, 0);
    std::vector<std::thread> openers;
    for (size_t f = 0; f < nf; f++)
      openers.emplace_back([&, f] { okf[f] = maps[f].open(a.files[f], per, budget) ? 1 : 0; });
    for (auto &th : openers) th.join();
    for (size_t f = 0; f < nf; f++) fast = fast && okf[f];
  }

  phase("files opened / indexed");
  // ---- pass 1: readData (src/humid.cc:89-100) ----
  t = start_message(log, "Reading data");
  const size_t wpr = a.word_length > 32 ? 2 : 1;   // uint64 per word (include/humid_hip.h)
  std::vector<uint64_t> words;
  std::vector<uint8_t> filtered;
  std::vector<uint8_t> bases;                      // device-side packing: word_length raw symbols per record
  // HUMID_DEVICE_PACK=1: the host only gathers the raw symbols and the GPU packs them
  // (humid_dedup_run_bases).  Not the default: 24 raw bytes per read instead of 9 packed ones cross
  // PCIe, which costs more than the host's packing saves (profiles/r02d_cli_e2e.txt).
  const bool device_pack = fast && a.dump_words.empty() && !sharded && getenv("HUMID_DEVICE_PACK") != nullptr;
  uint64_t n_records = 0;
  if (fast) {
    size_t n = maps[0].records();
    for (auto &m : maps) n = m.records() < n ? m.records() : n;     // stop at the shortest file
    n_records = n;
    n_known.store((long long)n, std::memory_order_release);
    const size_t nf = maps.size();
    if (device_pack) {
      bases.resize(n * a.word_length);
      parallel_ranges(n, threads, [&](size_t b, size_t e, unsigned) {
        std::string_view seqs[64], name0, nm, st, ql;
        for (size_t i = b; i < e; i++) {
          for (size_t f = 0; f < nf; f++) {
            maps[f].lines(i, nm, seqs[f], st, ql);
            if (f == 0) name0 = nm;
          }
          gather_bases(name0, seqs, nf, plan, &bases[i * a.word_length]);
        }
      });
    } else {
    words.resize(n * wpr);
    filtered.resize(n);
    parallel_ranges(n, threads, [&](size_t b, size_t e, unsigned) {
      std::string_view seqs[64], name0, nm, st, ql;
      for (size_t i = b; i < e; i++) {
        for (size_t f = 0; f < nf; f++) {
          maps[f].lines(i, nm, seqs[f], st, ql);
          if (f == 0) name0 = nm;
        }
        if (wpr == 2) {
          filtered[i] = make_word_wide(name0, seqs, nf, plan, &words[2 * i]) ? 1 : 0;
        } else {
          uint64_t w;
          filtered[i] = make_word(name0, seqs, nf, plan, w) ? 1 : 0;
          words[i] = w;
        }
      }
    });
    }
  } else {
    n_known.store(0, std::memory_order_release);
    MultiReader in(a.files);
    if (!in.ok()) { std::fprintf(stderr, "humid: cannot open %s\n", in.bad_file().c_str()); return 1; }
    std::vector<FastqRecord> recs;
    while (in.next(recs)) {
      uint64_t w[2];
      bool f = wpr == 2 ? make_word_wide(recs, plan, w) : make_word(recs, plan, w[0]);
      words.push_back(w[0]);
      if (wpr == 2) words.push_back(w[1]);
      filtered.push_back(f ? 1 : 0);
    }
  }
  end_message(log, t);
  const uint64_t N = device_pack ? n_records : filtered.size();

  if (!a.dump_words.empty()) {   // development aid: host-side parsing can be checked without a GPU
    std::ofstream out(a.dump_words, std::ios::out | std::ios::binary);
    out.write((const char *)&N, 8);
    out.write((const char *)words.data(), (std::streamsize)(N * 8 * wpr));
    out.write((const char *)filtered.data(), (std::streamsize)N);
    return 0;
  }

  // ---- the hot path on the GPU ----
  phase("pass 1 done");
  if (ctx_init.th.joinable()) ctx_init.th.join();
  if (!sharded && (ctx_rc != HUMID_OK || !ctx)) {
    std::fprintf(stderr, "humid: %s\n", ctx_err.c_str());
    return 1;
  }
  phase("context ready");
  // results: in the page-locked staging when there is one (pass 2 reads them there), else in vectors
  std::vector<uint32_t> cid_vec;
  std::vector<uint8_t> keep_vec;
  uint32_t *cluster_id = nullptr;
  uint8_t *keep = nullptr;
  const uint64_t *run_words = words.data();
  const uint8_t *run_filt = filtered.data();
  const bool staged = pinned != nullptr && N > 0 && !device_pack && (uint64_t)n_known.load() == N;
  if (staged) {
    const uint64_t wb = N * 8 * wpr;
    uint8_t *p_words = pinned, *p_filt = pinned + wb;
    cluster_id = (uint32_t *)(pinned + ((wb + N + 15) & ~(uint64_t)15));
    keep = (uint8_t *)(cluster_id + N);
    parallel_ranges(N, threads, [&](size_t b, size_t e, unsigned) {
      memcpy(p_words + b * 8 * wpr, (const uint8_t *)words.data() + b * 8 * wpr, (e - b) * 8 * wpr);
      memcpy(p_filt + b, filtered.data() + b, e - b);
    });
    run_words = (const uint64_t *)p_words;
    run_filt = p_filt;
    phase("words in page-locked staging");
  } else {
    cid_vec.resize(N ? N : 1);
    keep_vec.resize(N ? N : 1);
    cluster_id = cid_vec.data();
    keep = keep_vec.data();
  }
  humid_summary sum;
  std::memset(&sum, 0, sizeof sum);
  if (a.edit) humid_ctx_set_option(ctx, "edit_distance", 1);
  t = start_message(log, a.edit ? "Calculating neighbours using Levenshtein distance"     // src/humid.cc:142
                                : "Calculating neighbours using Hamming distance");
  const uint32_t method = a.maximum ? HUMID_METHOD_MAXIMUM : HUMID_METHOD_DIRECTIONAL;
  ShardedResult shr;
  int rc;
  if (sharded) {
    // the read set in input-order shards, one rank (thread + context + GPU) per shard: sharded.cpp
    rc = ranks->run(run_words, run_filt, N, (uint32_t)a.word_length, (uint32_t)a.distance, method, a.stats, cluster_id, keep, shr,
                    a.edit);
    sum = shr.sum;
    if (rc == HUMID_OK && getenv("HUMID_TIMING"))
      std::fprintf(stderr, "[humid]   %u ranks, bulk data by %s: set-up %.1f ms, ranks %.1f ms\n", a.gpus,
                   shr.comm.c_str(), shr.ms_init, shr.ms_run);
  } else
  rc = device_pack
               ? humid_dedup_run_bases(ctx, bases.data(), N, (uint32_t)a.word_length, (uint32_t)a.distance, method,
                                       cluster_id, keep, &sum)
               : humid_dedup_run(ctx, run_words, run_filt, N, (uint32_t)a.word_length,
                                 (uint32_t)a.distance, method, cluster_id, keep, &sum);
  if (rc != HUMID_OK) {
    log << "failed.\n";
    std::fprintf(stderr, "humid: %s\n", sharded ? shr.error.c_str() : humid_last_error(ctx));
    humid_ctx_destroy(ctx);
    return 1;
  }
  end_message(log, t);
  phase("device path done");
  if (getenv("HUMID_TIMING"))
    std::fprintf(stderr, "[humid]   of which on the device: upload%s %.1f ms, hot path %.2f ms, download %.1f ms\n",
                 device_pack ? " + packing" : "", sum.ms_h2d, sum.ms_total, sum.ms_d2h);
  t = start_message(log, a.maximum ? "Calculating maximum clusters" : "Calculating directional clusters");
  end_message(log, t);
  std::vector<uint64_t>().swap(words);
  std::vector<uint8_t>().swap(bases);

  make_dirs(a.dir_name);

  // ---- pass 2: writeFiltered (src/humid.cc:203-241) / writeAnnotated (:251-292) ----
  if (a.filter || a.annotate) {
    time_t tf = 0, ta = 0;
    if (a.filter) tf = start_message(log, "Writing filtered results");
    std::vector<FastqWriter *> dedup, annot;
    for (const std::string &f : a.files) {
      // fast path: .gz outputs are written as gzip members compressed on all cores
      if (a.filter) dedup.push_back(new FastqWriter(make_file_name(f, a.dir_name, "dedup"), fast));
      if (a.annotate) annot.push_back(new FastqWriter(make_file_name(f, a.dir_name, "annotated"), fast));
    }
    bool ok = true;
    for (FastqWriter *w : dedup) ok = ok && w->ok();
    for (FastqWriter *w : annot) ok = ok && w->ok();
    if (!ok) { std::fprintf(stderr, "humid: cannot create output files in %s\n", a.dir_name.c_str()); return 1; }
    // plain outputs of the fast path go through a shared mapping (write_mapped); what it cannot take
    // (gzip outputs, files that cannot be preallocated) goes through the batch loop below
    std::vector<char> done_dedup(a.files.size(), 0), done_annot(a.files.size(), 0);
    bool mapped_failed = false;
    if (fast && getenv("HUMID_NO_MAPPED_WRITE") == nullptr) {
      auto digits = [](uint32_t v) { size_t d = 1; while (v >= 10) { v /= 10; d++; } return d; };
      // the files of a pair are written side by side (page-cache filling scales per file), each with
      // its share of the workers
      const unsigned per_file = threads / (unsigned)maps.size() ? threads / (unsigned)maps.size() : 1;
      std::vector<char> failed_f(maps.size(), 0);
      std::vector<std::thread> file_workers;
      for (size_t f = 0; f < maps.size(); f++) file_workers.emplace_back([&, f] {
        const unsigned threads = per_file;               // (shadows: this file's share)
        bool mapped_failed = false;
        if (a.filter && !dedup[f]->gz_members()) {
          const std::string path = make_file_name(a.files[f], a.dir_name, "dedup");
          const int r = write_mapped(path, N, threads,
              [&](size_t i) { return keep[i] ? maps[f].raw(i).size() : (size_t)0; },
              [&](size_t i, char *q) {
                if (!keep[i]) return q;
                const std::string_view rr = maps[f].raw(i);
                memcpy(q, rr.data(), rr.size());
                return q + rr.size();
              });
          if (r > 0) done_dedup[f] = 1;
          if (r < 0) mapped_failed = true;
        }
        if (a.annotate && !annot[f]->gz_members()) {
          // annotated record = header + ':' + cluster id + the rest of the record verbatim (src/humid.cc:281)
          const std::string path = make_file_name(a.files[f], a.dir_name, "annotated");
          const int r = write_mapped(path, N, threads,
              [&](size_t i) { return maps[f].raw(i).size() + 1 + digits(cluster_id[i]); },
              [&](size_t i, char *q) {
                const std::string_view rr = maps[f].raw(i);
                const char *nl = (const char *)memchr(rr.data(), '\n', rr.size());
                const size_t hl = nl ? (size_t)(nl - rr.data()) : rr.size();
                memcpy(q, rr.data(), hl);
                q += hl;
                *q++ = ':';
                char dig[10];
                unsigned nd = 0;
                uint32_t v = cluster_id[i];
                do { dig[nd++] = (char)('0' + v % 10); v /= 10; } while (v);
                while (nd) *q++ = dig[--nd];
                memcpy(q, rr.data() + hl, rr.size() - hl);
                return q + (rr.size() - hl);
              });
          if (r > 0) done_annot[f] = 1;
          if (r < 0) mapped_failed = true;
        }
        failed_f[f] = mapped_failed ? 1 : 0;
      });
      for (auto &th : file_workers) th.join();
      for (char x : failed_f) mapped_failed = mapped_failed || x;
    }
    if (fast) {
      // records come straight from the mappings; batches are formatted on all cores and written
      // in order
      const size_t nf = maps.size();
      const size_t batch = 1u << 20;
      std::vector<std::string> bufs(threads), zbufs(threads);
      for (size_t b0 = 0; b0 < N; b0 += batch) {
        const size_t b1 = b0 + batch < N ? b0 + batch : N;
        for (size_t f = 0; f < nf; f++) {
          for (int what = 0; what < 2; what++) {           // 0 = dedup, 1 = annotated
            if ((what == 0 && !a.filter) || (what == 1 && !a.annotate)) continue;
            if ((what == 0 && done_dedup[f]) || (what == 1 && done_annot[f])) continue;   // written through the mapping
            FastqWriter *wr = what == 0 ? dedup[f] : annot[f];
            const bool zip = wr->gz_members();
            parallel_ranges(b1 - b0, threads, [&](size_t rb, size_t re, unsigned w) {
              std::string &o = bufs[w];
              o.clear();
              if (what == 0) {
                for (size_t i = b0 + rb; i < b0 + re; i++)
                  if (keep[i]) { std::string_view r = maps[f].raw(i); o.append(r.data(), r.size()); }
              } else if (re > rb) {
                // annotated record = header + ':' + cluster id + the rest of the record verbatim
                // (src/humid.cc:281); written with pointer arithmetic into a buffer sized up front
                const size_t first = b0 + rb, last = b0 + re;
                const size_t in_bytes = (size_t)(maps[f].rec_off[last] - maps[f].rec_off[first]);
                o.resize(in_bytes + 11 * (last - first));
                char *q = &o[0];
                for (size_t i = first; i < last; i++) {
                  const std::string_view r = maps[f].raw(i);
                  const char *nl = (const char *)memchr(r.data(), '\n', r.size());
                  const size_t hl = nl ? (size_t)(nl - r.data()) : r.size();
                  memcpy(q, r.data(), hl);
                  q += hl;
                  *q++ = ':';
                  char dig[10];
                  unsigned nd = 0;
                  uint32_t v = cluster_id[i];
                  do { dig[nd++] = (char)('0' + v % 10); v /= 10; } while (v);
                  while (nd) *q++ = dig[--nd];
                  memcpy(q, r.data() + hl, r.size() - hl);
                  q += r.size() - hl;
                }
                o.resize((size_t)(q - o.data()));
              }
              if (zip) {
                zbufs[w].clear();
                if (!o.empty()) FastqWriter::compress_member(o.data(), o.size(), zbufs[w]);
              }
            });
            const unsigned used = (threads <= 1 || b1 - b0 < 4096) ? 1 : threads;
            wr->write_parts(zip ? zbufs : bufs, used);       // every part at its final offset, in parallel
          }
        }
      }
    } else {
    MultiReader in(a.files);
    std::vector<FastqRecord> recs;
    std::string s;
    uint64_t i = 0;
    while (i < N && in.next(recs)) {
      if (a.filter && keep[i]) {
        for (size_t f = 0; f < recs.size(); f++) {
          s.clear();
          recs[f].append_to(s);
          dedup[f]->write(s.data(), s.size());
        }
      }
      if (a.annotate) {
        const std::string tag = ":" + std::to_string(cluster_id[i]);   // src/humid.cc:281
        for (size_t f = 0; f < recs.size(); f++) {
          recs[f].name += tag;
          s.clear();
          recs[f].append_to(s);
          annot[f]->write(s.data(), s.size());
        }
      }
      i++;
    }
    }
    // a short or failed write (disk full, I/O error) must not end in exit code 0
    bool wrote_ok = !mapped_failed;
    // (a FastqWriter whose file went through the mapping wrote nothing: it must not truncate the file)
    for (size_t f = 0; f < dedup.size(); f++) { if (!done_dedup[f]) wrote_ok = dedup[f]->close() && wrote_ok; delete dedup[f]; }
    for (size_t f = 0; f < annot.size(); f++) { if (!done_annot[f]) wrote_ok = annot[f]->close() && wrote_ok; delete annot[f]; }
    if (!wrote_ok) {
      log << "failed.\n";
      std::fprintf(stderr, "humid: writing the output FastQ files in %s failed (disk full or I/O error): "
                           "the outputs are incomplete\n", a.dir_name.c_str());
      humid_ctx_destroy(ctx);
      return 1;
    }
    if (a.filter) end_message(log, tf);
    if (a.annotate) { ta = start_message(log, "Writing annotated results"); end_message(log, ta); }
  }

  if (fast && getenv("HUMID_SLOW_EXIT") == nullptr)
    for (auto &m : maps) m.drop_pages(threads);          // the inputs are not read again
  phase("pass 2 done");
  // ---- statistics (src/humid.cc:301-357, src/cluster.cc:89-95) ----
  if (a.stats) {
    t = start_message(log, "Calculating count and neighbour stats");
    const char *names[3] = {"/counts.dat", "/neigh.dat", "/clusters.dat"};
    int ok = 1;
    for (uint32_t which = 0; which < 3 && ok == 1; which++) {
      if (!sharded) { ok = write_hist(ctx, which, a.dir_name + names[which]); continue; }
      std::ofstream out(a.dir_name + names[which], std::ios::out | std::ios::binary);   // gathered by the ranks
      for (auto &kv : shr.hist[which]) out << kv.first << ' ' << kv.second << '\n';
      out.close();
      if (out.fail()) ok = -1;
    }
    end_message(log, t);
    if (ok == 0) { std::fprintf(stderr, "humid: %s\n", humid_last_error(ctx)); humid_ctx_destroy(ctx); return 1; }
    std::ofstream out(a.dir_name + "/stats.dat", std::ios::out | std::ios::binary);
    out << "total: " << sum.total << '\n';
    out << "usable: " << sum.usable << '\n';
    out << "unique: " << sum.unique << '\n';
    out << "clusters: " << sum.clusters << '\n';
    out.close();
    if (ok < 0 || out.fail()) {
      std::fprintf(stderr, "humid: writing the statistics files in %s failed\n", a.dir_name.c_str());
      humid_ctx_destroy(ctx);
      return 1;
    }
  }
  log.close();
  if (log.fail()) std::fprintf(stderr, "humid: writing the log %s failed\n", a.log_name.c_str());
  phase("outputs closed");
  if (getenv("HUMID_SLOW_EXIT")) humid_host_free(pinned);     // (the quick exit below leaves it to the kernel)
  humid_ctx_destroy(ctx);
  phase("context destroyed");
  // every output is closed: leave without the static destructors of the HIP runtime and without
  // unmapping the inputs page by page (HUMID_SLOW_EXIT=1 keeps the ordinary exit)
  if (getenv("HUMID_TIMING")) {                          // against the caller's clock: start-up and tear-down outside main()
    timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    std::fprintf(stderr, "[humid] leaving main() at %.6f (epoch s)\n", (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec);
  }
  if (getenv("HUMID_SLOW_EXIT") == nullptr) {
    std::fflush(nullptr);
    _exit(0);
  }
  return 0;
}
