// humid -- command-line host of the MI355X-native HUMID hot path.
//
// Keeps the reference's CLI and FastQ-in / FastQ-out contract
// (/root/reference/src/humid.cc:369-429): same flags, same log lines, same output names
// (<name>_dedup.<ext>, <name>_annotated.<ext>), same .dat statistics.  FastQ is streamed twice:
// pass 1 builds the packed words, libhumid_hip.so (include/humid_hip.h) does counts ->
// neighbours -> clusters on the GPU, pass 2 writes with the per-read (cluster_id, keep) arrays
// instead of trie.find() (src/humid.cc:223-231,276-277).
#include <sys/stat.h>

#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "../../../include/humid_hip.h"
#include "fastq_io.hpp"
#include "fastq_mmap.hpp"
#include <iterator>
#include <thread>

#include "fast_inflate.hpp"
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
};

void usage(const char *argv0) {
  std::fprintf(stderr,
               "usage: %s [-n 24] [-m 1] [-l /dev/stderr] [-d .] [-s] [-q] [-a] [-e] [-x] files...\n"
               "Deduplicate a dataset.\n"
               "  -n  word length\n  -m  allowed mismatches\n  -l  log file name\n  -d  output directory\n"
               "  -s  calculate statistics\n  -q  write deduplicated FastQ files (flag turns it OFF)\n"
               "  -a  write annotated FastQ files\n  -e  use edit distance (Levenshtein neighbours; -m <= 3)\n"
               "  -x  use maximum clustering method\n",
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
  Args a;
  if (!parse(argc, argv, a)) { usage(argv[0]); return 2; }
  if (a.edit && a.distance > 3) {
    std::fprintf(stderr, "humid: edit distance (-e) is supported for -m <= 3 by the HIP path\n");
    return 2;
  }
  if (a.word_length == 0 || a.word_length > 64) {
    std::fprintf(stderr, "humid: word length %zu is not supported by the HIP path (1..64)\n", a.word_length);
    return 2;
  }
  std::ofstream log(a.log_name.c_str(), std::ios::out | std::ios::binary);

  // The HIP runtime and the context come up (a few hundred ms) while pass 1 parses the files.
  humid_ctx *ctx = nullptr;
  int ctx_rc = HUMID_OK;
  std::string ctx_err;
  struct Joiner {
    std::thread th;
    ~Joiner() { if (th.joinable()) th.join(); }
  } ctx_init;
  if (a.dump_words.empty())
    ctx_init.th = std::thread([&] {
      ctx_rc = humid_ctx_create(&ctx, -1, nullptr);
      if (ctx_rc != HUMID_OK) ctx_err = humid_last_error(nullptr);
    });
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
  if (fast) {
    size_t n = maps[0].records();
    for (auto &m : maps) n = m.records() < n ? m.records() : n;     // stop at the shortest file
    words.resize(n * wpr);
    filtered.resize(n);
    const size_t nf = maps.size();
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
  } else {
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
  const uint64_t N = filtered.size();

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
  if (ctx_rc != HUMID_OK || !ctx) {
    std::fprintf(stderr, "humid: %s\n", ctx_err.c_str());
    return 1;
  }
  phase("context ready");
  std::vector<uint32_t> cluster_id(N ? N : 1);
  std::vector<uint8_t> keep(N ? N : 1);
  humid_summary sum;
  std::memset(&sum, 0, sizeof sum);
  if (a.edit) humid_ctx_set_option(ctx, "edit_distance", 1);
  t = start_message(log, a.edit ? "Calculating neighbours using Levenshtein distance"     // src/humid.cc:142
                                : "Calculating neighbours using Hamming distance");
  int rc = humid_dedup_run(ctx, words.data(), filtered.data(), N, (uint32_t)a.word_length,
                           (uint32_t)a.distance, a.maximum ? HUMID_METHOD_MAXIMUM : HUMID_METHOD_DIRECTIONAL,
                           cluster_id.data(), keep.data(), &sum);
  if (rc != HUMID_OK) {
    log << "failed.\n";
    std::fprintf(stderr, "humid: %s\n", humid_last_error(ctx));
    humid_ctx_destroy(ctx);
    return 1;
  }
  end_message(log, t);
  phase("device path done");
  t = start_message(log, a.maximum ? "Calculating maximum clusters" : "Calculating directional clusters");
  end_message(log, t);
  std::vector<uint64_t>().swap(words);

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
    bool wrote_ok = true;
    for (FastqWriter *w : dedup) { wrote_ok = w->close() && wrote_ok; delete w; }
    for (FastqWriter *w : annot) { wrote_ok = w->close() && wrote_ok; delete w; }
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

  phase("pass 2 done");
  // ---- statistics (src/humid.cc:301-357, src/cluster.cc:89-95) ----
  if (a.stats) {
    t = start_message(log, "Calculating count and neighbour stats");
    const char *names[3] = {"/counts.dat", "/neigh.dat", "/clusters.dat"};
    int ok = 1;
    for (uint32_t which = 0; which < 3 && ok == 1; which++) ok = write_hist(ctx, which, a.dir_name + names[which]);
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
  humid_ctx_destroy(ctx);
  phase("context destroyed");
  // every output is closed: leave without the static destructors of the HIP runtime and without
  // unmapping the inputs page by page (HUMID_SLOW_EXIT=1 keeps the ordinary exit)
  if (getenv("HUMID_SLOW_EXIT") == nullptr) {
    std::fflush(nullptr);
    _exit(0);
  }
  return 0;
}
