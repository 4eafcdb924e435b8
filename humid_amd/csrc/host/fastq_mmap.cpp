#include "fastq_mmap.hpp"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include "fast_inflate.hpp"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

unsigned host_threads() {
  unsigned t = 0;
  if (const char *e = getenv("HUMID_THREADS")) { t = (unsigned)atoi(e); if (t) return t > 512 ? 512 : t; }   // as asked for
  if (t == 0) t = std::thread::hardware_concurrency();
  if (t == 0) t = 1;
  if (t > 64) t = 64;
  // a CPU bandwidth limit on the control group (containers: "1600000 100000" = 16 CPUs on a 256-CPU host): threads
  // beyond it do not add CPU time, they use the period's share up early and are then all stopped for the rest of the
  // period (measured on such a box, end of round 3: 16 / 32 / 64 / 128 threads give the same 0.45-0.58 s end to end).
  // Twice the limit: the workers of both passes spend part of their time in page faults.
  if (FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
    char quota[32] = {0};
    unsigned long period = 0;
    if (std::fscanf(f, "%31s %lu", quota, &period) == 2 && period > 0 && std::strcmp(quota, "max") != 0) {
      const unsigned long cpus = (std::strtoul(quota, nullptr, 10) + period - 1) / period;
      if (cpus >= 1 && 2 * cpus < t) t = (unsigned)(2 * cpus < 4 ? 4 : 2 * cpus);
    }
    std::fclose(f);
  }
  return t;
}

size_t retain_budget_bytes() {
  if (const char *e = getenv("HUMID_RETAIN_GB")) return (size_t)(atof(e) * 1073741824.0);
  const long pages = sysconf(_SC_PHYS_PAGES), psz = sysconf(_SC_PAGE_SIZE);
  if (pages <= 0 || psz <= 0) return (size_t)8 << 30;
  return (size_t)pages * (size_t)psz / 3;
}

void parallel_ranges(size_t n, unsigned threads, const std::function<void(size_t, size_t, unsigned)> &fn) {
  if (threads <= 1 || n < 4096) { fn(0, n, 0); return; }
  std::vector<std::thread> pool;
  for (unsigned w = 0; w < threads; w++) {
    size_t b = n * w / threads, e = n * (w + 1) / threads;
    pool.emplace_back([=, &fn] { fn(b, e, w); });
  }
  for (auto &t : pool) t.join();
}

void MappedFastq::drop_pages(unsigned threads) const {
  if (!data || size < (1u << 20)) return;
  const size_t page = 4096;
  const uintptr_t lo = ((uintptr_t)data + page - 1) & ~(uintptr_t)(page - 1), hi = ((uintptr_t)data + size) & ~(uintptr_t)(page - 1);
  if (hi <= lo) return;
  const size_t pages = (hi - lo) / page;
  parallel_ranges(pages, threads, [&](size_t b, size_t e, unsigned) {
    if (e > b) madvise((void *)(lo + b * page), (e - b) * page, MADV_DONTNEED);
  });
}

MappedFastq::~MappedFastq() {
  if (data) munmap((void *)data, size);
  if (fd_ >= 0) close(fd_);
}

static inline const char *next_line(const char *p, const char *end) {
  const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
  return nl ? nl + 1 : end;
}

// gzip (one or more members) -> an anonymous mapping that replaces data/size; false when the
// stream is damaged or the inflated size would exceed max_inflated
namespace {
struct AnonBuf {             // growing anonymous mapping (mremap: no copy when it can extend in place)
  char *p = nullptr;
  size_t cap = 0, limit = 0;
};
char *grow_anon(void *user, size_t min_cap, size_t *capacity) {
  AnonBuf *b = (AnonBuf *)user;
  if (min_cap <= b->cap) { *capacity = b->cap; return b->p; }
  if (min_cap > b->limit) return nullptr;
  void *m = b->p ? mremap(b->p, b->cap, min_cap, MREMAP_MAYMOVE)
                 : mmap(nullptr, min_cap, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
  if (m == MAP_FAILED) return nullptr;
  b->p = (char *)m;
  b->cap = min_cap;
  *capacity = min_cap;
  return b->p;
}
}  // namespace

bool MappedFastq::inflate_all(const char *z, size_t zn, size_t max_inflated) {
  if (getenv("HUMID_ZLIB_INFLATE") == nullptr) {
    // the host's own decoder first (fast_inflate.hpp); zlib below takes over if it declines
    AnonBuf b;
    b.limit = max_inflated;
    size_t n = 0;
    if (humid_host::fast_gunzip((const uint8_t *)z, zn, grow_anon, &b, &n, host_threads())) {
      if (n == 0) { if (b.p) munmap(b.p, b.cap); data = nullptr; size = 0; return true; }
      const size_t page = (size_t)sysconf(_SC_PAGE_SIZE);
      const size_t keep = (n + page - 1) / page * page;
      if (keep < b.cap) munmap(b.p + keep, b.cap - keep);
      data = b.p;
      size = n;
      return true;
    }
    if (b.p) munmap(b.p, b.cap);
  }
  size_t cap = zn * 4 + (1u << 20);
  if (cap > max_inflated) cap = max_inflated;
  if (cap < (1u << 16)) cap = 1u << 16;
  char *out = (char *)mmap(nullptr, cap, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
  if (out == MAP_FAILED) return false;
  size_t n_out = 0, n_in = 0;
  z_stream zs;
  memset(&zs, 0, sizeof zs);
  if (inflateInit2(&zs, 15 + 32) != Z_OK) { munmap(out, cap); return false; }
  bool good = true;
  while (good) {
    if (n_out == cap) {                                   // grow by half (the mapping may move)
      size_t want = cap + cap / 2;
      if (want > max_inflated) want = max_inflated;
      if (want <= cap) { good = false; break; }           // over the retention bound
      void *m2 = mremap(out, cap, want, MREMAP_MAYMOVE);
      if (m2 == MAP_FAILED) { good = false; break; }
      out = (char *)m2;
      cap = want;
    }
    const size_t in_chunk = zn - n_in < (1u << 30) ? zn - n_in : (1u << 30);
    const size_t out_chunk = cap - n_out < (1u << 30) ? cap - n_out : (1u << 30);
    zs.next_in = (Bytef *)(z + n_in);
    zs.avail_in = (uInt)in_chunk;
    zs.next_out = (Bytef *)(out + n_out);
    zs.avail_out = (uInt)out_chunk;
    const int rc = inflate(&zs, Z_NO_FLUSH);
    n_in += in_chunk - zs.avail_in;
    n_out += out_chunk - zs.avail_out;
    if (rc == Z_STREAM_END) {
      if (n_in == zn) break;                              // last member done
      if (inflateReset(&zs) != Z_OK) good = false;        // another member follows
    } else if (rc != Z_OK && rc != Z_BUF_ERROR) {
      good = false;
    } else if (n_in == zn && zs.avail_out != 0) {
      good = false;                                       // input ended inside a member
    }
  }
  inflateEnd(&zs);
  if (!good) { munmap(out, cap); return false; }
  if (n_out == 0) { munmap(out, cap); data = nullptr; size = 0; return true; }
  if (n_out < cap) {                                      // give the tail back
    const size_t page = (size_t)sysconf(_SC_PAGE_SIZE);
    const size_t keep = (n_out + page - 1) / page * page;
    if (keep < cap) munmap(out + keep, cap - keep);
  }
  data = out;
  size = n_out;
  return true;
}

bool MappedFastq::open(const std::string &path, unsigned threads, size_t max_inflated) {
  fd_ = ::open(path.c_str(), O_RDONLY);
  if (fd_ < 0) return false;
  struct stat st;
  if (fstat(fd_, &st) != 0 || !S_ISREG(st.st_mode)) return false;
  size = (size_t)st.st_size;
  rec_off.clear();
  if (size == 0) { rec_off.push_back(0); canonical = true; return true; }
  void *m = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd_, 0);
  if (m == MAP_FAILED) { data = nullptr; return false; }
  data = (const char *)m;
  madvise(m, size, MADV_SEQUENTIAL);
  if (size >= 2 && (unsigned char)data[0] == 0x1f && (unsigned char)data[1] == 0x8b) {
    const char *z = data;
    const size_t zn = size;
    data = nullptr;
    size = 0;
    const bool ok = inflate_all(z, zn, max_inflated);
    munmap((void *)z, zn);
    if (!ok) { data = nullptr; size = 0; return false; }
    if (size == 0) { rec_off.push_back(0); canonical = true; return true; }
  }
  if (data[0] != '@' || data[size - 1] != '\n') return false;    // missing final newline, not FastQ...
  const char *const end = data + size;
  if (threads == 0) threads = 1;
  std::vector<std::vector<uint64_t>> part(threads);
  std::vector<uint64_t> first_pos(threads, 0), last_end(threads, 0);
  std::vector<char> ok(threads, 1);
  parallel_ranges(size, threads, [&](size_t b, size_t e, unsigned w) {
    const char *p = data + b;
    const char *const stop = data + e;
    if (b != 0) {
      // first record start at or after b: a line that begins with '@' whose second-next line
      // begins with '+' (a quality line may begin with '@', but then the line two below it is
      // a sequence, never '+')
      p = next_line(p - 1, end);
      while (p < end) {
        if (*p == '@') {
          const char *l2 = next_line(next_line(p, end), end);
          if (l2 < end && *l2 == '+') break;
        }
        p = next_line(p, end);
      }
    }
    first_pos[w] = (uint64_t)(p - data);
    std::vector<uint64_t> &out = part[w];
    while (p < stop) {
      const char *l1 = next_line(p, end), *l2 = next_line(l1, end), *l3 = next_line(l2, end);
      const char *nx = next_line(l3, end);
      // canonical: '@' header, '+' separator, four '\n'-terminated lines, no '\r'
      if (*p != '@' || l2 >= end || *l2 != '+' || l3 >= end || nx[-1] != '\n' ||
          memchr(p, '\r', (size_t)(nx - p)) != nullptr) { ok[w] = 0; return; }
      out.push_back((uint64_t)(p - data));
      p = nx;
    }
    last_end[w] = (uint64_t)(p - data);
  });
  for (unsigned w = 0; w < threads; w++)
    if (!ok[w]) return false;
  // every worker must end exactly where the next one started (and the last at the end of file)
  for (unsigned w = 0; w < threads; w++) {
    const uint64_t want = (w + 1 < threads) ? first_pos[w + 1] : (uint64_t)size;
    if (last_end[w] != want) return false;
  }
  size_t total = 0;
  for (auto &v : part) total += v.size();
  rec_off.reserve(total + 1);
  for (auto &v : part) rec_off.insert(rec_off.end(), v.begin(), v.end());
  rec_off.push_back((uint64_t)size);
  canonical = true;
  return true;
}

void MappedFastq::lines(size_t i, std::string_view &name, std::string_view &seq, std::string_view &strand,
                        std::string_view &qual) const {
  const char *p = data + rec_off[i];
  const char *const end = data + rec_off[i + 1];
  const char *l1 = next_line(p, end), *l2 = next_line(l1, end), *l3 = next_line(l2, end);
  name = std::string_view(p, (size_t)(l1 - p - 1));
  seq = std::string_view(l1, (size_t)(l2 - l1 - 1));
  strand = std::string_view(l2, (size_t)(l3 - l2 - 1));
  qual = std::string_view(l3, (size_t)(end - l3 - 1));
}
