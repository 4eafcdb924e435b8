#include "fastq_mmap.hpp"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdlib>
#include <cstring>
#include <thread>

unsigned host_threads() {
  unsigned t = 0;
  if (const char *e = getenv("HUMID_THREADS")) t = (unsigned)atoi(e);
  if (t == 0) t = std::thread::hardware_concurrency();
  if (t == 0) t = 1;
  return t > 64 ? 64 : t;
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

MappedFastq::~MappedFastq() {
  if (data) munmap((void *)data, size);
  if (fd_ >= 0) close(fd_);
}

static inline const char *next_line(const char *p, const char *end) {
  const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
  return nl ? nl + 1 : end;
}

bool MappedFastq::open(const std::string &path, unsigned threads) {
  if (path.size() >= 3 && path.compare(path.size() - 3, 3, ".gz") == 0) return false;
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
  if (data[0] != '@' || data[size - 1] != '\n') return false;    // gzip magic, missing final newline...
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
