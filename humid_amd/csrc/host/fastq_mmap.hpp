// fastq_mmap.hpp -- fast host path of the `humid` CLI for plain (uncompressed) FastQ files:
// the files are mapped, the records are indexed in parallel, pass 1 builds the packed words on
// all cores and pass 2 copies / annotates records straight from the mapping (record offsets are
// kept, nothing is parsed twice).  SURVEY.md section 8(f).1: the single-threaded FastQ streaming
// of the reference (src/fastq.cc:96-114 + fastp) is what bounds end-to-end time.
//
// gzip input (any number of members) is inflated ONCE into memory (one thread per file, zlib; the
// image has no isa-l) and then treated like a mapping, so pass 2 does not inflate again; the
// retained bytes are bounded (HUMID_RETAIN_GB, default a third of the physical memory over all
// files) -- beyond that the streaming reader takes over.
//
// Only "canonical" contents take this path: four lines per record, '\n' line ends, no blank lines.
// Anything else (CRLF, blank lines) goes through the streaming reader of fastq_io.hpp, whose
// output is byte-identical by construction (tests/test_cli_host.py compares the two).
#pragma once
#include <cstdint>
#include <functional>
#include <string>
#include <string_view>
#include <vector>

struct MappedFastq {
  const char *data = nullptr;
  size_t size = 0;
  std::vector<uint64_t> rec_off;   // start of every record, plus a final sentinel (= end of the last)
  bool canonical = false;

  ~MappedFastq();
  MappedFastq() = default;
  MappedFastq(const MappedFastq &) = delete;
  MappedFastq &operator=(const MappedFastq &) = delete;

  // maps (or inflates) the file and indexes its records with `threads` workers; false = use the
  // streaming path.  max_inflated: bound on the bytes kept for a gzip file.
  bool open(const std::string &path, unsigned threads, size_t max_inflated = ~(size_t)0);
  size_t records() const { return rec_off.empty() ? 0 : rec_off.size() - 1; }

  // the four lines of record i (no line terminators)
  void lines(size_t i, std::string_view &name, std::string_view &seq, std::string_view &strand,
             std::string_view &qual) const;
  // After the last use: gives the pages of the mapping back with `threads` workers
  // (madvise MADV_DONTNEED, which only takes the address-space lock shared).  Left to process exit, the
  // kernel unmaps 6 GB of page-cache pages on ONE core: 0.2 s of a 0.65 s run (tools/e2e_edges.py).
  void drop_pages(unsigned threads) const;
  std::string_view raw(size_t i) const {   // whole record including its final '\n'
    return std::string_view(data + rec_off[i], (size_t)(rec_off[i + 1] - rec_off[i]));
  }

 private:
  bool inflate_all(const char *z, size_t zn, size_t max_inflated);
  int fd_ = -1;
};

size_t retain_budget_bytes();             // HUMID_RETAIN_GB or a third of the physical memory

unsigned host_threads();                  // HUMID_THREADS or hardware concurrency, 1..64
// runs fn(begin, end, worker) over [0, n) split into contiguous ranges, one per worker
void parallel_ranges(size_t n, unsigned threads, const std::function<void(size_t, size_t, unsigned)> &fn);
