// fastq_io.hpp -- minimal FastQ reader/writer for the `humid` host (plain + gzip via zlib).
//
// The reference streams FastQ through OpenGene/fastp (FastqReader::read() -> Read*, Writer;
// used at /root/reference/src/fastq.cc:37-47,96-114 and src/humid.cc:214-238,262-289).  fastp
// is an un-vendored submodule, so this host carries its own record reader with the same
// observable behaviour for this path: four-line records, name line kept verbatim including
// '@', '\r' stripped, reading stops at the first file that runs out (src/fastq.cc:41-43).
#pragma once
#include <zlib.h>

#include <cstdint>
#include <string>
#include <vector>

struct FastqRecord {
  std::string name;     // header line including the leading '@' (fastp Read::mName)
  std::string seq;      // fastp Read::mSeq
  std::string strand;   // the '+' line
  std::string quality;
  // fastp Read::toString(): the four lines, '\n' terminated
  void append_to(std::string &out) const {
    out.append(name).push_back('\n');
    out.append(seq).push_back('\n');
    out.append(strand).push_back('\n');
    out.append(quality).push_back('\n');
  }
};

class FastqReader {
 public:
  explicit FastqReader(const std::string &path);
  ~FastqReader();
  bool ok() const { return gz_ != nullptr; }
  bool read(FastqRecord &rec);   // false at end of file / truncated record
 private:
  bool getline(std::string &line);
  gzFile gz_ = nullptr;          // gzopen reads plain files transparently
  std::vector<char> buf_;
  size_t pos_ = 0, len_ = 0;
  bool eof_ = false;
};

// Lock-step over several files (src/fastq.cc:96-114 readFiles): one record from each file per
// step; stops when ANY file is exhausted.
class MultiReader {
 public:
  explicit MultiReader(const std::vector<std::string> &files);
  ~MultiReader();
  bool ok() const { return ok_; }
  const std::string &bad_file() const { return bad_; }
  bool next(std::vector<FastqRecord> &recs);
 private:
  std::vector<FastqReader *> readers_;
  bool ok_ = true;
  std::string bad_;
};

// Output file; gzip-compressed when the name ends in ".gz" (as fastp's Writer decides).
// members = true: the file is written as a sequence of independent gzip members, so that the
// caller can compress blocks on several threads (compress_member) and hand them over in order
// (write_member).  Tools read such a file exactly like a single-member one.
class FastqWriter {
 public:
  explicit FastqWriter(const std::string &path, bool members = false);
  ~FastqWriter();
  // false once the file could not be created or ANY write / compression / close failed (disk
  // full, I/O error): the caller must report it -- a truncated output with exit code 0 is silent
  // data loss in a deduplication tool
  bool ok() const { return !failed_ && (closed_ || fd_ >= 0 || gz_ != nullptr); }
  bool gz_members() const { return members_; }
  bool close();                                     // flushes, closes; returns ok()
  void write(const char *data, size_t n);
  void write_member(const std::string &z);          // one precompressed gzip member
  // parts[0 .. n_parts) back to back (plain text, or precompressed members of a members-mode file)
  void write_parts(const std::vector<std::string> &parts, unsigned n_parts);
  void flush();
  // data -> one complete gzip member (fastp Options default compression level 4)
  static bool compress_member(const char *data, size_t n, std::string &out, int level = 4);
 private:
  void put(const char *data, size_t n);             // at the current end of the file
  int fd_ = -1;                                     // plain files and members-mode gzip files
  uint64_t off_ = 0;
  gzFile gz_ = nullptr;                             // streaming gzip (single thread)
  bool members_ = false;
  bool wrote_ = false;
  bool failed_ = false;
  bool closed_ = false;
  void gz_put(const char *data, size_t n);          // streaming gzip, any size
  std::string pending_, z_;
};
