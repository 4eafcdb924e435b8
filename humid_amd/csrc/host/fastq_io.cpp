#include "fastq_io.hpp"

#include <fcntl.h>
#include <unistd.h>

#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

static const size_t kBuf = 1u << 20;

FastqReader::FastqReader(const std::string &path) : buf_(kBuf) {
  gz_ = gzopen(path.c_str(), "rb");
  if (gz_) gzbuffer(gz_, 1u << 18);
}

FastqReader::~FastqReader() {
  if (gz_) gzclose(gz_);
}

bool FastqReader::getline(std::string &line) {
  line.clear();
  if (!gz_) return false;
  while (true) {
    if (pos_ == len_) {
      if (eof_) return !line.empty();
      int n = gzread(gz_, buf_.data(), (unsigned)buf_.size());
      if (n <= 0) { eof_ = true; return !line.empty(); }
      len_ = (size_t)n;
      pos_ = 0;
    }
    const char *start = buf_.data() + pos_;
    const char *nl = (const char *)memchr(start, '\n', len_ - pos_);
    if (nl) {
      line.append(start, (size_t)(nl - start));
      pos_ += (size_t)(nl - start) + 1;
      if (!line.empty() && line.back() == '\r') line.pop_back();
      return true;
    }
    line.append(start, len_ - pos_);
    pos_ = len_;
  }
}

bool FastqReader::read(FastqRecord &rec) {
  // skip blank lines between records
  do {
    if (!getline(rec.name)) return false;
  } while (rec.name.empty());
  if (!getline(rec.seq)) return false;
  if (!getline(rec.strand)) return false;
  if (!getline(rec.quality)) rec.quality.clear();
  return true;
}

MultiReader::MultiReader(const std::vector<std::string> &files) {
  for (const std::string &f : files) {
    FastqReader *r = new FastqReader(f);
    if (!r->ok()) { ok_ = false; if (bad_.empty()) bad_ = f; }
    readers_.push_back(r);
  }
}

MultiReader::~MultiReader() {
  for (FastqReader *r : readers_) delete r;
}

bool MultiReader::next(std::vector<FastqRecord> &recs) {
  recs.resize(readers_.size());
  bool all = true;
  for (size_t i = 0; i < readers_.size(); i++)
    if (!readers_[i]->read(recs[i])) all = false;   // src/fastq.cc:39-45: every reader advances
  return all;
}

static bool ends_with(const std::string &s, const char *suf) {
  size_t n = strlen(suf);
  return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}

FastqWriter::FastqWriter(const std::string &path, bool members) {
  if (ends_with(path, ".gz") && !members) {
    gz_ = gzopen(path.c_str(), "wb4");   // fastp Options default compression level 4
    if (gz_) gzbuffer(gz_, 1u << 18);
  } else {
    fd_ = ::open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0666);
    members_ = ends_with(path, ".gz");
  }
  pending_.reserve(kBuf + 4096);
}

bool FastqWriter::compress_member(const char *data, size_t n, std::string &out, int level) {
  out.clear();
  z_stream zs;
  memset(&zs, 0, sizeof zs);
  if (deflateInit2(&zs, level, Z_DEFLATED, 15 + 16, 8, Z_DEFAULT_STRATEGY) != Z_OK) return false;
  out.resize(deflateBound(&zs, (uLong)n) + 64);
  size_t n_in = 0, n_out = 0;
  int rc = Z_OK;
  while (rc != Z_STREAM_END) {
    const size_t in_chunk = n - n_in < (1u << 30) ? n - n_in : (1u << 30);
    const size_t out_chunk = out.size() - n_out < (1u << 30) ? out.size() - n_out : (1u << 30);
    zs.next_in = (Bytef *)(data + n_in);
    zs.avail_in = (uInt)in_chunk;
    zs.next_out = (Bytef *)(&out[0] + n_out);
    zs.avail_out = (uInt)out_chunk;
    rc = deflate(&zs, n_in + in_chunk == n ? Z_FINISH : Z_NO_FLUSH);
    n_in += in_chunk - zs.avail_in;
    n_out += out_chunk - zs.avail_out;
    if (rc != Z_OK && rc != Z_STREAM_END && rc != Z_BUF_ERROR) break;
    if (rc == Z_BUF_ERROR && n_out == out.size()) out.resize(out.size() * 2);
  }
  deflateEnd(&zs);
  out.resize(n_out);
  return rc == Z_STREAM_END;
}

// false: the bytes are not all in the file (ENOSPC, EIO, ...); an interrupted call is retried
static bool pwrite_all(int fd, const char *data, size_t n, uint64_t off) {
  while (n) {
    const ssize_t w = ::pwrite(fd, data, n, (off_t)off);
    if (w < 0 && errno == EINTR) continue;
    if (w <= 0) return false;
    data += w;
    n -= (size_t)w;
    off += (uint64_t)w;
  }
  return true;
}

void FastqWriter::put(const char *data, size_t n) {
  if (n == 0) return;
  if (fd_ < 0) { failed_ = true; return; }
  if (!pwrite_all(fd_, data, n, off_)) failed_ = true;
  off_ += n;
  wrote_ = true;
}

void FastqWriter::gz_put(const char *data, size_t n) {
  if (!gz_) { failed_ = true; return; }
  while (n) {                                       // gzwrite takes an unsigned length
    const unsigned chunk = n < (1u << 30) ? (unsigned)n : (1u << 30);
    if (gzwrite(gz_, data, chunk) != (int)chunk) { failed_ = true; return; }
    data += chunk;
    n -= chunk;
  }
}

void FastqWriter::flush() {
  if (pending_.empty()) return;
  if (gz_) gz_put(pending_.data(), pending_.size());
  else if (members_) {
    if (compress_member(pending_.data(), pending_.size(), z_)) put(z_.data(), z_.size());
    else failed_ = true;
  } else put(pending_.data(), pending_.size());
  pending_.clear();
}

void FastqWriter::write(const char *data, size_t n) {
  pending_.append(data, n);
  if (pending_.size() >= kBuf) flush();
}

void FastqWriter::write_member(const std::string &z) {
  flush();
  put(z.data(), z.size());
}

void FastqWriter::write_parts(const std::vector<std::string> &parts, unsigned n_parts) {
  flush();
  if (fd_ < 0) {                                    // streaming gzip: in order, one thread
    for (unsigned k = 0; k < n_parts; k++)
      if (!parts[k].empty()) gz_put(parts[k].data(), parts[k].size());
    return;
  }
  std::vector<uint64_t> at(n_parts + 1, off_);
  for (unsigned k = 0; k < n_parts; k++) at[k + 1] = at[k] + parts[k].size();
  if (at[n_parts] == off_) return;
  // one writer: concurrent pwrite()s into ONE file serialise on the inode lock and measured
  // 15-30 % slower than this loop (tools/ab_write.sh, round 1)
  for (unsigned k = 0; k < n_parts; k++)
    if (!pwrite_all(fd_, parts[k].data(), parts[k].size(), at[k])) failed_ = true;
  off_ = at[n_parts];
  wrote_ = true;
}

bool FastqWriter::close() {
  if (closed_) return ok();
  if (fd_ < 0 && !gz_) { failed_ = true; closed_ = true; return false; }   // never opened
  flush();
  if (fd_ >= 0 && members_ && !wrote_) {            // valid empty gzip
    if (compress_member("", 0, z_)) put(z_.data(), z_.size());
    else failed_ = true;
  }
  if (gz_) { if (gzclose(gz_) != Z_OK) failed_ = true; gz_ = nullptr; }
  if (fd_ >= 0) {
    // deferred write errors (NFS, quota) surface here; close() is never retried (POSIX: the
    // descriptor is gone either way)
    if (::close(fd_) != 0 && errno != EINTR) failed_ = true;
    fd_ = -1;
  }
  closed_ = true;
  return ok();
}

FastqWriter::~FastqWriter() { close(); }
