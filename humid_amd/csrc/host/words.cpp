#include "words.hpp"

#include <cstring>

namespace humid_host {

static inline int code_of(char c) {
  switch (c) {
    case 'A': return 0;
    case 'C': return 1;
    case 'G': return 2;
    case 'T': return 3;
    default: return -1;
  }
}

bool valid_umi(std::string_view umi) {
  if (umi.empty()) return false;
  for (char c : umi)
    if (code_of(c) < 0) return false;
  return true;
}

std::string_view last_field(std::string_view s, char sep) {
  size_t p = s.rfind(sep);
  if (p == std::string_view::npos) return std::string_view();
  return s.substr(p + 1);
}

std::string_view header_umi(std::string_view header) {
  std::string_view id = header.substr(0, header.find(' '));
  std::string_view u = last_field(id, '_');
  if (valid_umi(u)) return u;
  u = last_field(id, ':');
  if (valid_umi(u)) return u;
  return std::string_view();
}

std::vector<size_t> nt_from_file(size_t files, size_t length) {
  std::vector<size_t> v(files, files ? length / files : 0);
  if (files) v.back() += length % files;
  return v;
}

WordPlan make_plan(size_t first_header_umi, size_t n_files, size_t word_nt) {
  WordPlan p;
  p.word_nt = word_nt;
  size_t from_files = word_nt > first_header_umi ? word_nt - first_header_umi : 0;
  p.take = nt_from_file(n_files, from_files);
  p.header_umi = first_header_umi < word_nt ? first_header_umi : word_nt;
  return p;
}

// appends `want` symbols of `s` (cut or padded with a non-ACGT symbol) to the packed word
template <class Acc>
static inline void push_symbols(std::string_view s, size_t want, Acc &w, bool &filtered) {
  size_t have = s.size() < want ? s.size() : want;
  for (size_t i = 0; i < have; i++) {
    int c = code_of(s[i]);
    if (c < 0) { c = 2; filtered = true; }       // unknown base: code of 'G', word filtered
    w = (w << 2) | (Acc)c;
  }
  for (size_t i = have; i < want; i++) {         // 'N' padding of a short UMI / read
    w = (w << 2) | 2u;
    filtered = true;
  }
}

bool make_word(std::string_view first_header, const std::string_view *seqs, size_t n_files,
               const WordPlan &plan, uint64_t &word) {
  uint64_t w = 0;
  bool filtered = false;
  if (plan.header_umi > 0) push_symbols(header_umi(first_header), plan.header_umi, w, filtered);
  for (size_t f = 0; f < n_files; f++) push_symbols(seqs[f], plan.take[f], w, filtered);
  word = w;
  return filtered;
}

static inline uint8_t *copy_padded(std::string_view s, size_t want, uint8_t *out) {
  const size_t have = s.size() < want ? s.size() : want;
  memcpy(out, s.data(), have);
  if (have < want) memset(out + have, 'N', want - have);
  return out + want;
}

void gather_bases(std::string_view first_header, const std::string_view *seqs, size_t n_files,
                  const WordPlan &plan, uint8_t *out) {
  if (plan.header_umi > 0) out = copy_padded(header_umi(first_header), plan.header_umi, out);
  for (size_t f = 0; f < n_files; f++) out = copy_padded(seqs[f], plan.take[f], out);
}

bool make_word_wide(std::string_view first_header, const std::string_view *seqs, size_t n_files,
                    const WordPlan &plan, uint64_t word[2]) {
  unsigned __int128 w = 0;
  bool filtered = false;
  if (plan.header_umi > 0) push_symbols(header_umi(first_header), plan.header_umi, w, filtered);
  for (size_t f = 0; f < n_files; f++) push_symbols(seqs[f], plan.take[f], w, filtered);
  word[0] = (uint64_t)(w >> 64);     // the first word_nt - 32 symbols
  word[1] = (uint64_t)w;             // the last 32
  return filtered;
}

bool make_word_wide(const std::vector<FastqRecord> &recs, const WordPlan &plan, uint64_t word[2]) {
  std::string_view seqs[64];
  const size_t n = recs.size() < 64 ? recs.size() : 64;
  for (size_t f = 0; f < n; f++) seqs[f] = recs[f].seq;
  return make_word_wide(recs.front().name, seqs, n, plan, word);
}

bool make_word(const std::vector<FastqRecord> &recs, const WordPlan &plan, uint64_t &word) {
  std::string_view seqs[64];
  const size_t n = recs.size() < 64 ? recs.size() : 64;
  for (size_t f = 0; f < n; f++) seqs[f] = recs[f].seq;
  return make_word(recs.front().name, seqs, n, plan, word);
}

std::string make_file_name(const std::string &path, const std::string &dir, const std::string &suffix) {
  size_t slash = path.find_last_of('/');
  std::string base = slash == std::string::npos ? path : path.substr(slash + 1);
  size_t dot = base.find('.');
  std::string out = dir + '/';
  if (dot == std::string::npos) return out + base + '_' + suffix;   // the reference throws here
  return out + base.substr(0, dot) + '_' + suffix + base.substr(dot);
}

}  // namespace humid_host
