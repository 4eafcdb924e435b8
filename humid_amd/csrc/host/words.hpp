// words.hpp -- word extraction of the `humid` host: FastQ records -> 2-bit packed word + filtered.
//
// Behaviour follows /root/reference/src/fastq.cc (extractUMI_ :72-93, getNucleotides :116-144,
// makeWord :146-161, ntFromFile :220-230, makeFileName :174-181) and src/humid.cc preCompute
// :38-59; the implementation packs straight into the C-ABI word (include/humid_hip.h) instead of
// building vector<uint8_t> words.
#pragma once
#include <cstdint>
#include <string>
#include <string_view>
#include <vector>

#include "fastq_io.hpp"

namespace humid_host {

// UMI in a header line: text before the first space; last '_' field if it is all ACGT, else last
// ':' field if it is all ACGT, else none.  `header` may include the leading '@'.
std::string_view header_umi(std::string_view header);

bool valid_umi(std::string_view umi);                                   // non-empty, only ACGT
std::string_view last_field(std::string_view s, char sep);             // "" when sep is absent
std::vector<size_t> nt_from_file(size_t files, size_t length);         // remainder to the LAST file

struct WordPlan {
  size_t word_nt = 24;
  size_t header_umi = 0;           // symbols taken from the first file's header UMI
  std::vector<size_t> take;        // symbols taken from each file's read
};
// first_header_umi: UMI length found in the first record of the first file (peekUMI)
WordPlan make_plan(size_t first_header_umi, size_t n_files, size_t word_nt);

// One record set -> packed word (first symbol most significant).  Returns true when the word is
// filtered (a symbol outside ACGT, including 'N' padding of short UMIs/reads; coded as 'G').
// Requires plan.word_nt <= 32.
bool make_word(const std::vector<FastqRecord> &recs, const WordPlan &plan, uint64_t &word);
// the same on views (fast path: records read straight from the file mapping)
bool make_word(std::string_view first_header, const std::string_view *seqs, size_t n_files,
               const WordPlan &plan, uint64_t &word);

// 33 <= plan.word_nt <= 64: two uint64, [0] = the first word_nt-32 symbols, [1] = the last 32
bool make_word_wide(const std::vector<FastqRecord> &recs, const WordPlan &plan, uint64_t word[2]);
bool make_word_wide(std::string_view first_header, const std::string_view *seqs, size_t n_files,
                    const WordPlan &plan, uint64_t word[2]);

// getNucleotides (src/fastq.cc:116-144) as raw bytes, for the device-side packing of
// humid_dedup_run_bases: out[plan.word_nt] = the header UMI cut or padded with 'N' to plan.header_umi
// symbols, then the first plan.take[f] bytes of every file's read, 'N' where a read is short.  The
// bytes are copied as they stand in the FastQ; the device maps and validates them.
void gather_bases(std::string_view first_header, const std::string_view *seqs, size_t n_files,
                  const WordPlan &plan, uint8_t *out);

// <dir>/<basename with _suffix inserted before the first '.'>
std::string make_file_name(const std::string &path, const std::string &dir, const std::string &suffix);

}  // namespace humid_host
