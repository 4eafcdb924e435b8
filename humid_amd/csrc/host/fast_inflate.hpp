// fast_inflate.hpp -- DEFLATE (RFC 1951) / gzip (RFC 1952) decoder of the `humid` host.
//
// The reference reads gzip FastQ through isa-l's igzip (north star: "isa-l gunzip"); the image has
// zlib only, whose inflate is what bounds the end-to-end time of gzip inputs (one stream per file,
// about 0.5 GB/s).  This decoder is written for that one job -- whole files inflated into one
// contiguous, growing buffer that is kept -- and uses the usual fast-path ideas: a 64-bit bit
// buffer refilled without branches, two-level decode tables whose entries already carry base value
// and extra-bit count, word-wise match copies.  Member CRC-32s are checked afterwards on all cores
// (zlib crc32 + crc32_combine).  Anything it does not like makes the caller fall back to zlib.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace humid_host {

// Inflates every gzip member of [in, in + n_in) into a buffer obtained from `grow`:
//   grow(user, min_capacity) -> pointer to a buffer of at least min_capacity bytes whose first
//   *n_out bytes are the output so far (it may move), or nullptr to give up (retention bound).
// Returns true on success; *n_out = total inflated size.  threads: workers of the CRC check.
typedef char *(*GrowFn)(void *user, size_t min_capacity, size_t *capacity);
bool fast_gunzip(const uint8_t *in, size_t n_in, GrowFn grow, void *user, size_t *n_out, unsigned threads);

}  // namespace humid_host
