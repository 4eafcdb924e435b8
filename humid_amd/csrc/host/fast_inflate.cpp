#include "fast_inflate.hpp"

#include <zlib.h>   // crc32, crc32_combine only

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

namespace humid_host {
namespace {

// Decode table entry: [31:16] literal / base value / subtable offset, [15] literal, [14] end of
// block, [13] subtable pointer, [11:8] length of the Huffman code itself (subtable pointer: index
// bits of the subtable), [7:0] bits to consume = code + extra bits (second level: what is left of
// the code + extra bits).  One shift consumes code and extra bits together; the extra value is cut
// out of a saved copy of the bit buffer, off the look-up -> shift -> look-up dependency chain.
// 0 = no code ends here (invalid input).
constexpr uint32_t F_LIT = 0x8000u, F_EOB = 0x4000u, F_SUB = 0x2000u;
// [12] with F_LIT in the first level of the literal/length table: TWO literals whose codes both fit
// the index ([23:16] the first, [31:24] the second; [7:0] = the sum of both code lengths).  Table
// look-ups are a serial dependency chain (bits -> entry -> shift -> bits); text and sequence data
// is mostly literals with short codes, so pairing them nearly halves the chain.
constexpr uint32_t F_LIT2 = 0x1000u;
constexpr unsigned LL_TB = 11, D_TB = 8, PRE_TB = 7;

const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59,
                               67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769,
                                1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

enum Kind { K_LITLEN, K_DIST, K_PRE };

inline uint32_t payload(Kind k, unsigned sym) {
  switch (k) {
    case K_LITLEN:
      if (sym < 256) return (sym << 16) | F_LIT;
      if (sym == 256) return F_EOB;
      if (sym <= 285) return ((uint32_t)kLenBase[sym - 257] << 16) | ((uint32_t)kLenExtra[sym - 257] << 8);
      return 0;
    case K_DIST:
      if (sym < 30) return ((uint32_t)kDistBase[sym] << 16) | ((uint32_t)kDistExtra[sym] << 8);
      return 0;
    default:
      return sym << 16;
  }
}

// payload (extra-bit count in [11:8]) + code length -> table entry
inline uint32_t finish_entry(uint32_t pl, unsigned codelen) {
  const unsigned extra = (pl >> 8) & 15;
  return (pl & ~0x0f00u) | (codelen << 8) | (codelen + extra);
}

inline unsigned reverse_bits(unsigned code, unsigned len) {
  unsigned r = 0;
  for (unsigned i = 0; i < len; i++) { r = (r << 1) | (code & 1); code >>= 1; }
  return r;
}

// canonical Huffman code lengths -> two-level table (first level tb bits)
bool build_table(const uint8_t *lens, unsigned n, unsigned tb, Kind kind, std::vector<uint32_t> &t) {
  unsigned count[16] = {0};
  for (unsigned i = 0; i < n; i++) count[lens[i] & 15]++;
  count[0] = 0;
  unsigned long used = 0;
  for (unsigned len = 1; len <= 15; len++) used += (unsigned long)count[len] << (15 - len);
  if (used > (1ul << 15)) return false;                       // over-subscribed
  unsigned next[16];
  unsigned code = 0;
  for (unsigned len = 1; len <= 15; len++) { code = (code + count[len - 1]) << 1; next[len] = code; }
  const unsigned first = 1u << tb;
  t.assign(first, 0);
  uint8_t submax[1u << LL_TB];
  memset(submax, 0, first);
  struct Long { uint16_t sym, rev; uint8_t len; };
  Long longs[320];
  unsigned n_long = 0;
  for (unsigned sym = 0; sym < n; sym++) {
    const unsigned len = lens[sym];
    if (!len) continue;
    const unsigned rev = reverse_bits(next[len]++, len);
    const uint32_t pl = payload(kind, sym);
    if (len <= tb) {
      if (pl || kind == K_PRE)
        for (unsigned i = rev; i < first; i += 1u << len) t[i] = finish_entry(pl, len);
    } else {
      const unsigned p = rev & (first - 1);
      if (len > submax[p]) submax[p] = (uint8_t)len;
      longs[n_long++] = Long{(uint16_t)sym, (uint16_t)rev, (uint8_t)len};
    }
  }
  for (unsigned p = 0; p < first; p++) {
    if (!submax[p]) continue;
    const unsigned bits = submax[p] - tb;
    const size_t off = t.size();
    if (off + (1u << bits) > 65535u) return false;
    t.resize(off + (1u << bits), 0);
    t[p] = ((uint32_t)off << 16) | F_SUB | (bits << 8) | tb;
  }
  for (unsigned k = 0; k < n_long; k++) {
    const Long &l = longs[k];
    const uint32_t pl = payload(kind, l.sym);
    if (!pl && kind != K_PRE) continue;
    const uint32_t main = t[l.rev & (first - 1)];
    const unsigned off = main >> 16, bits = (main >> 8) & 15;
    for (unsigned i = l.rev >> tb; i < (1u << bits); i += 1u << (l.len - tb)) t[off + i] = finish_entry(pl, l.len - tb);
  }
  return true;
}

// pair up literals in the first level of a literal/length table (see F_LIT2)
void pair_literals(std::vector<uint32_t> &t) {
  const unsigned first = 1u << LL_TB;
  std::vector<uint32_t> o(t.begin(), t.begin() + first);
  for (unsigned i = 0; i < first; i++) {
    const uint32_t e1 = o[i];
    if ((e1 & (F_LIT | F_SUB)) != F_LIT) continue;
    const unsigned l1 = e1 & 0xff;
    if (l1 >= LL_TB) continue;
    const uint32_t e2 = o[i >> l1];
    const unsigned l2 = e2 & 0xff;
    if ((e2 & (F_LIT | F_SUB)) != F_LIT || l2 == 0 || l1 + l2 > LL_TB) continue;
    t[i] = F_LIT | F_LIT2 | (e1 & 0x00ff0000u) | ((e2 & 0x00ff0000u) << 8) | (l1 + l2);
  }
}

struct Bits {
  const uint8_t *p, *end;
  uint64_t buf = 0;
  unsigned cnt = 0;
  size_t phantom = 0;            // zero bytes supplied past the end (safe refill)
  bool clean = true;             // no garbage above cnt
};

inline uint64_t load64(const uint8_t *p) {
  uint64_t v;
  memcpy(&v, p, 8);
  return v;                      // little-endian host (x86-64)
}

inline void refill_fast(Bits &b) {   // needs b.p + 8 <= b.end; afterwards 56 <= cnt <= 63
  b.buf |= load64(b.p) << b.cnt;
  b.p += (63 - b.cnt) >> 3;
  b.cnt |= 56;
  b.clean = false;
}

inline void make_clean(Bits &b) {
  if (!b.clean) { b.buf &= (b.cnt >= 64) ? ~0ull : ((1ull << b.cnt) - 1); b.clean = true; }
}

inline void refill_safe(Bits &b) {
  make_clean(b);
  while (b.cnt <= 56) {
    uint64_t byte = 0;
    if (b.p < b.end) byte = *b.p++; else { b.phantom++; }
    b.buf |= byte << b.cnt;
    b.cnt += 8;
  }
}

inline void consume(Bits &b, unsigned n) { b.buf >>= n; b.cnt -= n; }

struct Decoder {
  std::vector<uint32_t> lt, dt, pt;
  Bits in;
  bool final_block = false;
};

enum { R_EOB = 0, R_NEED_OUT = 1, R_ERROR = -1, R_INPUT_TAIL = 2 };

// symbols of one Huffman block.  FAST: unchecked refills and word copies, leaves with
// R_INPUT_TAIL / R_NEED_OUT near the end of either buffer.  The bit buffer lives in locals for the
// whole loop (byte stores to `out` may alias anything reachable through a reference).
template <bool FAST>
int decode_huffman(Decoder &d, uint8_t *out, size_t &out_pos, size_t out_cap, size_t member_begin) {
  const uint32_t *const lt = d.lt.data(), *const dt = d.dt.data();
  if (!FAST) make_clean(d.in);
  uint64_t buf = d.in.buf;
  unsigned cnt = d.in.cnt;
  const uint8_t *ip = d.in.p;
  const uint8_t *const iend = d.in.end;
  size_t phantom = d.in.phantom;
  uint8_t *op = out + out_pos;
  uint8_t *const olimit = out + out_cap - 320;          // caller guarantees out_cap >= 320
  const uint8_t *const mbegin = out + member_begin;
  int rc = R_ERROR;
#define HUMID_CONSUME(n) do { buf >>= (n); cnt -= (n); } while (0)
#define HUMID_REFILL()                                                               \
  do {                                                                               \
    if (FAST) {                                                                      \
      buf |= load64(ip) << cnt;          /* afterwards 56 <= cnt <= 63 */            \
      ip += (63 - cnt) >> 3;                                                         \
      cnt |= 56;                                                                     \
    } else {                                                                         \
      while (cnt <= 56) {                                                            \
        uint64_t byte = 0;                                                           \
        if (ip < iend) byte = *ip++; else phantom++;                                 \
        buf |= byte << cnt;                                                          \
        cnt += 8;                                                                    \
      }                                                                              \
    }                                                                                \
  } while (0)
  uint32_t e;
  for (;;) {
    if (op > olimit) { rc = R_NEED_OUT; break; }
    if (FAST && iend - ip < 16) { rc = R_INPUT_TAIL; break; }        // two refills per turn at most
    HUMID_REFILL();
    if (!FAST && phantom > 8) break;                                  // decoding zeros past a truncated input
    e = lt[buf & ((1u << LL_TB) - 1)];
  have_entry:
    if ((e & (F_LIT | F_SUB)) == F_LIT) {
      // up to four look-ups of one or two literals from one refill (<= 11 bits each, 44 of >= 56)
      unsigned k = 0;
      do {
        HUMID_CONSUME(e & 0xff);
        const uint16_t two = (uint16_t)(e >> 16);        // second byte is scratch for a single literal
        memcpy(op, &two, 2);
        op += 1 + ((e >> 12) & 1);
        e = lt[buf & ((1u << LL_TB) - 1)];
      } while (++k < 4 && (e & (F_LIT | F_SUB)) == F_LIT);
      // e was looked up with the bits that are left: go on without a refill only when a whole
      // length / distance pair (<= 48 bits) is certainly there
      if (!FAST || cnt < 48 || (e & (F_LIT | F_SUB)) == F_LIT) continue;
    }
    if (e & F_SUB) {
      HUMID_CONSUME(LL_TB);
      e = lt[(e >> 16) + (buf & ((1u << ((e >> 8) & 15)) - 1))];
    }
    const uint64_t saved = buf;
    HUMID_CONSUME(e & 0xff);                                           // code and extra bits at once
    if (e & F_LIT) { *op++ = (uint8_t)(e >> 16); continue; }           // second-level entries are single literals
    if (e & F_EOB) { rc = R_EOB; break; }
    {
      if (!FAST) { HUMID_REFILL(); }
      uint32_t f;
      f = dt[buf & ((1u << D_TB) - 1)];                                // does not wait for the length's extra bits
      if ((e >> 16) == 0 || (e & 0xff) == 0) break;                    // no such code
      const unsigned lcode = (e >> 8) & 15, ltot = e & 0xff;
      const unsigned len = (e >> 16) + (unsigned)((saved >> lcode) & ((1u << (ltot - lcode)) - 1));
      if (f & F_SUB) {
        HUMID_CONSUME(D_TB);
        f = dt[(f >> 16) + (buf & ((1u << ((f >> 8) & 15)) - 1))];
      }
      const uint64_t saved2 = buf;
      HUMID_CONSUME(f & 0xff);
      if ((f >> 16) == 0 || (f & 0xff) == 0) break;
      const unsigned dcode = (f >> 8) & 15, dtot = f & 0xff;
      const size_t dist = (f >> 16) + (size_t)((saved2 >> dcode) & ((1u << (dtot - dcode)) - 1));
      if (dist > (size_t)(op - mbegin)) break;                         // before the start of the member
      const uint8_t *src = op - dist;
      uint8_t *const stop = op + len;
      // FAST: the next symbol's table entry is fetched BEFORE the copy, so that its latency hides
      // behind the stores (same conditions as at the top of the loop)
      const bool ahead = FAST && stop <= olimit && iend - ip >= 16;
      if (ahead) {
        HUMID_REFILL();
        e = lt[buf & ((1u << LL_TB) - 1)];
      }
      if (FAST && dist >= 8) {
        // word copies (each 8-byte source lies wholly before its destination), overshooting by up to
        // 15 bytes.  Sequence data is matched in pieces of about 10 bytes: two words without a
        // loop cover nearly every match, so the (unpredictable) trip count costs no branch miss
        memcpy(op, src, 8);
        memcpy(op + 8, src + 8, 8);
        if (len > 16) {
          uint8_t *o2 = op + 16;
          const uint8_t *s2 = src + 16;
          do { memcpy(o2, s2, 8); o2 += 8; s2 += 8; } while (o2 < stop);
        }
      } else if (dist == 1) {
        memset(op, *src, len);
      } else {
        while (op < stop) *op++ = *src++;
      }
      op = stop;
      if (ahead) goto have_entry;
    }
  }
#undef HUMID_CONSUME
#undef HUMID_REFILL
  d.in.buf = buf;
  d.in.cnt = cnt;
  d.in.p = ip;
  d.in.phantom = phantom;
  d.in.clean = !FAST;
  out_pos = (size_t)(op - out);
  return rc;
}

// bit-level helpers of the (rare) header paths; always the safe refill
inline unsigned take(Bits &b, unsigned n) {
  refill_safe(b);
  const unsigned v = (unsigned)(b.buf & ((1u << n) - 1));
  consume(b, n);
  return v;
}

inline size_t byte_pos(Bits &b, const uint8_t *base) {   // input position of the next unread byte (byte aligned)
  return (size_t)(b.p - base) + b.phantom - (b.cnt >> 3);
}

inline void align_to_byte(Bits &b) {
  make_clean(b);
  consume(b, b.cnt & 7);
}

// drop the bit buffer and continue reading bytes at the aligned position
inline const uint8_t *rewind_to_bytes(Bits &b) {
  align_to_byte(b);
  const uint8_t *q = b.p - (b.cnt >> 3);
  b.buf = 0;
  b.cnt = 0;
  b.clean = true;
  return q;
}

// sum of 2^(15 - len) over the used codes == 2^15: the code is complete (every zlib-made code is,
// except a distance code with a single symbol)
inline bool kraft_complete(const uint8_t *lens, unsigned n) {
  unsigned long used = 0;
  for (unsigned i = 0; i < n; i++)
    if (lens[i]) used += 1ul << (15 - lens[i]);
  return used == (1ul << 15);
}

// strict: additionally insist on complete codes -- the block-start search of the parallel decoder
// uses this to tell a real header from random bits
bool read_dynamic_tables(Decoder &d, bool strict = false) {
  Bits &b = d.in;
  const unsigned hlit = take(b, 5) + 257, hdist = take(b, 5) + 1, hclen = take(b, 4) + 4;
  if (hlit > 286 || hdist > 30) return false;
  static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
  uint8_t pre[19] = {0};
  for (unsigned i = 0; i < hclen; i++) pre[order[i]] = (uint8_t)take(b, 3);
  if (strict && !kraft_complete(pre, 19)) return false;
  if (!build_table(pre, 19, PRE_TB, K_PRE, d.pt)) return false;
  uint8_t lens[286 + 30 + 16] = {0};
  unsigned i = 0;
  while (i < hlit + hdist) {
    refill_safe(b);
    if (b.phantom > 8) return false;
    const uint32_t e = d.pt[b.buf & ((1u << PRE_TB) - 1)];
    if ((e & 0xff) == 0) return false;
    consume(b, e & 0xff);
    const unsigned sym = e >> 16;
    if (sym < 16) { lens[i++] = (uint8_t)sym; continue; }
    unsigned rep, val = 0;
    if (sym == 16) { if (i == 0) return false; val = lens[i - 1]; rep = 3 + take(b, 2); }
    else if (sym == 17) rep = 3 + take(b, 3);
    else rep = 11 + take(b, 7);
    if (i + rep > hlit + hdist) return false;
    while (rep--) lens[i++] = (uint8_t)val;
  }
  if (lens[256] == 0) return false;                                 // no end-of-block code
  if (strict) {
    unsigned nd = 0;
    for (unsigned k = 0; k < hdist; k++) nd += lens[hlit + k] != 0;
    if (!kraft_complete(lens, hlit) || (nd > 1 && !kraft_complete(lens + hlit, hdist))) return false;
  }
  if (!build_table(lens, hlit, LL_TB, K_LITLEN, d.lt) || !build_table(lens + hlit, hdist, D_TB, K_DIST, d.dt)) return false;
  pair_literals(d.lt);
  return true;
}

bool fixed_tables(Decoder &d) {
  uint8_t lens[288 + 32];
  for (unsigned i = 0; i < 144; i++) lens[i] = 8;
  for (unsigned i = 144; i < 256; i++) lens[i] = 9;
  for (unsigned i = 256; i < 280; i++) lens[i] = 7;
  for (unsigned i = 280; i < 288; i++) lens[i] = 8;
  for (unsigned i = 0; i < 32; i++) lens[288 + i] = 5;
  if (!build_table(lens, 288, LL_TB, K_LITLEN, d.lt) || !build_table(lens + 288, 32, D_TB, K_DIST, d.dt)) return false;
  pair_literals(d.lt);
  return true;
}

struct Member { size_t begin, end; uint32_t crc; };

// ------------------------------------------------------------------------------------------
// Parallel inflate of a gzip file -- one large member or many (bgzip, this tool's own output) --
// in two passes, after Kerbiriou & Chikhi's pugz idea.
//
// A DEFLATE stream can only be entered at a block boundary, and a block may refer to the 32 KiB
// before it.  Pass 1: the compressed bytes are cut into chunks; for every cut a block start is
// SEARCHED (a dynamic-block header with complete Huffman codes whose block decodes to its end and
// is followed by another valid header) and every chunk is decoded on its own thread from its block
// start to the next chunk's, into 16-bit symbols: 0..255 = a byte, 256 + k = "byte k of the unknown
// 32 KiB window before this chunk" (the window is simply laid out in front of the chunk's output,
// so copies need no special case and unknown bytes propagate through matches by themselves).
// Pass 2: chunk after chunk (cheap: 32 KiB each) the windows become known; then all chunks are
// translated to bytes in parallel into the final buffer.  Members may end inside a chunk (trailer
// recorded, next header skipped).  Every member's CRC-32 and size decide: anything unexpected (no
// block start found, a chunk that does not end exactly on the next start, a mismatch) makes the
// caller decode serially instead.
// ------------------------------------------------------------------------------------------
constexpr size_t WIN = 32768;

inline size_t bit_position(const Bits &b, const uint8_t *base) { return 8 * (size_t)(b.p - base) - b.cnt; }

void start_at_bit(Bits &b, const uint8_t *in, size_t n_in, size_t bitpos) {
  b = Bits();
  b.p = in + (bitpos >> 3);
  b.end = in + n_in;
  if (bitpos & 7) { refill_safe(b); consume(b, (unsigned)(bitpos & 7)); }
}

struct Out16 {
  uint16_t *base = nullptr;      // WIN placeholder symbols, then the chunk's output
  size_t cap = 0, pos = 0;
  size_t limit = ~(size_t)0;     // symbols a chunk may grow to (memory bound of the parallel path)
  ~Out16() { free(base); }
  bool reserve(size_t extra) {
    if (cap - pos >= extra) return true;
    size_t want = cap + cap / 2 + extra;
    if (want > limit) return false;
    uint16_t *p = (uint16_t *)realloc(base, want * 2);
    if (!p) return false;
    base = p;
    cap = want;
    return true;
  }
  bool init(size_t guess) {
    cap = WIN + guess + 1024;
    base = (uint16_t *)malloc(cap * 2);
    if (!base) return false;
    for (size_t k = 0; k < WIN; k++) base[k] = (uint16_t)(256 + k);
    pos = WIN;
    return true;
  }
};

// one Huffman block into 16-bit symbols (see above); R_EOB or R_ERROR
int decode_huffman16(Decoder &d, Out16 &o) {
  const uint32_t *const lt = d.lt.data(), *const dt = d.dt.data();
  make_clean(d.in);
  uint64_t buf = d.in.buf;
  unsigned cnt = d.in.cnt;
  const uint8_t *ip = d.in.p;
  const uint8_t *const iend = d.in.end;
  size_t phantom = d.in.phantom;
  int rc = R_ERROR;
  for (;;) {
    if (o.cap - o.pos < 600 && !o.reserve(1u << 20)) break;
    uint16_t *op = o.base + o.pos;
    if (iend - ip >= 16) {
      buf |= load64(ip) << cnt;
      ip += (63 - cnt) >> 3;
      cnt |= 56;
    } else {
      buf &= (cnt >= 64) ? ~0ull : ((1ull << cnt) - 1);
      while (cnt <= 56) {
        uint64_t byte = 0;
        if (ip < iend) byte = *ip++; else phantom++;
        buf |= byte << cnt;
        cnt += 8;
      }
      if (phantom > 8) break;
    }
    uint32_t e = lt[buf & ((1u << LL_TB) - 1)];
    unsigned k = 0;
    while (k < 4 && (e & (F_LIT | F_SUB)) == F_LIT) {       // up to four look-ups of one or two literals
      buf >>= (e & 0xff);
      cnt -= (e & 0xff);
      op[0] = (uint16_t)((e >> 16) & 0xff);
      op[1] = (uint16_t)(e >> 24);
      op += 1 + ((e >> 12) & 1);
      e = lt[buf & ((1u << LL_TB) - 1)];
      k++;
    }
    o.pos = (size_t)(op - o.base);
    if (k) continue;
    if (e & F_SUB) {
      buf >>= LL_TB;
      cnt -= LL_TB;
      e = lt[(e >> 16) + (buf & ((1u << ((e >> 8) & 15)) - 1))];
    }
    const uint64_t saved = buf;
    buf >>= (e & 0xff);
    cnt -= (e & 0xff);
    if (e & F_LIT) { *op++ = (uint16_t)((e >> 16) & 0xff); o.pos++; continue; }
    if (e & F_EOB) { rc = R_EOB; break; }
    if ((e >> 16) == 0 || (e & 0xff) == 0) break;
    const unsigned lcode = (e >> 8) & 15, ltot = e & 0xff;
    const unsigned len = (e >> 16) + (unsigned)((saved >> lcode) & ((1u << (ltot - lcode)) - 1));
    uint32_t f = dt[buf & ((1u << D_TB) - 1)];
    if (f & F_SUB) {
      buf >>= D_TB;
      cnt -= D_TB;
      f = dt[(f >> 16) + (buf & ((1u << ((f >> 8) & 15)) - 1))];
    }
    const uint64_t saved2 = buf;
    buf >>= (f & 0xff);
    cnt -= (f & 0xff);
    if ((f >> 16) == 0 || (f & 0xff) == 0) break;
    const unsigned dcode = (f >> 8) & 15, dtot = f & 0xff;
    const size_t dist = (f >> 16) + (size_t)((saved2 >> dcode) & ((1u << (dtot - dcode)) - 1));
    if (dist > o.pos) break;                                 // before even the window
    const uint16_t *src = op - dist;
    if (dist >= 4) {                                         // four symbols per copy, overshoot < 4
      uint16_t *const stop = op + len;
      do { memcpy(op, src, 8); op += 4; src += 4; } while (op < stop);
    } else {
      for (unsigned i = 0; i < len; i++) op[i] = src[i];
    }
    o.pos += len;
  }
  d.in.buf = buf;
  d.in.cnt = cnt;
  d.in.p = ip;
  d.in.phantom = phantom;
  d.in.clean = false;
  return rc;
}

size_t parse_gzip_header(const uint8_t *in, size_t n_in, size_t ip) {   // -> first deflate byte, 0 = bad
  if (n_in - ip < 18 || in[ip] != 0x1f || in[ip + 1] != 0x8b || in[ip + 2] != 8) return 0;
  const unsigned flg = in[ip + 3];
  if (flg & 0xe0) return 0;
  size_t q = ip + 10;
  if (flg & 4) { if (q + 2 > n_in) return 0; q += 2 + (size_t)(in[q] | (in[q + 1] << 8)); }
  if (flg & 8) { while (q < n_in && in[q]) q++; q++; }
  if (flg & 16) { while (q < n_in && in[q]) q++; q++; }
  if (flg & 2) q += 2;
  return q < n_in ? q : 0;
}

struct MemberEnd { size_t out_pos; uint32_t crc, isize; };   // out_pos: symbols of the chunk before the member's end

// blocks from the decoder's position up to (exactly) stop_bit, or, stop_bit == 0, to the end of the
// input.  A member that ends on the way is recorded (trailer read, next header skipped) and decoding
// goes on in the next member.  *at_end: the input ended right after a member.
bool decode_blocks16(Decoder &d, const uint8_t *in, size_t n_in, Out16 &o, size_t stop_bit,
                     std::vector<MemberEnd> &ends, bool *at_end) {
  *at_end = false;
  for (;;) {
    if (stop_bit) {
      const size_t bp = bit_position(d.in, in);
      if (bp == stop_bit) return true;
      if (bp > stop_bit) return false;
    }
    const unsigned bfinal = take(d.in, 1), btype = take(d.in, 2);
    if (d.in.phantom) return false;
    if (btype == 0) {
      const uint8_t *s = rewind_to_bytes(d.in);
      if (d.in.end - s < 4) return false;
      const unsigned len = s[0] | (s[1] << 8), nlen = s[2] | (s[3] << 8);
      if ((len ^ nlen) != 0xffff || (size_t)(d.in.end - s) < 4 + (size_t)len) return false;
      if (!o.reserve((size_t)len + 8)) return false;
      for (unsigned k = 0; k < len; k++) o.base[o.pos + k] = s[4 + k];
      o.pos += len;
      d.in.p = s + 4 + len;
    } else if (btype == 1 || btype == 2) {
      if (btype == 1 ? !fixed_tables(d) : !read_dynamic_tables(d)) return false;
      if (decode_huffman16(d, o) != R_EOB) return false;
    } else {
      return false;
    }
    if (bfinal) {
      const uint8_t *s = rewind_to_bytes(d.in);
      if (d.in.phantom || d.in.end - s < 8) return false;
      MemberEnd me;
      me.out_pos = o.pos - WIN;
      me.crc = (uint32_t)s[0] | ((uint32_t)s[1] << 8) | ((uint32_t)s[2] << 16) | ((uint32_t)s[3] << 24);
      me.isize = (uint32_t)s[4] | ((uint32_t)s[5] << 8) | ((uint32_t)s[6] << 16) | ((uint32_t)s[7] << 24);
      ends.push_back(me);
      const size_t after = (size_t)(s + 8 - in);
      if (after == n_in) { *at_end = true; return stop_bit == 0; }
      const size_t q = parse_gzip_header(in, n_in, after);            // the next member
      if (!q) return false;
      if (stop_bit && q * 8 > stop_bit) return false;
      d.in.p = in + q;
    }
  }
}

// first bit position in [from_bit, to_bit) that looks like the start of a dynamic block: strict
// header, the block decodes to its end-of-block, and another well-formed block follows
size_t find_block_start(const uint8_t *in, size_t n_in, size_t from_bit, size_t to_bit) {
  Decoder d;
  for (size_t bp = from_bit; bp < to_bit; bp++) {
    // cheap filter: BTYPE = 2 (bits 01 in stream order after BFINAL), HLIT <= 29, HDIST <= 29
    const size_t by = bp >> 3;
    if (by + 4 >= n_in) break;
    const uint32_t w = ((uint32_t)in[by] | ((uint32_t)in[by + 1] << 8) | ((uint32_t)in[by + 2] << 16) |
                        ((uint32_t)in[by + 3] << 24)) >> (bp & 7);
    if ((w & 6) != 4 || ((w >> 3) & 31) > 29 || ((w >> 8) & 31) > 29) continue;
    const bool cand_final = (w & 1) != 0;
    start_at_bit(d.in, in, n_in, bp + 3);
    if (!read_dynamic_tables(d, true) || d.in.phantom) continue;
    Out16 scratch;
    if (!scratch.init(1u << 16)) return 0;
    if (decode_huffman16(d, scratch) != R_EOB || d.in.phantom) continue;
    if (scratch.pos - WIN < 256) continue;                    // real blocks of a large file are not tiny
    if (cand_final) {
      // a member's last block: its trailer and the next member's header (or the end) must follow
      const uint8_t *s2 = rewind_to_bytes(d.in);
      if (d.in.phantom || d.in.end - s2 < 8) continue;
      const size_t after = (size_t)(s2 + 8 - in);
      if ((uint32_t)(s2[4] | (s2[5] << 8) | (s2[6] << 16) | ((uint32_t)s2[7] << 24)) < scratch.pos - WIN) continue;
      if (after != n_in && !parse_gzip_header(in, n_in, after)) continue;
      return bp;
    }
    // the following block header
    const unsigned nb_final = take(d.in, 1), nb_type = take(d.in, 2);
    (void)nb_final;
    if (d.in.phantom || nb_type == 3) continue;
    if (nb_type == 2 && !read_dynamic_tables(d, true)) continue;
    if (nb_type == 0) {
      const uint8_t *s2 = rewind_to_bytes(d.in);
      if (d.in.end - s2 < 4 || (((unsigned)(s2[0] | (s2[1] << 8)) ^ (unsigned)(s2[2] | (s2[3] << 8))) != 0xffff)) continue;
    }
    return bp;
  }
  return 0;
}

uint32_t parallel_crc32(const uint8_t *data, size_t n, unsigned threads) {
  const size_t chunk = 8u << 20;
  const size_t np = (n + chunk - 1) / chunk;
  if (np == 0) return (uint32_t)crc32(0L, Z_NULL, 0);
  std::vector<uLong> part(np);
  const unsigned nt = threads < np ? (threads ? threads : 1) : (unsigned)np;
  std::vector<std::thread> pool;
  for (unsigned w = 0; w < nt; w++)
    pool.emplace_back([&, w] {
      for (size_t k = w; k < np; k += nt) {
        const size_t b = k * chunk, e = b + chunk < n ? b + chunk : n;
        part[k] = crc32(crc32(0L, Z_NULL, 0), data + b, (uInt)(e - b));
      }
    });
  for (auto &t : pool) t.join();
  uLong c = crc32(0L, Z_NULL, 0);
  for (size_t k = 0; k < np; k++) {
    const size_t b = k * chunk, e = b + chunk < n ? b + chunk : n;
    c = crc32_combine(c, part[k], (z_off_t)(e - b));
  }
  return (uint32_t)c;
}

struct Chunk16 {
  size_t start_bit = 0, stop_bit = 0;     // stop_bit 0: the last chunk (runs to the final block)
  Out16 out;
  bool ok = false;
  std::vector<uint8_t> window;            // the 32 KiB before this chunk, once known
  size_t offset = 0;                      // of its bytes in the final buffer
  std::vector<MemberEnd> ends;            // members that end inside this chunk
};

bool gunzip_parallel(const uint8_t *in, size_t n_in, GrowFn grow, void *user, size_t *n_out, unsigned threads) {
  const size_t q = parse_gzip_header(in, n_in, 0);
  if (threads > 32) threads = 32;
  // test hook: HUMID_PAR_INFLATE_CHUNK (bytes) forces the chunk size, so that small files take
  // this path too
  size_t forced = 0;
  if (const char *e = getenv("HUMID_PAR_INFLATE_CHUNK")) forced = (size_t)atol(e);
  if (!q || threads < 2 || n_in < q + 64 || (!forced && n_in - q < ((size_t)16 << 20))) return false;
  size_t ch = (n_in - q) / ((size_t)threads * 4);
  if (ch < ((size_t)2 << 20)) ch = (size_t)2 << 20;
  if (ch > ((size_t)16 << 20)) ch = (size_t)16 << 20;
  if (forced) ch = forced < 1024 ? 1024 : forced;
  const size_t n_cut = (n_in - q) / ch;                 // cuts at q + k*ch, k = 1 .. n_cut (the last may be dropped)
  std::vector<size_t> found(n_cut + 1, 0);
  {
    std::vector<std::thread> pool;
    for (unsigned w = 0; w < threads; w++)
      pool.emplace_back([&, w] {
        for (size_t k = 1 + w; k <= n_cut; k += threads) {
          const size_t from = (q + k * ch) * 8;
          size_t to = (q + (k + 1) * ch) * 8;
          if (to > (n_in - 16) * 8) to = (n_in - 16) * 8;
          if (from < to) found[k] = find_block_start(in, n_in, from, to);
        }
      });
    for (auto &t : pool) t.join();
  }
  std::vector<size_t> starts;
  starts.push_back(q * 8);
  for (size_t k = 1; k <= n_cut; k++)
    if (found[k]) starts.push_back(found[k]);
  if (starts.size() < 2) return false;
  const size_t n_chunks = starts.size();
  std::vector<Chunk16> chunks(n_chunks);
  for (size_t k = 0; k < n_chunks; k++) {
    chunks[k].start_bit = starts[k];
    chunks[k].stop_bit = k + 1 < n_chunks ? starts[k + 1] : 0;
  }
  std::vector<uint8_t> window(WIN, 0);
  bool window_known = false;                // chunk 0 has no window at all
  size_t total = 0, cap = 0;
  uint8_t *out = nullptr;
  bool fail = false;
  for (size_t w0 = 0; w0 < n_chunks && !fail; w0 += threads) {
    const size_t w1 = w0 + threads < n_chunks ? w0 + threads : n_chunks;
    {   // ---- pass 1 of this wave: every chunk on its own thread ----
      std::vector<std::thread> pool;
      for (size_t k = w0; k < w1; k++)
        pool.emplace_back([&, k] {
          Chunk16 &c = chunks[k];
          const size_t zbytes = ((c.stop_bit ? c.stop_bit : n_in * 8) - c.start_bit) / 8;
          // FastQ inflates 4-6x; a chunk that wants more than 40x (or 64 M symbols) is not worth
          // 2 bytes of memory per symbol on every thread: the serial decoder takes the file
          c.out.limit = WIN + (zbytes * 40 > ((size_t)64 << 20) ? zbytes * 40 : ((size_t)64 << 20));
          if (!c.out.init(zbytes * 5)) return;
          Decoder d;
          start_at_bit(d.in, in, n_in, c.start_bit);
          bool at_end = false;
          if (!decode_blocks16(d, in, n_in, c.out, c.stop_bit, c.ends, &at_end)) return;
          if ((c.stop_bit == 0) != at_end) return;                     // only the last chunk reaches the end
          c.ok = true;
        });
      for (auto &t : pool) t.join();
    }
    // ---- pass 2a: windows, in order ----
    size_t wave_bytes = 0;
    for (size_t k = w0; k < w1; k++) {
      Chunk16 &c = chunks[k];
      if (!c.ok) { fail = true; break; }
      c.window = window;
      c.offset = total + wave_bytes;
      const size_t n = c.out.pos - WIN;
      const uint16_t *sym = c.out.base + WIN;
      std::vector<uint8_t> next(WIN, 0);
      const size_t keep_old = n < WIN ? WIN - n : 0;            // a short chunk keeps part of the old window
      for (size_t j = 0; j < keep_old; j++) next[j] = window[WIN - keep_old + j];
      for (size_t j = n < WIN ? 0 : n - WIN; j < n; j++) {
        const uint16_t s = sym[j];
        if (s >= 256 && !window_known) { fail = true; break; }   // the first chunk refers to nothing
        next[keep_old + (j - (n < WIN ? 0 : n - WIN))] = s < 256 ? (uint8_t)s : window[s - 256];
      }
      if (fail) break;
      window.swap(next);
      window_known = true;
      wave_bytes += n;
    }
    if (fail) break;
    // ---- pass 2b: symbols -> bytes in the final buffer, all chunks of the wave at once ----
    if (total + wave_bytes > cap) {
      size_t want = total + wave_bytes;
      if (w1 < n_chunks) want += want / 2;
      char *p = grow(user, want, &cap);
      if (!p && want > total + wave_bytes) p = grow(user, total + wave_bytes, &cap);
      if (!p) { fail = true; break; }
      out = (uint8_t *)p;
    }
    {
      std::vector<std::thread> pool;
      std::vector<char> bad(w1 - w0, 0);
      for (size_t k = w0; k < w1; k++)
        pool.emplace_back([&, k] {
          Chunk16 &c = chunks[k];
          const size_t n = c.out.pos - WIN;
          const uint16_t *sym = c.out.base + WIN;
          const uint8_t *win = c.window.data();
          uint8_t *dst = out + c.offset;
          const bool first = k == 0;
          unsigned seen = 0;
          for (size_t j = 0; j < n; j++) {
            const uint16_t s = sym[j];
            seen |= s;
            dst[j] = s < 256 ? (uint8_t)s : win[s - 256];
          }
          if (first && seen >= 256) bad[k - w0] = 1;
          free(c.out.base);
          c.out.base = nullptr;
          c.out.cap = c.out.pos = 0;
          std::vector<uint8_t>().swap(c.window);
        });
      for (auto &t : pool) t.join();
      for (char b : bad) fail = fail || b;
    }
    total += wave_bytes;
  }
  if (fail) return false;
  // ---- every member's size and CRC-32 (members on all cores; a single member in pieces) ----
  struct Span { size_t b, e; uint32_t crc, isize; };
  std::vector<Span> spans;
  size_t mb = 0;
  for (const Chunk16 &c : chunks)
    for (const MemberEnd &me : c.ends) {
      spans.push_back(Span{mb, c.offset + me.out_pos, me.crc, me.isize});
      mb = c.offset + me.out_pos;
    }
  if (spans.empty() || mb != total) return false;
  for (const Span &sp : spans)
    if (sp.e < sp.b || (uint32_t)(sp.e - sp.b) != sp.isize) return false;
  if (spans.size() == 1) {
    if (parallel_crc32(out, total, threads) != spans[0].crc) return false;
  } else {
    std::vector<char> bad(threads, 0);
    std::vector<std::thread> pool;
    for (unsigned w = 0; w < threads; w++)
      pool.emplace_back([&, w] {
        for (size_t k = w; k < spans.size(); k += threads) {
          uLong c = crc32(0L, Z_NULL, 0);
          for (size_t o2 = spans[k].b; o2 < spans[k].e; o2 += (1u << 30)) {
            const size_t n = spans[k].e - o2 < (1u << 30) ? spans[k].e - o2 : (1u << 30);
            c = crc32(c, out + o2, (uInt)n);
          }
          if ((uint32_t)c != spans[k].crc) bad[w] = 1;
        }
      });
    for (auto &t : pool) t.join();
    for (char b : bad) if (b) return false;
  }
  *n_out = total;
  return true;
}

}  // namespace

bool fast_gunzip(const uint8_t *in, size_t n_in, GrowFn grow, void *user, size_t *n_out, unsigned threads) {
  *n_out = 0;
  if (getenv("HUMID_SERIAL_INFLATE") == nullptr && gunzip_parallel(in, n_in, grow, user, n_out, threads)) {
    if (getenv("HUMID_TIMING")) std::fprintf(stderr, "[humid] gzip member inflated by the parallel two-pass decoder\n");
    return true;
  }
  *n_out = 0;
  size_t cap = 0, pos = 0;
  uint8_t *out = nullptr;
  auto need = [&](size_t extra) -> bool {                 // at least `extra` free bytes behind pos
    if (out && cap - pos >= extra) return true;
    size_t want = pos + extra;
    if (want < cap + cap / 2) want = cap + cap / 2;
    if (want < n_in * 4) want = n_in * 4;
    char *p = grow(user, want, &cap);
    if (!p && want > pos + extra) p = grow(user, pos + extra, &cap);   // the bound may still allow the minimum
    if (!p) return false;
    out = (uint8_t *)p;
    return cap - pos >= extra;
  };
  std::vector<Member> members;
  Decoder d;
  size_t ip = 0;
  while (ip < n_in) {
    // ---- gzip member header (RFC 1952) ----
    if (n_in - ip < 18 || in[ip] != 0x1f || in[ip + 1] != 0x8b || in[ip + 2] != 8) return false;
    const unsigned flg = in[ip + 3];
    if (flg & 0xe0) return false;
    size_t q = ip + 10;
    if (flg & 4) { if (q + 2 > n_in) return false; q += 2 + (size_t)(in[q] | (in[q + 1] << 8)); }
    if (flg & 8) { while (q < n_in && in[q]) q++; q++; }
    if (flg & 16) { while (q < n_in && in[q]) q++; q++; }
    if (flg & 2) q += 2;
    if (q >= n_in) return false;
    d.in = Bits();
    d.in.p = in + q;
    d.in.end = in + n_in;
    const size_t member_begin = pos;
    // ---- deflate blocks ----
    for (;;) {
      const unsigned bfinal = take(d.in, 1), btype = take(d.in, 2);
      if (d.in.phantom > 8) return false;
      if (btype == 0) {
        const uint8_t *s = rewind_to_bytes(d.in);
        if (d.in.phantom || d.in.end - s < 4) return false;
        const unsigned len = s[0] | (s[1] << 8), nlen = s[2] | (s[3] << 8);
        if ((len ^ nlen) != 0xffff || (size_t)(d.in.end - s) < 4 + (size_t)len) return false;
        if (!need((size_t)len + 1)) return false;
        memcpy(out + pos, s + 4, len);
        pos += len;
        d.in.p = s + 4 + len;
      } else if (btype == 1 || btype == 2) {
        if (btype == 1 ? !fixed_tables(d) : !read_dynamic_tables(d)) return false;
        for (;;) {
          if (!need(1u << 16)) return false;
          int rc = (d.in.end - d.in.p >= 16) ? decode_huffman<true>(d, out, pos, cap, member_begin)
                                             : decode_huffman<false>(d, out, pos, cap, member_begin);
          if (rc == R_EOB) break;
          if (rc == R_ERROR) return false;
          if (rc == R_INPUT_TAIL) {
            if (!need(1u << 16)) return false;
            rc = decode_huffman<false>(d, out, pos, cap, member_begin);
            if (rc == R_EOB) break;
            if (rc == R_ERROR) return false;
          }
          // R_NEED_OUT: grow and go on
          if (!need(1u << 20)) return false;
        }
      } else {
        return false;
      }
      if (bfinal) break;
    }
    // ---- trailer: CRC-32 and ISIZE ----
    const uint8_t *s = rewind_to_bytes(d.in);
    if (d.in.phantom || s > d.in.end || d.in.end - s < 8) return false;   // consumed bits that were not there
    const uint32_t crc = (uint32_t)s[0] | ((uint32_t)s[1] << 8) | ((uint32_t)s[2] << 16) | ((uint32_t)s[3] << 24);
    const uint32_t isize = (uint32_t)s[4] | ((uint32_t)s[5] << 8) | ((uint32_t)s[6] << 16) | ((uint32_t)s[7] << 24);
    if (isize != (uint32_t)(pos - member_begin)) return false;
    members.push_back(Member{member_begin, pos, crc});
    ip = (size_t)(s + 8 - in);
  }
  // ---- CRC-32 of every member: chunks on all cores, combined in order ----
  if (threads == 0) threads = 1;
  const size_t chunk = 8u << 20;
  struct Piece { size_t b, e; uLong crc; };
  std::vector<Piece> pieces;
  std::vector<size_t> first_piece;
  for (const Member &m : members) {
    first_piece.push_back(pieces.size());
    for (size_t b0 = m.begin; b0 < m.end; b0 += chunk) pieces.push_back(Piece{b0, b0 + chunk < m.end ? b0 + chunk : m.end, 0});
  }
  first_piece.push_back(pieces.size());
  if (!pieces.empty()) {
    std::vector<std::thread> pool;
    const unsigned nt = threads < pieces.size() ? threads : (unsigned)pieces.size();
    for (unsigned w = 0; w < nt; w++)
      pool.emplace_back([&, w] {
        for (size_t k = w; k < pieces.size(); k += nt) {
          Piece &pc = pieces[k];
          uLong c = crc32(0L, Z_NULL, 0);
          for (size_t o = pc.b; o < pc.e; o += (1u << 30)) {
            const size_t n = pc.e - o < (1u << 30) ? pc.e - o : (1u << 30);
            c = crc32(c, out + o, (uInt)n);
          }
          pc.crc = c;
        }
      });
    for (auto &t : pool) t.join();
  }
  for (size_t mi = 0; mi < members.size(); mi++) {
    uLong c = crc32(0L, Z_NULL, 0);
    for (size_t k = first_piece[mi]; k < first_piece[mi + 1]; k++)
      c = crc32_combine(c, pieces[k].crc, (z_off_t)(pieces[k].e - pieces[k].b));
    if ((uint32_t)c != members[mi].crc) return false;
  }
  *n_out = pos;
  return true;
}

}  // namespace humid_host
