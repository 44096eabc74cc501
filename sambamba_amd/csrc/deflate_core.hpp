// deflate_core.hpp -- one BGZF block's worth of raw DEFLATE (RFC 1951), written once for the device (deflate.hip: one lane per
// BGZF block) and for the host (the CPU unit test compiles this header with g++ and inflates the result with zlib).
//
// Replaces bgzfCompress (BioD/bio/core/bgzf/compress.d:34-103: zlib deflateInit2(level, Z_DEFLATED, -15, 8) + deflate(Z_FINISH)
// + crc32 per <= 0xFF00-byte block).  Any valid deflate stream of the block is acceptable to every BGZF reader -- the
// reference inflates with zlib (block.d:158-185) --, so the format is pinned by RFC 1951 / the SAM specification, not by
// zlib's bit-exact output.  `level` is honoured the way zlib's is -- more work for fewer bytes -- in three modes:
//   level 0            one stored block;
//   levels 1 .. 3      the FIXED Huffman code over greedy matches of >= 4 bytes found through a 2^kHashBits-entry table of last
//                      positions (one pass over the input, no lazy evaluation);
//   levels 4 .. 9, -1  a DYNAMIC Huffman code (RFC 1951 3.2.7) over the same matches: one pass counts the symbols, the code
//                      lengths come from Moffat & Katajainen's in-place minimum-redundancy construction limited to 15 bits, a
//                      second pass over the input emits the tokens (the matcher is deterministic, so both passes see the same
//                      tokens and no token buffer is needed); the fixed code is kept when its stream is shorter.
// Every choice is a function of the input alone, so the same input gives the same bytes on the device and on the host.  An
// incompressible block falls back to one stored block.
#pragma once
#include <stdint.h>
#include <string.h>

#ifndef SBX_HD
#if defined(__HIPCC__)
#define SBX_HD __host__ __device__
#else
#define SBX_HD
#endif
#endif

namespace sbx {

constexpr int kHashBits = 11;                       // 2048 entries x u16 = 4 KiB of scratch per block being compressed
constexpr uint32_t kBgzfPayload = 0xFF00;           // payload bytes per BGZF block (bgzf/constants.d:33)
constexpr uint32_t kBgzfSlot = 65536;               // bytes reserved per compressed block (a BGZF block is at most 64 KiB)
constexpr uint32_t kMinMatch = 4, kMaxMatch = 258, kMaxDist = 32768;
constexpr int kDynamicFromLevel = 4;                // levels >= this one (and -1, zlib's default = 6) build a dynamic Huffman code
constexpr int kThoroughFromLevel = 7;               // levels >= this one look at four candidates per position and evaluate lazily
SBX_HD inline bool level_is_dynamic(int level) { return level == -1 || level >= kDynamicFromLevel; }

// Working memory of the dynamic mode, one per block being compressed (device: a 4 KiB slice of global scratch per lane).
constexpr uint32_t kDistBase = 288;                 // the distance alphabet sits behind the literal/length alphabet in the arrays below
constexpr uint32_t kNumLitLen = 286, kNumDist = 30, kNumCl = 19;
struct DynWork {
    uint16_t freq[320];                             // symbol counts: [0, 286) literal/length, [288, 318) distance
    uint16_t code[320];                             // bit-reversed canonical codes
    uint8_t len[320];                               // code lengths
    uint32_t a[288];                                // sort keys, then the working array of the minimum-redundancy construction
    uint16_t sym[288];                              // symbols in the order of `a`
    uint16_t cl_freq[20], cl_code[20];              // the code length alphabet (RFC 1951 3.2.7)
    uint8_t cl_len[20];
};
constexpr uint32_t kWorkBytes = 4096;
static_assert(sizeof(DynWork) <= kWorkBytes, "DynWork must fit its slice");

struct BitSink {
    uint8_t* out;
    uint32_t pos, cap;
    uint64_t acc;
    uint32_t nbits;
    bool overflow;
    SBX_HD void init(uint8_t* o, uint32_t c) { out = o; pos = 0; cap = c; acc = 0; nbits = 0; overflow = false; }
    SBX_HD void put(uint32_t bits, uint32_t n) {      // n <= 32, LSB first
        acc |= (uint64_t)bits << nbits;
        nbits += n;
        if (nbits >= 32) {
            if (pos + 4 <= cap) {
                const uint32_t w = (uint32_t)acc;
                out[pos] = (uint8_t)w; out[pos + 1] = (uint8_t)(w >> 8); out[pos + 2] = (uint8_t)(w >> 16); out[pos + 3] = (uint8_t)(w >> 24);
            } else overflow = true;
            pos += 4;
            acc >>= 32;
            nbits -= 32;
        }
    }
    SBX_HD uint32_t finish() {                         // pads to a byte boundary; returns the number of bytes
        while (nbits > 0) {
            if (pos < cap) out[pos] = (uint8_t)acc; else overflow = true;
            ++pos;
            acc >>= 8;
            nbits = nbits > 8 ? nbits - 8 : 0;
        }
        return pos;
    }
};

SBX_HD inline uint32_t rev_bits(uint32_t v, uint32_t n) {     // Huffman codes are packed most-significant bit first (RFC 1951 3.1.1)
    uint32_t r = 0;
    for (uint32_t i = 0; i < n; ++i) { r = (r << 1) | (v & 1u); v >>= 1; }
    return r;
}

// literal / end-of-block / length symbol 0..287 of the fixed code (RFC 1951 3.2.6)
SBX_HD inline void put_litlen(BitSink& s, uint32_t sym) {
    if (sym < 144) s.put(rev_bits(0x30 + sym, 8), 8);
    else if (sym < 256) s.put(rev_bits(0x190 + (sym - 144), 9), 9);
    else if (sym < 280) s.put(rev_bits(sym - 256, 7), 7);
    else s.put(rev_bits(0xC0 + (sym - 280), 8), 8);
}

// length 3..258 -> symbol 257..285 + extra bits (RFC 1951 3.2.5)
SBX_HD inline void put_length(BitSink& s, uint32_t len) {
    if (len == 258) { put_litlen(s, 285); return; }
    const uint32_t l = len - 3;                                  // 0..254
    if (l < 8) { put_litlen(s, 257 + l); return; }
    uint32_t e = 0;                                              // extra bits: floor(log2(l)) - 2
    for (uint32_t t = l >> 3; t; t >>= 1) ++e;
    const uint32_t base_sym = 261 + 4 * e, first = (4u << e);    // l in [4 << e, 8 << e)
    const uint32_t idx = (l - first) >> e;
    put_litlen(s, base_sym + idx);
    s.put((l - first) & ((1u << e) - 1u), e);
}

// distance 1..32768 -> 5-bit symbol 0..29 + extra bits
SBX_HD inline void put_distance(BitSink& s, uint32_t dist) {
    const uint32_t d = dist - 1;
    if (d < 4) { s.put(rev_bits(d, 5), 5); return; }
    uint32_t e = 0;                                              // extra bits: floor(log2(d)) - 1
    for (uint32_t t = d >> 2; t; t >>= 1) ++e;
    const uint32_t first = (2u << e);                            // d in [2 << e, 4 << e)
    const uint32_t sym = 2 * e + 2 + ((d - first) >> e);
    s.put(rev_bits(sym, 5), 5);
    s.put((d - first) & ((1u << e) - 1u), e);
}

SBX_HD inline uint32_t load32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

// Raw deflate of in[0, n) (n <= 65535) into out[0, cap): returns the number of bytes, or 0 when it does not fit.
// table: 1 << kHashBits entries, all zero on entry (entry = position + 1 of the last occurrence of a 4-byte hash).
SBX_HD inline uint32_t deflate_fixed(const uint8_t* in, uint32_t n, uint8_t* out, uint32_t cap, uint16_t* table) {
    BitSink s;
    s.init(out, cap);
    s.put(1, 1);           // BFINAL
    s.put(1, 2);           // BTYPE = 01: fixed Huffman codes
    uint32_t i = 0;
    while (i < n) {
        uint32_t best = 0, dist = 0;
        if (i + kMinMatch <= n) {
            const uint32_t x = load32(in + i);
            const uint32_t h = (x * 2654435761u) >> (32 - kHashBits);
            const uint32_t cand1 = table[h];
            table[h] = (uint16_t)(i + 1);
            if (cand1 != 0) {
                const uint32_t c = cand1 - 1;
                if (i - c <= kMaxDist && load32(in + c) == x) {
                    uint32_t l = 4;
                    const uint32_t lim = n - i < kMaxMatch ? n - i : kMaxMatch;
                    while (l < lim && in[c + l] == in[i + l]) ++l;
                    best = l;
                    dist = i - c;
                }
            }
        }
        if (best >= kMinMatch) {
            put_length(s, best);
            put_distance(s, dist);
            // the second position of the match is hashed too (cheap, and it is what keeps runs and record-to-record copies chained)
            if (i + 1 + kMinMatch <= n) {
                const uint32_t h1 = (load32(in + i + 1) * 2654435761u) >> (32 - kHashBits);
                table[h1] = (uint16_t)(i + 2);
            }
            i += best;
        } else {
            put_litlen(s, in[i]);
            ++i;
        }
    }
    put_litlen(s, 256);    // end of block
    const uint32_t bytes = s.finish();
    return s.overflow ? 0u : bytes;
}

// ---- dynamic Huffman codes -------------------------------------------------------------------------------------------------
// length 3..258 -> symbol 257..285, number and value of its extra bits (RFC 1951 3.2.5; the arithmetic of put_length)
SBX_HD inline uint32_t length_symbol(uint32_t len, uint32_t* eb, uint32_t* ev) {
    *eb = 0; *ev = 0;
    if (len == 258) return 285;
    const uint32_t l = len - 3;
    if (l < 8) return 257 + l;
    uint32_t e = 0;
    for (uint32_t t = l >> 3; t; t >>= 1) ++e;
    const uint32_t first = 4u << e;
    *eb = e; *ev = (l - first) & ((1u << e) - 1u);
    return 261 + 4 * e + ((l - first) >> e);
}
// distance 1..32768 -> symbol 0..29, number and value of its extra bits
SBX_HD inline uint32_t distance_symbol(uint32_t dist, uint32_t* eb, uint32_t* ev) {
    *eb = 0; *ev = 0;
    const uint32_t d = dist - 1;
    if (d < 4) return d;
    uint32_t e = 0;
    for (uint32_t t = d >> 2; t; t >>= 1) ++e;
    const uint32_t first = 2u << e;
    *eb = e; *ev = (d - first) & ((1u << e) - 1u);
    return 2 * e + 2 + ((d - first) >> e);
}

// The matcher as a token source: emit.literal(byte) / emit.match(length, distance) for in[0, n).  table: all zero on entry.
// Deterministic, so two calls over the same input (and a zeroed table) produce the same tokens.  The table's 2^kHashBits entries
// (position + 1 of an earlier occurrence of a 4-byte hash) are buckets of kWays entries, newest first -- kWays = 4: one 8-byte
// load brings the four candidates of a position --; kInsert: the first kInsert positions of a match are entered into the table
// too (deflate_fixed enters two); kLazy: a match shorter than kLazyBelow is dropped for a literal when the next position has a
// longer one (zlib's lazy evaluation, deflate.c deflate_slow, without its chain).
constexpr uint32_t kLazyBelow = 32;
template <int kWays>
struct MatchTable {
    static constexpr uint32_t kBits = kHashBits - (kWays == 4 ? 2 : kWays == 2 ? 1 : 0);
    const uint8_t* in;
    uint32_t n;
    uint16_t* table;
    SBX_HD uint16_t* bucket(uint32_t x) const { return table + ((x * 2654435761u) >> (32 - kBits)) * kWays; }
    SBX_HD void insert(uint32_t i) const {
        if (i + kMinMatch > n) return;
        uint16_t* b = bucket(load32(in + i));
        if (kWays == 4) {                               // (a bucket of four is one aligned 8-byte word, entry 0 in its low bits)
            uint64_t v;
            memcpy(&v, __builtin_assume_aligned(b, 8), 8);
            v = (v << 16) | (uint64_t)(uint16_t)(i + 1);
            memcpy(__builtin_assume_aligned(b, 8), &v, 8);
        } else {
            for (int w = kWays - 1; w > 0; --w) b[w] = b[w - 1];
            b[0] = (uint16_t)(i + 1);
        }
    }
    // longest match of >= kMinMatch bytes at i among the bucket's candidates (the newest wins a tie); 0 when there is none
    SBX_HD uint32_t find(uint32_t i, uint32_t* dist, bool enter) const {
        uint32_t best = 0;
        *dist = 0;
        if (i + kMinMatch > n) return 0;
        const uint32_t x = load32(in + i);
        uint16_t* b = bucket(x);
        uint16_t cand[kWays];
        if (kWays == 4) {
            uint64_t v;
            memcpy(&v, __builtin_assume_aligned(b, 8), 8);
            for (int w = 0; w < kWays; ++w) cand[w] = (uint16_t)(v >> (16 * w));
            if (enter) {
                v = (v << 16) | (uint64_t)(uint16_t)(i + 1);
                memcpy(__builtin_assume_aligned(b, 8), &v, 8);
            }
        } else {
            for (int w = 0; w < kWays; ++w) cand[w] = b[w];
            if (enter) {
                for (int w = kWays - 1; w > 0; --w) b[w] = cand[w - 1];
                b[0] = (uint16_t)(i + 1);
            }
        }
        const uint32_t lim = n - i < kMaxMatch ? n - i : kMaxMatch;
        for (int w = 0; w < kWays; ++w) {
            if (!cand[w]) continue;
            const uint32_t c = cand[w] - 1u;
            if (c >= i || i - c > kMaxDist || load32(in + c) != x) continue;
            uint32_t l = 4;
            while (l < lim && in[c + l] == in[i + l]) ++l;
            if (l > best) { best = l; *dist = i - c; }
        }
        return best;
    }
};

template <int kWays, bool kLazy, uint32_t kInsert, class Emit>
SBX_HD inline void lz77_tokens(const uint8_t* in, uint32_t n, uint16_t* table, Emit& emit) {
    const MatchTable<kWays> t{in, n, table};
    uint32_t i = 0;
    while (i < n) {
        uint32_t dist = 0;
        const uint32_t best = t.find(i, &dist, true);
        if (best >= kMinMatch) {
            if (kLazy && best < kLazyBelow && i + 1 < n) {
                uint32_t d2 = 0;
                if (t.find(i + 1, &d2, false) > best) { emit.literal(in[i]); ++i; continue; }
            }
            emit.match(best, dist);
            const uint32_t ins = best < kInsert ? best : kInsert;
            for (uint32_t j = 1; j < ins; ++j) t.insert(i + j);
            i += best;
        } else {
            emit.literal(in[i]);
            ++i;
        }
    }
}

struct CountTokens {
    uint16_t* freq;
    SBX_HD void literal(uint32_t b) { ++freq[b]; }
    SBX_HD void match(uint32_t len, uint32_t dist) {
        uint32_t eb, ev;
        ++freq[length_symbol(len, &eb, &ev)];
        ++freq[kDistBase + distance_symbol(dist, &eb, &ev)];
    }
};
struct WriteTokens {
    BitSink* s;
    const uint16_t* code;
    const uint8_t* len;
    SBX_HD void literal(uint32_t b) { s->put(code[b], len[b]); }
    SBX_HD void match(uint32_t length, uint32_t dist) {
        uint32_t eb, ev;
        const uint32_t ls = length_symbol(length, &eb, &ev);
        s->put((uint32_t)code[ls] | (ev << len[ls]), len[ls] + eb);            // <= 15 + 5 bits
        const uint32_t ds = kDistBase + distance_symbol(dist, &eb, &ev);
        s->put((uint32_t)code[ds] | (ev << len[ds]), len[ds] + eb);            // <= 15 + 13 bits
    }
};

// a prefix code needs two symbols to be complete (zlib's inflate, like the RFC, wants complete codes apart from a lone 1-bit
// distance code): symbols 0 / 1 are given a count of one until two symbols are in use
SBX_HD inline void at_least_two_symbols(uint16_t* freq, uint32_t n_sym) {
    uint32_t used = 0;
    for (uint32_t s = 0; s < n_sym; ++s) used += freq[s] != 0;
    for (uint32_t s = 0; used < 2 && s < n_sym; ++s)
        if (!freq[s]) { freq[s] = 1; ++used; }
}

// Code lengths (<= max_len bits, max_len <= 15) of a minimum-redundancy prefix code for freq[0, n_sym) (at least two of them
// nonzero); len[s] = 0 for unused symbols.  Sorted by (count, symbol) -- a total order, so the result does not depend on the sort
// --, then Moffat & Katajainen's in-place construction ("In-place calculation of minimum-redundancy codes", WADS 1995: three
// sweeps over the sorted counts -- internal node weights, then depths of internal nodes, then depths of the leaves), then the usual
// length limit: lengths above max_len are cut to it and the Kraft sum is paid back by lengthening the cheapest shorter codes.
// a, sym: scratch for n_sym entries.
SBX_HD inline void huffman_lengths(const uint16_t* freq, uint32_t n_sym, uint32_t max_len, uint8_t* len, uint32_t* a, uint16_t* sym) {
    uint32_t m = 0;
    for (uint32_t s = 0; s < n_sym; ++s) {
        len[s] = 0;
        if (freq[s]) a[m++] = ((uint32_t)freq[s] << 9) | s;
    }
    // shell sort, ascending (gaps of Ciura's sequence)
    const uint32_t gaps[7] = {132, 57, 23, 10, 4, 1, 0};
    for (uint32_t g = 0; gaps[g]; ++g) {
        const uint32_t gap = gaps[g];
        for (uint32_t i = gap; i < m; ++i) {
            const uint32_t v = a[i];
            uint32_t j = i;
            for (; j >= gap && a[j - gap] > v; j -= gap) a[j] = a[j - gap];
            a[j] = v;
        }
    }
    for (uint32_t i = 0; i < m; ++i) { sym[i] = (uint16_t)(a[i] & 511u); a[i] >>= 9; }
    if (m == 2) { a[0] = 1; a[1] = 1; }
    else {
        // sweep 1: a[next] = weight of internal node `next`, children taken from the two queues (leaves from `leaf`, internal nodes
        // from `root`); a consumed internal node stores its parent's index
        a[0] += a[1];
        uint32_t root = 0, leaf = 2;
        for (uint32_t next = 1; next + 1 < m; ++next) {
            if (leaf >= m || a[root] < a[leaf]) { a[next] = a[root]; a[root++] = next; }
            else a[next] = a[leaf++];
            if (leaf >= m || (root < next && a[root] < a[leaf])) { a[next] += a[root]; a[root++] = next; }
            else a[next] += a[leaf++];
        }
        // sweep 2: parent indices -> depths of the internal nodes (the root, node m - 2, has depth 0)
        a[m - 2] = 0;
        for (uint32_t next = m - 2; next-- > 0;) a[next] = a[a[next]] + 1;
        // sweep 3: depths of the leaves, deepest (least frequent) first in a[0 ..]
        int32_t avbl = 1, used = 0, dpth = 0;
        int32_t r = (int32_t)m - 2, nx = (int32_t)m - 1;
        while (avbl > 0) {
            while (r >= 0 && (int32_t)a[r] == dpth) { ++used; --r; }
            while (avbl > used) { a[nx--] = (uint32_t)dpth; --avbl; }
            avbl = 2 * used; ++dpth; used = 0;
        }
    }
    // length limit
    uint32_t num[16];
    for (uint32_t l = 0; l < 16; ++l) num[l] = 0;
    for (uint32_t i = 0; i < m; ++i) ++num[a[i] < max_len ? a[i] : max_len];
    uint32_t total = 0;
    for (uint32_t l = max_len; l >= 1; --l) total += num[l] << (max_len - l);
    while (total > (1u << max_len)) {
        --num[max_len];
        for (uint32_t l = max_len - 1; l >= 1; --l)
            if (num[l]) { --num[l]; num[l + 1] += 2; break; }
        --total;
    }
    // the least frequent symbols get the longest codes
    uint32_t i = 0;
    for (uint32_t l = max_len; l >= 1; --l)
        for (uint32_t k = num[l]; k; --k) len[sym[i++]] = (uint8_t)l;
}

// canonical codes of RFC 1951 3.2.2, stored bit-reversed (the way the bit stream wants them)
SBX_HD inline void canonical_codes(const uint8_t* len, uint32_t n_sym, uint32_t max_len, uint16_t* code) {
    uint32_t count[16], next[16];
    for (uint32_t l = 0; l < 16; ++l) count[l] = 0;
    for (uint32_t s = 0; s < n_sym; ++s) ++count[len[s]];
    count[0] = 0;
    uint32_t c = 0;
    for (uint32_t l = 1; l <= max_len; ++l) { c = (c + count[l - 1]) << 1; next[l] = c; }
    for (uint32_t s = 0; s < n_sym; ++s) code[s] = len[s] ? (uint16_t)rev_bits(next[len[s]]++, len[s]) : 0;
}

// The code lengths of both alphabets, run-length coded with the symbols 16 / 17 / 18 of RFC 1951 3.2.7: out(symbol, extra value,
// extra bits) per token.  lens = the hlit literal/length lengths followed by the hdist distance lengths.
template <class Out>
SBX_HD inline void code_length_tokens(const uint8_t* len, uint32_t hlit, uint32_t hdist, Out& out) {
    const uint32_t total = hlit + hdist;
    uint32_t k = 0;
    while (k < total) {
        const uint32_t v = k < hlit ? len[k] : len[kDistBase + k - hlit];
        uint32_t run = 1;
        while (k + run < total && (k + run < hlit ? len[k + run] : len[kDistBase + k + run - hlit]) == v) ++run;
        k += run;
        if (v == 0) {
            while (run >= 11) { const uint32_t t = run < 138 ? run : 138; out(18u, t - 11, 7u); run -= t; }
            if (run >= 3) { out(17u, run - 3, 3u); run = 0; }
            for (; run; --run) out(0u, 0u, 0u);
        } else {
            out(v, 0u, 0u);
            --run;
            while (run >= 3) { const uint32_t t = run < 6 ? run : 6; out(16u, t - 3, 2u); run -= t; }
            for (; run; --run) out(v, 0u, 0u);
        }
    }
}
struct CountClTokens {
    uint16_t* freq;
    uint32_t extra;
    SBX_HD void operator()(uint32_t s, uint32_t, uint32_t eb) { ++freq[s]; extra += eb; }
};
struct WriteClTokens {
    BitSink* s;
    const uint16_t* code;
    const uint8_t* len;
    SBX_HD void operator()(uint32_t t, uint32_t ev, uint32_t eb) { s->put((uint32_t)code[t] | (ev << len[t]), len[t] + eb); }
};

SBX_HD inline uint32_t fixed_litlen_bits(uint32_t s) { return s < 144 ? 8u : s < 256 ? 9u : s < 280 ? 7u : 8u; }

// Raw deflate of in[0, n) (n <= 65535) into out[0, cap) with a dynamic Huffman code (or the fixed one when that is shorter): returns
// the number of bytes, or 0 when it does not fit.  table: 1 << kHashBits entries, all zero on entry.
template <int kWays, bool kLazy, uint32_t kInsert>
SBX_HD inline uint32_t deflate_dynamic(const uint8_t* in, uint32_t n, uint8_t* out, uint32_t cap, uint16_t* table, DynWork* w) {
    for (uint32_t s = 0; s < 320; ++s) w->freq[s] = 0;
    CountTokens count{w->freq};
    lz77_tokens<kWays, kLazy, kInsert>(in, n, table, count);
    for (uint32_t k = 0; k < (1u << kHashBits); ++k) table[k] = 0;
    w->freq[256] = 1;                                           // end of block
    // what the fixed code would cost, before symbols are added to complete the codes (the extra bits cost the same either way)
    uint32_t fixed_bits = 3;
    for (uint32_t s = 0; s < kNumLitLen; ++s) fixed_bits += (uint32_t)w->freq[s] * fixed_litlen_bits(s);
    for (uint32_t s = 0; s < kNumDist; ++s) fixed_bits += (uint32_t)w->freq[kDistBase + s] * 5u;
    at_least_two_symbols(w->freq, kNumLitLen);
    at_least_two_symbols(w->freq + kDistBase, kNumDist);
    huffman_lengths(w->freq, kNumLitLen, 15, w->len, w->a, w->sym);
    huffman_lengths(w->freq + kDistBase, kNumDist, 15, w->len + kDistBase, w->a, w->sym);
    uint32_t hlit = kNumLitLen, hdist = kNumDist;
    while (hlit > 257 && !w->len[hlit - 1]) --hlit;
    while (hdist > 1 && !w->len[kDistBase + hdist - 1]) --hdist;
    for (uint32_t s = 0; s < 20; ++s) w->cl_freq[s] = 0;
    CountClTokens clc{w->cl_freq, 0};
    code_length_tokens(w->len, hlit, hdist, clc);
    at_least_two_symbols(w->cl_freq, kNumCl);
    huffman_lengths(w->cl_freq, kNumCl, 7, w->cl_len, w->a, w->sym);
    const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    uint32_t hclen = 19;
    while (hclen > 4 && !w->cl_len[order[hclen - 1]]) --hclen;
    uint32_t dyn_bits = 3 + 5 + 5 + 4 + 3 * hclen + clc.extra;
    for (uint32_t s = 0; s < kNumCl; ++s) dyn_bits += (uint32_t)w->cl_freq[s] * w->cl_len[s];
    for (uint32_t s = 0; s < kNumLitLen; ++s) dyn_bits += (uint32_t)w->freq[s] * w->len[s];
    for (uint32_t s = 0; s < kNumDist; ++s) dyn_bits += (uint32_t)w->freq[kDistBase + s] * w->len[kDistBase + s];
    if (fixed_bits <= dyn_bits) return deflate_fixed(in, n, out, cap, table);
    canonical_codes(w->len, kNumLitLen, 15, w->code);
    canonical_codes(w->len + kDistBase, kNumDist, 15, w->code + kDistBase);
    canonical_codes(w->cl_len, kNumCl, 7, w->cl_code);
    BitSink s;
    s.init(out, cap);
    s.put(1, 1);                      // BFINAL
    s.put(2, 2);                      // BTYPE = 10: dynamic Huffman codes
    s.put(hlit - 257, 5);
    s.put(hdist - 1, 5);
    s.put(hclen - 4, 4);
    for (uint32_t k = 0; k < hclen; ++k) s.put(w->cl_len[order[k]], 3);
    WriteClTokens clw{&s, w->cl_code, w->cl_len};
    code_length_tokens(w->len, hlit, hdist, clw);
    WriteTokens write{&s, w->code, w->len};
    lz77_tokens<kWays, kLazy, kInsert>(in, n, table, write);
    s.put(w->code[256], w->len[256]);
    const uint32_t bytes = s.finish();
    return s.overflow ? 0u : bytes;
}

// one stored block (BTYPE 00) holding in[0, n), n <= 65535: 5 + n bytes
SBX_HD inline uint32_t deflate_stored(const uint8_t* in, uint32_t n, uint8_t* out, uint32_t cap) {
    if (5u + n > cap) return 0;
    out[0] = 1;            // BFINAL = 1, BTYPE = 00, padding
    out[1] = (uint8_t)n; out[2] = (uint8_t)(n >> 8);
    out[3] = (uint8_t)~n; out[4] = (uint8_t)((~n) >> 8);
    for (uint32_t k = 0; k < n; ++k) out[5 + k] = in[k];
    return 5u + n;
}

// CRC-32 (IEEE 802.3, the one gzip and BGZF use), table driven; crc_table: 256 entries from crc32_make_table
SBX_HD inline void crc32_make_entry(uint32_t* table, uint32_t k) {
    uint32_t c = k;
    for (int j = 0; j < 8; ++j) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
    table[k] = c;
}
SBX_HD inline uint32_t crc32_bytes(const uint32_t* table, const uint8_t* p, uint32_t n) {
    uint32_t c = 0xFFFFFFFFu;
    for (uint32_t k = 0; k < n; ++k) c = table[(c ^ p[k]) & 0xFFu] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}

// A whole BGZF block around in[0, n) (n <= 0xFF00) at out[0, kBgzfSlot): header with the BC subfield, deflate data, CRC32,
// ISIZE (SAM specification 4.1; bgzf/compress.d:60-103).  Returns the block length.  level: 0 stored, 1 .. 3 fixed code, 4 .. 9 and
// -1 (Z_DEFAULT_COMPRESSION, the reference's default: bgzfCompress(chunk, level = -1)) dynamic code.
SBX_HD inline uint32_t bgzf_block(const uint8_t* in, uint32_t n, int level, uint8_t* out, uint16_t* table, DynWork* work, const uint32_t* crc_table) {
    const uint8_t hdr[16] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 'B', 'C', 2, 0};
    for (int k = 0; k < 16; ++k) out[k] = hdr[k];
    uint32_t clen = 0;
    if (level != 0)
        clen = !level_is_dynamic(level) ? deflate_fixed(in, n, out + 18, kBgzfSlot - 18 - 8, table)
               : level >= kThoroughFromLevel ? deflate_dynamic<4, true, 8>(in, n, out + 18, kBgzfSlot - 18 - 8, table, work)
                                             : deflate_dynamic<1, false, 8>(in, n, out + 18, kBgzfSlot - 18 - 8, table, work);
    if (clen == 0 || clen > n + 5u) clen = deflate_stored(in, n, out + 18, kBgzfSlot - 18 - 8);   // incompressible: one stored block
    const uint32_t total = 18 + clen + 8;
    out[16] = (uint8_t)(total - 1); out[17] = (uint8_t)((total - 1) >> 8);
    const uint32_t crc = crc32_bytes(crc_table, in, n);
    uint8_t* t = out + 18 + clen;
    t[0] = (uint8_t)crc; t[1] = (uint8_t)(crc >> 8); t[2] = (uint8_t)(crc >> 16); t[3] = (uint8_t)(crc >> 24);
    t[4] = (uint8_t)n; t[5] = (uint8_t)(n >> 8); t[6] = (uint8_t)(n >> 16); t[7] = (uint8_t)(n >> 24);
    return total;
}

}  // namespace sbx
