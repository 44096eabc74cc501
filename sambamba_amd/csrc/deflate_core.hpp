// deflate_core.hpp -- one BGZF block's worth of raw DEFLATE (RFC 1951) with the FIXED Huffman code and a greedy
// hash-table LZ77 matcher, written once for the device (deflate.hip: one lane per BGZF block) and for the host (the
// CPU unit test compiles this header with g++ and inflates the result with zlib).
//
// Replaces bgzfCompress (BioD/bio/core/bgzf/compress.d:34-103: zlib deflateInit2(level, Z_DEFLATED, -15, 8) + deflate(Z_FINISH)
// + crc32 per <= 0xFF00-byte block).  Any valid deflate stream of the block is acceptable to every BGZF reader -- the
// reference inflates with zlib (block.d:158-185) --, so the format is pinned by RFC 1951 / the SAM specification, not by
// zlib's bit-exact output; this encoder makes one choice (fixed code, greedy matches of >= 4 bytes found through a
// 2^kHashBits-entry table of last positions, no lazy evaluation) so that the same input gives the same bytes on the device and
// on the host.  An incompressible block falls back to one stored block.
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
// ISIZE (SAM specification 4.1; bgzf/compress.d:60-103).  Returns the block length.  level 0: stored; every other level of zlib's
// range, -1 (Z_DEFAULT_COMPRESSION, the reference's default: bgzfCompress(chunk, level = -1)) included: the one compressing mode.
SBX_HD inline uint32_t bgzf_block(const uint8_t* in, uint32_t n, int level, uint8_t* out, uint16_t* table, const uint32_t* crc_table) {
    const uint8_t hdr[16] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 'B', 'C', 2, 0};
    for (int k = 0; k < 16; ++k) out[k] = hdr[k];
    uint32_t clen = level != 0 ? deflate_fixed(in, n, out + 18, kBgzfSlot - 18 - 8, table) : 0;
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
