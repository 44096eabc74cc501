// format_core.hpp -- the row emitter of K6 `format_base_rows` (format.hip), `__host__ __device__` so that the same statements run on the
// CPU (tests/cpp/format_host.cpp: every value of dec4, the digit counts at every power of ten, random rows against snprintf).
//
// A row of `sambamba depth base` (PerBasePrinter.writeColumn, sambamba/depth.d:534-555) is a contig name and nine decimal numbers.
// Round 2's emitter wrote it byte by byte (one ds_write_b8 per character, digits from a loop of `% 10`): 4.4 ms per 90 M rows.  Here
//   * four decimal digits come out of ONE dword of arithmetic (dec4: the two halves of x / 100, x % 100 are divided by ten side by side),
//     already in memory order, and a number is one to three such groups with the leading zeros shifted out;
//   * the row is assembled in a 64-bit register (RowSink) and leaves eight bytes at a time -- gfx950 takes 8-byte LDS and global stores
//     at any byte address (see lz77_copy.hpp), so a row of ~35 bytes is five stores, and every store writes only bytes of its own row:
//     rows of neighbouring lanes are packed back to back.
#pragma once
#include <cstdint>
#include <cstring>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define SBX_FMT_HD __host__ __device__ __forceinline__
#else
#define SBX_FMT_HD inline
#endif

namespace sbx {
namespace fmt {

// decimal digits of w (1 .. 10)
SBX_FMT_HD uint32_t n_digits32(uint32_t w) {
    return 1u + (w >= 10u) + (w >= 100u) + (w >= 1000u) + (w >= 10000u) + (w >= 100000u) + (w >= 1000000u) + (w >= 10000000u) +
           (w >= 100000000u) + (w >= 1000000000u);
}
// ... when w < 10000 is known (the counters of a position whose coverage is below 10,000: every WGS)
SBX_FMT_HD uint32_t n_digits4(uint32_t w) { return 1u + (w >= 10u) + (w >= 100u) + (w >= 1000u); }
SBX_FMT_HD uint32_t n_digits64(uint64_t v) {
    if (v >= 10000000000ull) return 10u + n_digits32((uint32_t)(v / 10000000000ull));      // (2^64 / 10^10 < 2^31)
    return v > 0xFFFFFFFFull ? 10u : n_digits32((uint32_t)v);
}

// x < 10000 as four ASCII digits in memory order (thousands in the low byte)
SBX_FMT_HD uint32_t dec4(uint32_t x) {
    const uint32_t q = (x * 5243u) >> 19;                   // x / 100 (exact below 43,699)
    const uint32_t p = q | ((x - q * 100u) << 16);          // {x / 100, x % 100}: two numbers below 100
    const uint32_t tens = ((p * 103u) >> 10) & 0x000F000Fu; // both divided by ten (n * 103 >> 10, exact below 1,029; the halves do not meet)
    const uint32_t ones = p - tens * 10u;
    return (tens | (ones << 8)) + 0x30303030u;
}

// Bytes appended in order, eight at a time to memory at `p` (any alignment).  put(x, k): the low k bytes of x (1 <= k <= 8; the bytes of
// x above them must be zero).  finish() writes what is left and returns the end.
struct RowSink {
    uint8_t* p;
    uint64_t acc;
    uint32_t fill;          // bytes waiting in acc (0 .. 7)
    SBX_FMT_HD void init(uint8_t* at) { p = at; acc = 0; fill = 0; }
    SBX_FMT_HD void put(uint64_t x, uint32_t k) {
        acc |= x << (8u * fill);
        if (fill + k >= 8u) {
            __builtin_memcpy(p, &acc, 8);
            p += 8;
            acc = fill ? x >> (8u * (8u - fill)) : 0ull;
        }
        fill = (fill + k) & 7u;
    }
    SBX_FMT_HD uint8_t* finish() {
        uint32_t lo = (uint32_t)acc, hi = (uint32_t)(acc >> 32);
        if (fill & 4u) { __builtin_memcpy(p, &lo, 4); p += 4; lo = hi; }
        if (fill & 2u) { const uint16_t h = (uint16_t)lo; __builtin_memcpy(p, &h, 2); p += 2; lo >>= 16; }
        if (fill & 1u) { *p = (uint8_t)lo; p += 1; }
        fill = 0; acc = 0;
        return p;
    }
    // `lead` (one byte: the separator in front of the number) and the decimal digits of v
    SBX_FMT_HD void sep_num32(uint32_t lead, uint32_t v, uint32_t nd) {
        if (v < 10000u) {                   // nd <= 4: separator + digits are at most five bytes
            const uint64_t d = dec4(v) >> (8u * (4u - nd));
            put((uint64_t)lead | (d << 8), nd + 1u);
        } else {
            const uint32_t hi = v / 10000u, lo = v - hi * 10000u;
            if (hi < 10000u) {              // 5 .. 8 digits
                const uint64_t d = dec4(hi) >> (8u * (8u - nd));
                put((uint64_t)lead | (d << 8), nd - 3u);
            } else {                        // 9 or 10 digits
                const uint32_t top = hi / 10000u, mid = hi - top * 10000u;
                const uint64_t d = dec4(top) >> (8u * (12u - nd));
                put((uint64_t)lead | (d << 8), nd - 7u);
                put((uint64_t)dec4(mid), 4u);
            }
            put((uint64_t)dec4(lo), 4u);
        }
    }
    // x < 10^8 as eight digits
    SBX_FMT_HD void fixed8(uint32_t x) {
        const uint32_t hi = x / 10000u, lo = x - hi * 10000u;
        put((uint64_t)dec4(hi) | (uint64_t)dec4(lo) << 32, 8u);
    }
    SBX_FMT_HD void sep_num64(uint32_t lead, uint64_t v) {
        if (v <= 0xFFFFFFFFull) { sep_num32(lead, (uint32_t)v, n_digits32((uint32_t)v)); return; }
        // (a coverage beyond 2^32: never in practice)  v = top * 10^16 + mid * 10^8 + low
        const uint32_t nd = n_digits64(v);
        const uint64_t top = v / 10000000000000000ull, r = v - top * 10000000000000000ull;
        const uint32_t mid = (uint32_t)(r / 100000000ull), low = (uint32_t)(r - (uint64_t)mid * 100000000ull);
        if (top) { sep_num32(lead, (uint32_t)top, nd - 16u); fixed8(mid); }
        else sep_num32(lead, mid, nd - 8u);
        fixed8(low);
    }
    // a string of any length from memory (the contig and sample names)
    SBX_FMT_HD void str(const char* s, uint32_t len) {
        uint32_t i = 0;
        for (; i + 8u <= len; i += 8u) { uint64_t x; __builtin_memcpy(&x, s + i, 8); put(x, 8u); }
        if (i < len) {
            uint64_t x = 0;
            for (uint32_t k = 0; i + k < len; ++k) x |= (uint64_t)(uint8_t)s[i + k] << (8u * k);
            put(x, len - i);
        }
    }
};

}  // namespace fmt
}  // namespace sbx
