// lz77_copy.hpp -- the copy primitives of K1b `lz77_resolve` (inflate.hip), `__host__ __device__` so that the same statements run on the
// CPU (tests/cpp/inflate2_host.cpp: every length and alignment against memcpy, every period against the byte loop, and inside the
// lock-step model of the resolve rounds against zlib).
//
// gfx950 takes 4-, 8- and 16-byte LDS and global accesses at any byte address (hipcc emits ds_read_b64 / ds_write_b64 / global_load_dwordx2
// for an 8-byte memcpy of unknown alignment), so a short copy needs no arithmetic per dword:
//   8 .. 16 bytes  two 8-byte words at offsets 0 and n - 8 (they overlap in the middle: the same bytes are written twice),
//   4 .. 7 bytes   two dwords at offsets 0 and n - 4,
//   1 .. 3 bytes   one dword is read (the over-read stays inside the padded buffers) and 1 .. 3 bytes of it are written.
// Rounds 1-5 copied four dwords at offsets min(4 k, n - 4): config 2's K1b took 21.5 ms with them and 18.5 ms with these
// (profiles/round6/call_d_k1b_wide_config2.jsonl).
#pragma once
#include <cstdint>
#include <cstring>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define SBX_LZ_HD __host__ __device__ __forceinline__
#else
#define SBX_LZ_HD inline
#endif

namespace sbx {
namespace lz {

struct W2 { uint32_t x, y; };
SBX_LZ_HD uint32_t ld32(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
SBX_LZ_HD void st32(uint8_t* p, uint32_t v) { __builtin_memcpy(p, &v, 4); }
SBX_LZ_HD W2 ld64(const uint8_t* p) { W2 v; __builtin_memcpy(&v, p, 8); return v; }
SBX_LZ_HD void st64(uint8_t* p, W2 v) { __builtin_memcpy(p, &v, 8); }

#if defined(__HIP_DEVICE_COMPILE__)
SBX_LZ_HD uint32_t alignbyte(uint32_t hi, uint32_t lo, uint32_t s) { return __builtin_amdgcn_alignbyte(hi, lo, s); }
SBX_LZ_HD uint32_t perm(uint32_t hi, uint32_t lo, uint32_t sel) { return __builtin_amdgcn_perm(hi, lo, sel); }
#else
// v_alignbyte_b32: the low dword of {hi, lo} >> 8 (s & 3);  v_perm_b32 with selectors 0 .. 7: byte i of the result is byte sel.byte[i] of {hi, lo}
SBX_LZ_HD uint32_t alignbyte(uint32_t hi, uint32_t lo, uint32_t s) { return (uint32_t)(((((uint64_t)hi) << 32) | lo) >> (8u * (s & 3u))); }
SBX_LZ_HD uint32_t perm(uint32_t hi, uint32_t lo, uint32_t sel) {
    const uint64_t v = (((uint64_t)hi) << 32) | lo;
    uint32_t r = 0;
    for (uint32_t i = 0; i < 4; ++i) r |= (uint32_t)((v >> (8u * ((sel >> (8u * i)) & 7u))) & 0xFFu) << (8u * i);
    return r;
}
#endif

// One short copy task (n <= 16) of a lane, loads and stores separated so that a caller can put the loads of several tasks in flight
// before the first store.
struct Short16 {
    uint32_t w[4];
    SBX_LZ_HD void load(const uint8_t* s, uint32_t n) {
        if (n >= 8) {
            const W2 a = ld64(s), b = ld64(s + n - 8);
            w[0] = a.x; w[1] = a.y; w[2] = b.x; w[3] = b.y;
        } else if (n >= 4) {
            w[0] = ld32(s); w[1] = ld32(s + n - 4);
        } else if (n) {
            w[0] = ld32(s);
        }
    }
    SBX_LZ_HD void store(uint8_t* d, uint32_t n) const {
        if (n >= 8) {
            W2 a, b;
            a.x = w[0]; a.y = w[1]; b.x = w[2]; b.y = w[3];
            st64(d, a); st64(d + n - 8, b);
        } else if (n >= 4) {
            st32(d, w[0]); st32(d + n - 4, w[1]);
        } else if (n) {
            if (n & 2u) { const uint16_t h = (uint16_t)w[0]; __builtin_memcpy(d, &h, 2); }
            if (n & 1u) d[n & 2u] = (uint8_t)(w[0] >> (8u * (n & 2u)));
        }
    }
};

// A short self-overlapping match (a run, a dinucleotide repeat ...): output byte k = period[k mod dist], len <= 16, dist <= 8 < ... the
// period sits in the 8 bytes {x1, x0} at the match's source; output dword j is one byte permute of them, the selectors of a period being
// selector(dist, j) = the four bytes ((4 j + i) mod dist), i = 0 .. 3 (K1b keeps them in a 128-byte table in LDS).  The words are
// handed over in Short16's store layout.
SBX_LZ_HD uint32_t period_selector(uint32_t dist, uint32_t j) {
    uint32_t v = 0;
    for (uint32_t i = 0; i < 4; ++i) v |= ((4u * j + i) % dist) << (8u * i);
    return v;
}
SBX_LZ_HD void periodic16(uint32_t x0, uint32_t x1, uint32_t len, uint32_t s0, uint32_t s1, uint32_t s2, uint32_t s3, Short16* ws) {
    const uint32_t y0 = perm(x1, x0, s0), y1 = perm(x1, x0, s1), y2 = perm(x1, x0, s2), y3 = perm(x1, x0, s3);
    ws->w[0] = y0;
    if (len >= 8u) {
        const uint32_t to = len - 8u, tj = to >> 2, tsh = to & 3u;          // (tj == 2 only with tsh == 0)
        const uint32_t ta = tj == 0u ? y0 : tj == 1u ? y1 : y2;
        const uint32_t tb = tj == 0u ? y1 : tj == 1u ? y2 : y3;
        const uint32_t tc = tj == 0u ? y2 : y3;
        ws->w[1] = y1;
        ws->w[2] = alignbyte(tb, ta, tsh);
        ws->w[3] = alignbyte(tc, tb, tsh);
    } else {
        ws->w[1] = alignbyte(y1, y0, len >= 4u ? len - 4u : 0u);
    }
}

}  // namespace lz
}  // namespace sbx
