// inflate2_core.hpp -- the lane program of K1a `huffman_decode2` (round 4): one lane decodes one BGZF block's DEFLATE stream
// (RFC 1951; replaces the entropy-decoding half of decompressBgzfBlock, BioD/bio/core/bgzf/block.d:127-216) into the token
// streams K1b resolves.  The code is `__host__ __device__`: the kernel in inflate.hip instantiates it once per lane of a
// wavefront, tests/cpp/inflate2_host.cpp runs the very same statements lane by lane on the CPU against zlib
// (tests/test_inflate2_cpu.py) -- every decision of a lane depends on its own state only (`wave_any` merely keeps the
// wavefront's loops going), so the one-lane emulation is exact.
//
// What changed against round 3's K1a (k_huffman_decode, kept as the general kernel):
//  * The decoder never learns WHICH literal it decoded.  The parse of a DEFLATE stream depends only on whether a
//    literal/length symbol is a literal, the end-of-block code or a length code -- so the lane emits the literal's RANK among the
//    literals in canonical code order (8 bits) and leaves rank -> byte to a table of 256 bytes per deflate block that it
//    writes once while it sorts the code (k_translate_literals applies it to the literal stream afterwards).  The 324-byte
//    symbol permutation per lane that capped the kernel at 7 waves per CU is gone: per code length l the lane keeps
//    aux[l] = {litend_l, E_l} (first canonical index behind the literals of length l; number of non-literal symbols with
//    shorter codes) -- canonical index idx is a literal iff idx < litend_l, its rank is idx - E_l, and otherwise it is the
//    (idx - litend_l + E_l)-th non-literal symbol, looked up in a 30-byte list.  176 bytes of LDS per lane, laid out
//    lane-interleaved (dword j of lane i in bank i whatever j is): 14 waves per CU instead of 7.
//  * Bits come from a window fetched at the absolute bit position (two ring dwords + v_alignbit / v_lshrrev_b64) instead of a
//    64-bit buffer with a refill test in front of every symbol; literals are pushed into a 64-bit shift register (two
//    v_alignbit per literal) and leave through a 16-byte staging area in LDS, match entries through a ring of two groups: no register
//    FIFOs.  ONE vector memory instruction per iteration: the input load and a store that serves both token streams alternate
//    (flush_one, have_input) -- what such an instruction costs this kernel is in DESIGN.md section 3.
//  * The symbol loop exists in a plain and a general form (decode_lit / decode_dist), picked per wavefront: no flag tests per symbol.
//  * Everything the table build indexes dynamically (counters, insert positions) lives in LDS, not in compare-select chains
//    over registers, so the build does not set the kernel's register budget (<= 128 VGPRs: 4 waves per SIMD).
//  * Anything unusual -- stored blocks, over-subscribed or incomplete literal/length codes, more than kMaxSeg deflate blocks in a
//    BGZF block, any error -- ends the lane with kNeedsGeneral and the block is decoded again by the general kernel, which
//    also produces the authoritative error status.  The fast kernel has to be exact on valid streams and has to NOTICE invalid
//    ones; it never has to explain them.
#pragma once
#include <cstdint>
#include <cstring>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define SBX_HD __host__ __device__ __forceinline__
#else
#define SBX_HD inline
#endif

namespace sbx {
namespace inf2 {

constexpr uint32_t kNeedsGeneral = 100;     // status of a block the fast kernel hands to the general one
constexpr int kMaxSeg = 6;                  // deflate blocks (= literal translation tables) per BGZF block
// global scratch per BGZF block
constexpr int kScratchLens = 0;             // 160 bytes: the code lengths being read, 4 bits each
constexpr int kScratchInfo = 160;           // u32 n_seg, u32 n_lit, u32 seg_start[kMaxSeg]
constexpr int kScratchTabs = 256;           // kMaxSeg x 256 bytes: literal rank -> byte
constexpr int kScratchBytes = kScratchTabs + kMaxSeg * 256;    // 1792

// LDS of one wavefront, lane-interleaved.  dword arrays: dword j of lane i at off + 256 j + 4 i; u16 arrays: element e at
// off + 128 e + 2 i; byte arrays: element e at off + 64 e + i.
constexpr int kRingDw = 9;                  // 8 dwords of input + a copy of dword 0 behind them (a window is read as dwords t, t + 1)
constexpr int kOffRing = 0;
constexpr int kOffLitStage = kOffRing + 256 * kRingDw;     // 4 dwords: the 16-byte group of literal ranks being filled
constexpr int kOffEntStage = kOffLitStage + 256 * 4;       // 8 dwords: a ring of match entries, two 16-byte groups (a finished group may wait for its store
                                                           // while the next one fills: the symbol loop stores every second iteration only)
constexpr int kOffAux = kOffEntStage + 256 * 8;            // u16[16]: per code length {litend : 9, E : 6} (counters while a code is built)
constexpr int kSymEntries = 30;                            // entries of the two symbol lists (the fixed code's 31st and 32nd non-literal symbols, 286 and
                                                           // 287, are invalid: an index beyond the list is an error, nothing is kept for it)
constexpr int kOffLenSym = kOffAux + 128 * 16;             // u8[30]: non-literal symbols - 256 in canonical order (dist counters during the build)
constexpr int kOffDistSym = kOffLenSym + 64 * kSymEntries; // u8[30]: distance symbols in canonical order
constexpr int kWaveLds = kOffDistSym + 64 * kSymEntries;   // 11264 bytes = 44 dwords per lane: 14 wavefronts per CU
constexpr int kLenTabBytes = 64, kDistTabBytes = 128;      // RFC 1951 3.2.5 tables shared by the workgroup, behind the waves' areas

constexpr uint32_t kNone = 0xFFu, kStop = 0x1FFu;

#if defined(__HIP_DEVICE_COMPILE__)
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
SBX_HD uint32_t alignbit(uint32_t hi, uint32_t lo, uint32_t s) { return __builtin_amdgcn_alignbit(hi, lo, s); }
SBX_HD uint32_t brev32(uint32_t x) { return __builtin_bitreverse32(x); }
SBX_HD bool wave_any(bool x) { return __any(x) != 0; }
SBX_HD uint32_t bfe(uint32_t x, uint32_t off, uint32_t width) { return __builtin_amdgcn_ubfe(x, off, width); }
// acc += [v >= limit_a] * dd_a + [v >= limit_b] * dd_b for the two code lengths packed in (lim1, dd); vv = {v, v}
SBX_HD uint32_t pair_mask(uint32_t lim1, uint32_t vv) {
    const s16x2 d = __builtin_bit_cast(s16x2, lim1) - __builtin_bit_cast(s16x2, vv);
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2, d) >> 15);
}
SBX_HD uint32_t pair_dot(uint32_t m, uint32_t dd, uint32_t acc) {
    return __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, m), __builtin_bit_cast(u16x2, dd), acc, false);
}
#else
SBX_HD uint32_t alignbit(uint32_t hi, uint32_t lo, uint32_t s) { return (uint32_t)(((((uint64_t)hi) << 32) | lo) >> (s & 31u)); }
SBX_HD uint32_t brev32(uint32_t x) {
    x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
    x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
    x = ((x >> 4) & 0x0F0F0F0Fu) | ((x & 0x0F0F0F0Fu) << 4);
    x = ((x >> 8) & 0x00FF00FFu) | ((x & 0x00FF00FFu) << 8);
    return (x >> 16) | (x << 16);
}
SBX_HD bool wave_any(bool x) { return x; }
SBX_HD uint32_t bfe(uint32_t x, uint32_t off, uint32_t width) { return width == 0 ? 0u : (x >> (off & 31u)) & (width >= 32 ? 0xFFFFFFFFu : ((1u << width) - 1u)); }
SBX_HD uint32_t pair_mask(uint32_t lim1, uint32_t vv) {
    const uint32_t a = (uint16_t)((uint16_t)lim1 - (uint16_t)vv), b = (uint16_t)((uint16_t)(lim1 >> 16) - (uint16_t)(vv >> 16));
    return (a >> 15) | ((b >> 15) << 16);
}
SBX_HD uint32_t pair_dot(uint32_t m, uint32_t dd, uint32_t acc) { return acc + (m & 0xFFFFu) * (dd & 0xFFFFu) + (m >> 16) * (dd >> 16); }
#endif

SBX_HD uint32_t make_entry2(uint32_t lit_run, uint32_t len, uint32_t dist) { return (lit_run << 24) | ((dist - 1u) << 9) | len; }

struct u32x4h { uint32_t x, y, z, w; };


// Canonical code in registers, two 16-bit halves per register (as in round 3):
//   lim1[j] = { limit[2j+1] - 1, limit[2j+2] - 1 },  limit[l] = left-justified (15-bit) exclusive upper bound of the codes of length <= l
//   dd[j]   = { D[2j+2] - D[2j+1], D[2j+3] - D[2j+2] } (mod 2^9) | 1 << 13,  D[l] = first canonical index of length l - first code of length l
// For the 15-bit prefix v: acc = D[1] + sum_l [v >= limit_l] * dd_l;  code length = 1 + (acc >> 13), index = (acc + code) mod 2^9.
struct Code {
    uint32_t lim1[8], dd[8];
    uint32_t d1;        // D[1] mod 2^9
    uint32_t d1_lo;     // d1 + both halves of dd[0]: the accumulator behind pair 0 when the code has no codes of 1 or 2 bits
};

#if defined(__HIP_DEVICE_COMPILE__)
SBX_HD uint32_t splat16(uint32_t v) { const u16x2 x = {(unsigned short)v, (unsigned short)v}; return __builtin_bit_cast(uint32_t, x); }   // (folds into op_sel)
#define SBX_KEEP_BRANCH() asm volatile("")
#else
SBX_HD uint32_t splat16(uint32_t v) { return v | (v << 16); }
#define SBX_KEEP_BRANCH() (void)0
#endif

template <int kFrom, int kTo>
SBX_HD uint32_t decode_pairs(const Code& C, uint32_t v, uint32_t acc) {
    const uint32_t vv = splat16(v);
    uint32_t m[kTo - kFrom > 0 ? kTo - kFrom : 1];
#pragma unroll
    for (int j = kFrom; j < kTo; ++j) m[j - kFrom] = pair_mask(C.lim1[j], vv);
#pragma unroll
    for (int j = kFrom; j < kTo; ++j) acc = pair_dot(m[j - kFrom], C.dd[j], acc);
    return acc;
}

// kPlain: every lane's codes of the current deflate blocks are "plain" -- no codes of 1 or 2 bits (pair 0 of the sum is a constant,
// folded into d1_lo), the literal/length code complete within 14 bits and the distance code within 12 (the pairs above add nothing) --
// which is what zlib emits for a BAM (tools/token_stats.cpp: the literal/length code reaches 14 bits in 97 % of the blocks and never
// 15, the distance code 12 at most).  The symbol loop exists in both forms and a wavefront picks one per round of deflate blocks:
// no flag is tested inside the loop (round 3 tested scalar flags per symbol: two instructions to skip three).
template <bool kPlain>
SBX_HD uint32_t decode_lit(const Code& C, uint32_t v) { return kPlain ? decode_pairs<1, 7>(C, v, C.d1_lo) : decode_pairs<0, 8>(C, v, C.d1); }
template <bool kPlain>
SBX_HD uint32_t decode_dist(const Code& C, uint32_t v) { return kPlain ? decode_pairs<1, 6>(C, v, C.d1_lo) : decode_pairs<0, 8>(C, v, C.d1); }

// order of the code-length code lengths (RFC 1951 3.2.7), 5 bits each: entry i at bit 5 i
constexpr uint64_t kClOrderLo = 16ull | 17ull << 5 | 18ull << 10 | 0ull << 15 | 8ull << 20 | 7ull << 25 | 9ull << 30 | 6ull << 35 | 10ull << 40 | 5ull << 45 | 11ull << 50 | 4ull << 55;
constexpr uint64_t kClOrderHi = 12ull | 3ull << 5 | 13ull << 10 | 2ull << 15 | 14ull << 20 | 1ull << 25 | 15ull << 30;

struct LaneIo {             // what a lane is given
    const uint8_t* in;      // first byte of the raw DEFLATE payload
    uint32_t in_bits;       // its length in bits
    uint32_t osize;         // ISIZE of the BGZF block
    uint8_t* lit;           // 64-byte aligned literal stream slice of the block
    uint32_t* ent;          // 64-byte aligned entry stream slice
    uint8_t* scratch;       // kScratchBytes of global scratch of the block (16-byte aligned)
    bool live;              // false: a lane past the last block (takes part in the wave's loops, does nothing)
};
struct LaneResult {
    uint32_t status;        // 0 ok, kNeedsGeneral
    uint32_t n_ent, n_lit;
};

// The lane.  W = LDS area of the wavefront, lane = lane index (0..63), len_tab / dist_tab = the workgroup's RFC 1951 tables.
struct Lane {
    uint8_t* W;
    uint32_t l4, l2, l1;            // byte offsets of this lane inside dword / u16 / byte arrays
    const uint16_t* len_tab;
    const uint32_t* dist_tab;
    // input
    const uint8_t* gp;              // next 16-byte chunk to fetch
    u32x4h pend;                    // chunk on its way to the ring
    uint32_t wr_dw;                 // dwords put into the ring so far (multiple of 4): the ring holds dwords [wr_dw - 8, wr_dw)
    uint32_t bitpos;                // bits consumed, counted from the 4-byte aligned address below the payload
    // output
    uint32_t hi, lo;                // the last 8 literal ranks, newest in hi[31:24]
    uint32_t n_lit, n_ent, lit_mark, opos;
    uint32_t lit_flushed, ent_flushed;      // 16-byte groups written to the streams
    uint8_t* lit;
    uint32_t* ent;

    SBX_HD uint32_t& dw(int off, uint32_t j) const { return *(uint32_t*)(W + off + 256u * j + l4); }
    SBX_HD uint16_t& h16(int off, uint32_t e) const { return *(uint16_t*)(W + off + 128u * e + l2); }
    SBX_HD uint8_t& b8(int off, uint32_t e) const { return W[off + 64u * e + l1]; }

    SBX_HD static u32x4h load16(const uint8_t* p) {
        u32x4h v;
#if defined(__HIP_DEVICE_COMPILE__)
        // (a global_load: as a generic pointer carried around the loops the address becomes a flat_load, which also counts in lgkmcnt)
        typedef uint32_t u32x4g __attribute__((ext_vector_type(4)));
        const u32x4g q = *(const __attribute__((address_space(1))) u32x4g*)(uintptr_t)p;
        v.x = q.x; v.y = q.y; v.z = q.z; v.w = q.w;
#else
        __builtin_memcpy(&v, p, 16);
#endif
        return v;
    }
    SBX_HD static void store16(void* p, u32x4h v) { __builtin_memcpy(p, &v, 16); }

    SBX_HD void put_chunk(u32x4h v) {
        const uint32_t s = wr_dw & 7u;      // 0 or 4
        dw(kOffRing, s) = v.x; dw(kOffRing, s + 1) = v.y; dw(kOffRing, s + 2) = v.z; dw(kOffRing, s + 3) = v.w;
        if (s == 0u) dw(kOffRing, 8) = v.x;
        wr_dw += 4;
    }
    SBX_HD void init_input(const uint8_t* p) {
        const uint32_t lead = (uint32_t)((uintptr_t)p & 3u);
        const uint8_t* a = p - lead;
        wr_dw = 0;
        put_chunk(load16(a));
        put_chunk(load16(a + 16));
        pend = load16(a + 32);
        gp = a + 48;
        bitpos = 8u * lead;
    }
    // Once per step / loop iteration.  A step consumes at most 63 bits; behind a service the ring holds at least the five dwords from
    // the current one on, a window reads two of the four a step can reach.
    // The prefetch is NOT bounded by the block's payload: the end of the input is tested between deflate blocks (run(): in_bits), so on a
    // corrupt stream a lane may read on until its output position passes ISIZE -- every symbol it consumes produces at least one of at
    // most 65536 output bytes: < 128 KiB beyond its own block.  The caller keeps that much readable behind the compressed bytes
    // (engine.cpp kCompPad); the lane ends with kNeedsGeneral and the general kernel, which tests its input per symbol, names the error.
    SBX_HD void service() {
        if ((bitpos >> 5) + 4u >= wr_dw) {
            put_chunk(pend);
            pend = load16(gp);
            gp += 16;
        }
    }
    SBX_HD uint32_t window32(uint32_t at) const {
        const uint32_t t = (at >> 5) & 7u;
        return alignbit(dw(kOffRing, t + 1), dw(kOffRing, t), at);
    }
    SBX_HD uint64_t window64(uint32_t at) const {      // >= 33 valid bits
        const uint32_t t = (at >> 5) & 7u;
        return ((((uint64_t)dw(kOffRing, t + 1)) << 32) | dw(kOffRing, t)) >> (at & 31u);
    }

    // ---- output ---------------------------------------------------------------------------------------------------------
    SBX_HD void push_lit(uint32_t rank) {
        lo = alignbit(hi, lo, 8);
        hi = alignbit(rank, hi, 8);
        ++n_lit;
    }
    // after the literal slots of an iteration (at most two literals): the dword completed since n_lit0, if any, goes to the staging group
    SBX_HD void stage_lit_dword(uint32_t n_lit0) {
        if ((n_lit >> 2) != (n_lit0 >> 2)) {
            const uint32_t d = (n_lit & 3u) ? alignbit(hi, lo, 24) : hi;    // one literal may already sit above the finished dword
            dw(kOffLitStage, ((n_lit >> 2) - 1u) & 3u) = d;
        }
    }
    SBX_HD void stage_entry(uint32_t e) {
        dw(kOffEntStage, n_ent & 7u) = e;
        ++n_ent;
    }
    SBX_HD u32x4h lit_group() const {
        u32x4h g;
        g.x = dw(kOffLitStage, 0); g.y = dw(kOffLitStage, 1); g.z = dw(kOffLitStage, 2); g.w = dw(kOffLitStage, 3);
        return g;
    }
    SBX_HD u32x4h ent_group(uint32_t grp) const {           // group `grp` of the entry stream sits in half grp & 1 of the ring
        const uint32_t b = 4u * (grp & 1u);
        u32x4h g;
        g.x = dw(kOffEntStage, b); g.y = dw(kOffEntStage, b + 1u); g.z = dw(kOffEntStage, b + 2u); g.w = dw(kOffEntStage, b + 3u);
        return g;
    }
    // every finished group goes to its stream (outside the symbol loop)
    SBX_HD void flush_groups() {
        if ((n_lit >> 4) != lit_flushed) {
            store16(lit + 16u * lit_flushed, lit_group());
            ++lit_flushed;
        }
        while ((n_ent >> 2) != ent_flushed) {
            store16(ent + 4u * ent_flushed, ent_group(ent_flushed));
            ++ent_flushed;
        }
    }
    // The symbol loop: ONE store instruction every second iteration for both streams.  A vector memory instruction costs a
    // wavefront of this kernel ~1.2 us of its CU's memory pipeline whatever its execution mask says (profiles/round4: without its
    // two stores per iteration the kernel took 17.9 instead of 24.9 ms, with the two merged into one 20.1), so the loop issues its
    // input load in one iteration and this store in the next.  A lane writes its finished literal group if it has one, else its oldest
    // finished entry group.  Nothing is overwritten while it waits: the literal staging area is written again when four more literals
    // have been decoded (two iterations), the entry ring holds a finished group and four more entries -- literal groups take at
    // most every fourth store, entry groups are finished at most every fourth iteration.
    SBX_HD void flush_one() {
        const bool rl = (n_lit >> 4) != lit_flushed, re = (n_ent >> 2) != ent_flushed;
        if (rl || re) {
            const uint32_t b = rl ? 0u : 4u * (ent_flushed & 1u);
            const int off = rl ? kOffLitStage : kOffEntStage;
            u32x4h g;
            g.x = dw(off, b); g.y = dw(off, b + 1u); g.z = dw(off, b + 2u); g.w = dw(off, b + 3u);
            uint8_t* const p = rl ? lit + 16u * lit_flushed : (uint8_t*)(ent + 4u * ent_flushed);
            store16(p, g);
            lit_flushed += rl ? 1u : 0u;
            ent_flushed += rl ? 0u : 1u;
        }
    }
    // enough input in the ring for one iteration's windows (two literal/length symbols, then a 64-bit window)?  With the input
    // served every second iteration only a lane can, in principle, run short: it then sits an iteration out.
    SBX_HD bool have_input() const { return (wr_dw << 5) - bitpos >= 96u; }

    template <bool kPlain>
    SBX_HD void symbol_loop(const Code& CL, const Code& CD, const LaneIo& io, bool huff, uint32_t& err, bool& active) {
    // ---- symbol loop --------------------------------------------------------------------------------------------
    // An iteration: up to two literal/length symbols per lane (literals are pushed on the spot, the first other symbol stops the
    // lane's run and stays pending), then for the pending one at most one entry: the match (length extra bits, distance code,
    // distance extra bits), or -- when 255 literals have piled up -- a literal-run entry, the match waiting one more iteration.
    // st: kNone = decoding, nothing pending; < 32 = a non-literal symbol is pending (its index in the canonical list); kStop = the lane
    // is not (any longer) in this deflate block
    uint32_t st = huff ? kNone : kStop;
    // (two iterations per turn of the loop: the input is served in the first, the token store issued in the second)
    if (wave_any(st != kStop)) do {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
        if (half == 0) service(); else flush_one();
        const bool input = have_input();
        const uint32_t n_lit0 = n_lit;
        if (st == kNone && input) {
            // both symbols are decoded back to back -- the second one speculatively: it counts only if the first was a literal -- so that
            // their table reads are in flight together and nothing below is control flow
            const uint32_t w = window32(bitpos);
            const uint32_t v1 = brev32(w) >> 17;
            const uint32_t acc1 = decode_lit<kPlain>(CL, v1);
            const uint32_t len1 = (acc1 >> 13) + 1u;                        // complete code: <= 15
            const uint32_t a1 = h16(kOffAux, len1);
            const uint32_t v2 = brev32(w >> len1) >> 17;
            const uint32_t acc2 = decode_lit<kPlain>(CL, v2);
            const uint32_t len2 = (acc2 >> 13) + 1u;
            const uint32_t a2 = h16(kOffAux, len2);
            const uint32_t idx1 = ((v1 >> (15u - len1)) + acc1) & 0x1FFu, idx2 = ((v2 >> (15u - len2)) + acc2) & 0x1FFu;
            const uint32_t litend1 = a1 & 0x1FFu, E1 = a1 >> 9, litend2 = a2 & 0x1FFu, E2 = a2 >> 9;
            const bool lit1 = idx1 < litend1, lit2 = lit1 && idx2 < litend2;
            {
                const uint32_t nlo = alignbit(hi, lo, 8), nhi = alignbit(idx1 - E1, hi, 8);
                lo = lit1 ? nlo : lo;
                hi = lit1 ? nhi : hi;
            }
            {
                const uint32_t nlo = alignbit(hi, lo, 8), nhi = alignbit(idx2 - E2, hi, 8);
                lo = lit2 ? nlo : lo;
                hi = lit2 ? nhi : hi;
            }
            n_lit += (lit1 ? 1u : 0u) + (lit2 ? 1u : 0u);
            bitpos += len1 + (lit1 ? len2 : 0u);
            const uint32_t j1 = (idx1 - litend1 + E1) & 31u, j2 = (idx2 - litend2 + E2) & 31u;
            st = !lit1 ? j1 : !lit2 ? j2 : kNone;
        }
        stage_lit_dword(n_lit0);
        const uint32_t opos_now = opos + n_lit;
        const uint32_t run_len = n_lit - lit_mark;
        const bool split = run_len >= 255u;              // (also for a lane that waits for the others: they are 255 literals all the same)
        const bool do_d = !split && st < 32u && input;
        bool eob = false, m_ok = false;
        uint32_t m_e = 0, m_bits = 0, m_len = 0;
        if (do_d) {
            const uint32_t s5 = b8(kOffLenSym, st < (uint32_t)kSymEntries ? st : 0u);   // symbol - 256: 0 end of block, 1..29 length codes
            const uint32_t lt = len_tab[(s5 - 1u) & 31u];
            uint64_t w64 = window64(bitpos);
            const uint32_t le = lt >> 9, lb = lt & 0x1FFu;
            const uint32_t mlen = lb + bfe((uint32_t)w64, 0, le);
            w64 >>= le;
            const uint32_t dv = brev32((uint32_t)w64) >> 17;
            const uint32_t acc = decode_dist<kPlain>(CD, dv);
            const uint32_t dm1 = acc >> 13;                           // 15: no such code
            const uint32_t dl = dm1 + 1u;
            const uint32_t didx = ((dv >> ((14u - (dm1 & 15u)) & 31u)) + acc) & 0x1FFu;
            const uint32_t dsym = b8(kOffDistSym, didx < (uint32_t)kSymEntries ? didx : 0u);
            const uint32_t dt = dist_tab[dsym & 31u];
            w64 >>= dl;
            const uint32_t de = dt >> 16, db = dt & 0xFFFFu;
            const uint32_t dist = db + bfe((uint32_t)w64, 0, de);
            eob = s5 == 0u && st < (uint32_t)kSymEntries;
            // (a distance code that exists has an index below the number of codes, its symbol is below 30 by construction)
            m_ok = st < (uint32_t)kSymEntries && s5 - 1u < 29u && dm1 <= 14u && didx < (uint32_t)kSymEntries && dist <= opos_now && opos_now + mlen <= io.osize;
            m_e = make_entry2(run_len, mlen, dist);
            m_bits = le + dl + de;
            m_len = mlen;
        }
        const bool bad = (do_d && !eob && !m_ok) || (st != kStop && opos_now > io.osize);
        if (split || m_ok) stage_entry(split ? make_entry2(255, 0, 1) : m_e);
        lit_mark = split ? lit_mark + 255u : m_ok ? n_lit : lit_mark;
        bitpos += m_ok ? m_bits : 0u;
        opos += m_ok ? m_len : 0u;
        st = do_d ? (eob ? kStop : kNone) : st;
        if (bad) { err = 1; st = kStop; active = false; }
        }
    } while (wave_any(st != kStop));
    }

    // ---- one block ------------------------------------------------------------------------------------------------------
    SBX_HD LaneResult run(const LaneIo& io, uint8_t* W_, uint32_t lane, const uint16_t* len_tab_, const uint32_t* dist_tab_) {
        W = W_;
        l4 = 4u * lane; l2 = 2u * lane; l1 = lane;
        len_tab = len_tab_;
        dist_tab = dist_tab_;
        lit = io.lit;
        ent = io.ent;
        hi = lo = 0;
        n_lit = n_ent = lit_mark = opos = lit_flushed = ent_flushed = 0;
        init_input(io.in);
        const uint32_t lead_bits = bitpos;
        uint32_t* const lens32 = (uint32_t*)(io.scratch + kScratchLens);
        uint32_t* const info = (uint32_t*)(io.scratch + kScratchInfo);
        uint32_t n_seg = 0;
        uint32_t err = 0;
        Code CL, CD;
#pragma unroll
        for (int j = 0; j < 8; ++j) { CL.lim1[j] = CL.dd[j] = CD.lim1[j] = CD.dd[j] = 0; }
        CL.d1 = CL.d1_lo = CD.d1 = CD.d1_lo = 0;

        bool active = io.live && !(io.osize == 0 && io.in_bits == 0);      // nothing to do for an empty payload
        bool last = false;
        while (wave_any(active)) {
            // ---- block header -------------------------------------------------------------------------------------------
            service();
            uint32_t btype = 3;
            if (active) {
                const uint32_t w = window32(bitpos);
                last = (w & 1u) != 0;
                btype = (w >> 1) & 3u;
                bitpos += 3;
                if (btype == 0u || btype == 3u) { err = 1; active = false; }        // stored blocks: the general kernel's
                if (bitpos - lead_bits > io.in_bits) { err = 1; active = false; }
                if (n_seg >= (uint32_t)kMaxSeg) { err = 1; active = false; }
            }
            uint32_t nlit = 0, ndist = 0;
            bool has_eob = false;
            // counters of the build: literal/length code in the aux area {literals : 9 | others : 6}, distance code in the len-sym area
            if (active) {
#pragma unroll
                for (uint32_t l = 0; l < 16; ++l) { h16(kOffAux, l) = 0; b8(kOffLenSym, l) = 0; }
            }
            if (active && btype == 1u) {
                // fixed code (RFC 1951 3.2.6): 0-143 8 bits, 144-255 9, 256-279 7, 280-287 8; 30 distance codes of 5 bits
                for (uint32_t i = 0; i < 40; ++i)
                    lens32[i] = i < 18 ? 0x88888888u : i < 32 ? 0x99999999u : i < 35 ? 0x77777777u : i < 36 ? 0x88888888u : i < 39 ? 0x55555555u : 0x00555555u;
                nlit = 288; ndist = 30;
                h16(kOffAux, 7) = (uint16_t)(24u << 9);
                h16(kOffAux, 8) = (uint16_t)(144u | 8u << 9);
                h16(kOffAux, 9) = (uint16_t)112u;
                b8(kOffLenSym, 5) = 30;
                has_eob = true;
            }
            // ---- dynamic code: the code-length code, then nlit + ndist lengths ---------------------------------------------
            bool dyn = active && btype == 2u;
            uint64_t clen = 0;          // 3 bits per symbol of the code-length code
            int ncl_left = 0;
            if (dyn) {
                const uint32_t w = window32(bitpos);
                nlit = (w & 31u) + 257u;
                ndist = ((w >> 5) & 31u) + 1u;
                ncl_left = (int)((w >> 10) & 15u) + 4;
                bitpos += 14;
                if (nlit > 286u || ndist > 30u) { err = 1; active = false; dyn = false; ncl_left = 0; }
            }
            for (uint32_t i = 0; wave_any(ncl_left != 0); ++i) {          // i is the same in all lanes
                service();
                if (ncl_left != 0) {
                    const uint32_t ord = (uint32_t)((i < 12u ? kClOrderLo >> (5u * i) : kClOrderHi >> (5u * (i - 12u))) & 31u);
                    const uint32_t w = window32(bitpos);
                    clen |= (uint64_t)(w & 7u) << (3u * ord);
                    bitpos += 3;
                    --ncl_left;
                }
            }
            uint64_t ccnt = 0;          // 5 bits per length 0..7: codes of that length
            uint64_t csa = 0, csb = 0;  // symbols sorted by (length, value), 5 bits each: 12 in csa, 7 in csb
            if (dyn) {
                for (uint32_t s = 0; s < 19; ++s) ccnt += 1ull << (5u * (uint32_t)((clen >> (3u * s)) & 7u));
                uint32_t k = 0;
                for (uint32_t l = 1; l <= 7; ++l)
                    for (uint32_t s = 0; s < 19; ++s)
                        if (((clen >> (3u * s)) & 7u) == l) {
                            if (k < 12u) csa |= (uint64_t)s << (5u * k); else csb |= (uint64_t)s << (5u * (k - 12u));
                            ++k;
                        }
                int32_t left = 1;
                bool ok = true;
                for (uint32_t l = 1; l <= 7; ++l) { left = (left << 1) - (int32_t)((ccnt >> (5u * l)) & 31u); if (left < 0) ok = false; }
                if (!ok) { err = 1; active = false; dyn = false; }
            }
            {
                uint32_t i = 0, prev = 0, nib = 0;          // nib: the dword of eight lengths being filled
                const uint32_t total = nlit + ndist;
                bool more = dyn;
                while (wave_any(more)) {
                    service();
                    if (more) {
                        const uint32_t w = window32(bitpos);
                        // canonical walk over lengths 1..7
                        uint32_t code = 0, first = 0, index = 0, sym = 99, used = 0;
#pragma unroll
                        for (uint32_t l = 1; l <= 7; ++l) {
                            code |= (w >> (l - 1)) & 1u;
                            const uint32_t c = (uint32_t)(ccnt >> (5u * l)) & 31u;
                            if (used == 0u && code < first + c) {
                                const uint32_t q = index + (code - first);
                                sym = (uint32_t)((q < 12u ? csa >> (5u * q) : csb >> (5u * (q - 12u))) & 31u);
                                used = l;
                            }
                            index += c;
                            first = (first + c) << 1;
                            code <<= 1;
                        }
                        uint32_t rep = 0, val = 0;
                        if (used == 0u || sym >= 19u) { err = 1; more = false; }
                        else {
                            const uint32_t x = w >> used;
                            if (sym < 16u) { rep = 1; val = sym; prev = sym; bitpos += used; }
                            else if (sym == 16u) { rep = 3u + (x & 3u); val = prev; bitpos += used + 2u; if (i == 0u) { err = 1; more = false; rep = 0; } }
                            else if (sym == 17u) { rep = 3u + (x & 7u); val = 0; prev = 0; bitpos += used + 3u; }
                            else { rep = 11u + (x & 127u); val = 0; prev = 0; bitpos += used + 7u; }
                            if (i + rep > total) { err = 1; more = false; rep = 0; }
                        }
                        // the lengths go to the scratch 4 bits each and are counted per code length as they come
                        while (rep != 0u) {
                            if (val != 0u) {
                                if (i < 256u) h16(kOffAux, val) += 1u;
                                else if (i < nlit) { h16(kOffAux, val) += 512u; if (i == 256u) has_eob = true; }
                                else b8(kOffLenSym, val) += 1u;
                            }
                            nib |= val << (4u * (i & 7u));
                            ++i;
                            if ((i & 7u) == 0u) { lens32[(i >> 3) - 1u] = nib; nib = 0; }
                            --rep;
                        }
                        if (more && i >= total) {
                            more = false;
                            if (i & 7u) lens32[i >> 3] = nib;
                        }
                    }
                }
                if (dyn && err == 0u && !has_eob) err = 1;         // no end-of-block code
                if (err != 0u) { active = false; dyn = false; }
            }
            bool huff = active;        // (btype 1 or 2 with a readable header)
            // ---- distance code: limits / deltas, insert positions, symbols in canonical order -------------------------------
            bool dist_in12 = true, dist_no12 = true, lit_in14 = true, lit_no12 = true;
            if (huff) {
                uint32_t first = 0, offs = 0, lim_prev = 0, D_prev = 0, lim_a = 0, dd_a = 0;
                int32_t left = 1;
                bool ok = true;
#pragma unroll
                for (uint32_t l = 1; l <= 16; ++l) {
                    uint32_t lim, D;
                    if (l <= 15) {
                        const uint32_t c = b8(kOffLenSym, l);
                        left = (left << 1) - (int32_t)c;
                        if (left < 0) ok = false;
                        lim = (first + c) << (15u - l);
                        D = offs - first;
                        b8(kOffLenSym, l) = (uint8_t)offs;         // insert position of the sort below
                        offs += c;
                        first = (first + c) << 1;
                        if (l == 2) dist_no12 = offs == 0u;
                        if (l == 12) dist_in12 = lim == 32768u;
                    } else { lim = 32768u; D = D_prev; }
                    // lim1 half = lim - 1; dd half = (D[l+1] - D[l]) & 0x1FF | 0x2000 needs D of the NEXT length: emitted one step late
                    if (l == 1) CD.d1 = D & 0x1FFu;
                    else {
                        const uint32_t ddh = ((D - D_prev) & 0x1FFu) | 0x2000u;      // delta between length l-1 and l
                        // ddh belongs to pair slot of length l-1
                        const uint32_t lm = l - 1;       // 1..15
                        if (lm & 1u) { lim_a = (lim_prev - 1u) & 0xFFFFu; dd_a = ddh; }
                        else { CD.lim1[(lm >> 1) - 1u] = lim_a | ((lim_prev - 1u) & 0xFFFFu) << 16; CD.dd[(lm >> 1) - 1u] = dd_a | ddh << 16; }
                    }
                    lim_prev = lim; D_prev = D;
                }
                // length 15 is the first half of pair 7; its second half is the sentinel (limit 2^15: never reached, delta 0)
                CD.lim1[7] = lim_a | 32767u << 16;
                CD.dd[7] = dd_a | 0x2000u << 16;
                CD.d1_lo = CD.d1 + (CD.dd[0] & 0xFFFFu) + (CD.dd[0] >> 16);
                if (!ok) { err = 1; active = false; huff = false; }
            }
            if (huff) {
                // distance symbols in canonical order
                for (uint32_t base = nlit & ~7u; wave_any(huff && base < nlit + ndist); base += 8) {
                    if (huff && base < nlit + ndist) {
                        const uint32_t w = lens32[base >> 3];
#pragma unroll
                        for (uint32_t n = 0; n < 8; ++n) {
                            const uint32_t s = base + n, l = (w >> (4u * n)) & 15u;
                            if (s >= nlit && s < nlit + ndist && l != 0u) {
                                const uint32_t idx = b8(kOffLenSym, l);
                                b8(kOffLenSym, l) = (uint8_t)(idx + 1u);
                                if (idx < (uint32_t)kSymEntries) b8(kOffDistSym, idx) = (uint8_t)(s - nlit);
                            }
                        }
                    }
                }
            }
            // ---- literal/length code ------------------------------------------------------------------------------------
            if (huff) {
                uint32_t first = 0, offs = 0, lim_prev = 0, D_prev = 0, lim_a = 0, dd_a = 0, litbase = 0, E = 0;
                int32_t left = 1;
                bool ok = true;
#pragma unroll
                for (uint32_t l = 1; l <= 16; ++l) {
                    uint32_t lim, D;
                    if (l <= 15) {
                        const uint32_t cc = h16(kOffAux, l);
                        const uint32_t c_lit = cc & 0x1FFu, c_non = cc >> 9, c = c_lit + c_non;
                        left = (left << 1) - (int32_t)c;
                        if (left < 0) ok = false;
                        lim = (first + c) << (15u - l);
                        D = offs - first;
                        h16(kOffAux, l) = (uint16_t)(litbase | E << 9);      // insert positions {next literal rank, next non-literal index}
                        litbase += c_lit;
                        E += c_non;
                        offs += c;
                        first = (first + c) << 1;
                        if (l == 2) lit_no12 = offs == 0u;
                        if (l == 14) lit_in14 = lim == 32768u;
                    } else { lim = 32768u; D = D_prev; }
                    if (l == 1) CL.d1 = D & 0x1FFu;
                    else {
                        const uint32_t ddh = ((D - D_prev) & 0x1FFu) | 0x2000u;
                        const uint32_t lm = l - 1;
                        if (lm & 1u) { lim_a = (lim_prev - 1u) & 0xFFFFu; dd_a = ddh; }
                        else { CL.lim1[(lm >> 1) - 1u] = lim_a | ((lim_prev - 1u) & 0xFFFFu) << 16; CL.dd[(lm >> 1) - 1u] = dd_a | ddh << 16; }
                    }
                    lim_prev = lim; D_prev = D;
                }
                CL.lim1[7] = lim_a | 32767u << 16;
                CL.dd[7] = dd_a | 0x2000u << 16;
                CL.d1_lo = CL.d1 + (CL.dd[0] & 0xFFFFu) + (CL.dd[0] >> 16);
                // the fast kernel decodes COMPLETE literal/length codes only: every 15-bit prefix is a symbol, no validity tests per symbol
                if (!ok || left != 0) { err = 1; active = false; huff = false; }
            }
            uint8_t* const tab = io.scratch + kScratchTabs + 256u * (n_seg < (uint32_t)kMaxSeg ? n_seg : 0u);
            if (huff) {
                // literals: rank -> byte into the block's translation table; the others: canonical list in LDS
                for (uint32_t base = 0; wave_any(huff && base < nlit); base += 8) {
                    if (huff && base < nlit) {
                        const uint32_t w = lens32[base >> 3];
#pragma unroll
                        for (uint32_t n = 0; n < 8; ++n) {
                            const uint32_t s = base + n, l = (w >> (4u * n)) & 15u;
                            if (s < nlit && l != 0u) {
                                const uint32_t p = h16(kOffAux, l);
                                if (s < 256u) { h16(kOffAux, l) = (uint16_t)(p + 1u); tab[p & 0xFFu] = (uint8_t)s; }
                                else { h16(kOffAux, l) = (uint16_t)(p + 512u); if ((p >> 9) < (uint32_t)kSymEntries) b8(kOffLenSym, p >> 9) = (uint8_t)(s - 256u); }
                            }
                        }
                    }
                }
                // aux[l] = {litend_l, E_l} from the final insert positions: final[l] = {litbase_{l+1}, E_{l+1}}
                uint32_t fin = h16(kOffAux, 15);
#pragma unroll
                for (uint32_t l = 15; l >= 1; --l) {
                    const uint32_t below = l > 1 ? (uint32_t)h16(kOffAux, l - 1) : 0u;
                    const uint32_t E_l = below >> 9;
                    h16(kOffAux, l) = (uint16_t)(((fin & 0x1FFu) + E_l) | E_l << 9);
                    fin = below;
                }
                info[2 + n_seg] = n_lit;
                ++n_seg;
            }
            // wave-uniform: can the whole wavefront run the plain form of the loop?
            bool plain = lit_no12 && lit_in14 && dist_no12 && dist_in12;
#if defined(__HIP_DEVICE_COMPILE__)
            plain = __builtin_amdgcn_readfirstlane(__all(!huff || plain) ? 1u : 0u) != 0u;
#endif
            if (plain) symbol_loop<true>(CL, CD, io, huff, err, active);
            else symbol_loop<false>(CL, CD, io, huff, err, active);
            if (active && bitpos - lead_bits > io.in_bits) { err = 1; active = false; }
            if (active && last) active = false;
        }
        // ---- the rest of the streams -----------------------------------------------------------------------------------------
        flush_groups();
        {
            uint32_t run_len = n_lit - lit_mark;
            while (wave_any(run_len != 0u)) {
                if (run_len != 0u) {
                    const uint32_t r = run_len > 255u ? 255u : run_len;
                    stage_entry(make_entry2(r, 0, 1));
                    run_len -= r;
                }
                flush_groups();
            }
        }
        if (n_lit & 3u) dw(kOffLitStage, (n_lit >> 2) & 3u) = hi >> (8u * (4u - (n_lit & 3u)));
        if (n_lit & 15u) store16(lit + 16u * (n_lit >> 4), lit_group());
        if (n_ent & 3u) store16(ent + 4u * (n_ent >> 2), ent_group(n_ent >> 2));
        if (err == 0u && opos + n_lit != io.osize) err = 1;
        if (err == 0u && bitpos - lead_bits > io.in_bits) err = 1;
        if (io.live) {
            info[0] = err == 0u ? n_seg : 0u;
            info[1] = n_lit;
        }
        LaneResult R;
        R.status = err == 0u ? 0u : kNeedsGeneral;
        R.n_ent = n_ent;
        R.n_lit = n_lit;
        return R;
    }
};

// RFC 1951 3.2.5 as tables: length symbol 257 + i -> base | extra bits << 9 (u16), distance symbol i -> base | extra bits << 16
SBX_HD void rfc_tables_entry(uint32_t i, uint16_t* len_e, uint32_t* dist_e) {
    const uint32_t t = i - 4u;
    const bool direct = i < 8u || i >= 28u;
    const uint32_t le = direct ? 0u : t >> 2;
    const uint32_t lb = i < 8u ? i + 3u : i >= 28u ? 258u : ((4u + (t & 3u)) << le) + 3u;
    *len_e = (uint16_t)(lb | le << 9);
    const uint32_t de = i < 4u ? 0u : ((i >> 1) - 1u) & 15u;
    const uint32_t db = i < 4u ? i + 1u : ((2u + (i & 1u)) << de) + 1u;
    *dist_e = db | de << 16;
}

}  // namespace inf2
}  // namespace sbx
