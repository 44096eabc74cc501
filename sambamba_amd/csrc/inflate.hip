// inflate.hip -- K1 `bgzf_inflate`: raw-DEFLATE (RFC 1951) decode of BGZF payloads on gfx950.
//
// Replaces decompressBgzfBlock (BioD/bio/core/bgzf/block.d:127-216: zlib inflateInit2(-15) +
// inflate(Z_FINISH), one call per <=64 KiB block, run as std.parallelism tasks from
// BgzfInputStream.fillNextBlock, inputstream.d:414-417).  As in the reference's release build the
// CRC32 trailer is not verified (block.d:187).
//
// Design (MI355X-first, not a zlib translation).  A BAM is hundreds of thousands of independent
// small deflate streams, and DEFLATE has two very different halves, so it is split in two kernels:
//
//  K1a `huffman_decode2` (round 4; the lane program lives in inflate2_core.hpp) + `huffman_decode` (round 3's kernel, now the
//      GENERAL kernel for the blocks the fast one hands over) -- entropy decoding is a serial dependency chain per stream, so the
//      mapping is ONE LANE PER BGZF BLOCK: 64 independent decoders per wavefront, every VALU
//      instruction doing useful work in all lanes.  No lookup tables in memory for the code
//      lengths: the 15 left-justified limits of the literal/length and the distance code live in
//      VGPRs (two 16-bit limits per register) next to per-length deltas, and ONE accumulator of
//      v_dot2_u32_u16 products yields both the code length (1 + sum_l (peek >= limit[l])) and the
//      symbol-index delta -- 3 VALU ops per pair of lengths, branch-free, identical in every lane.
//      The general kernel keeps the literal/length symbol permutation (288 x 9 bits, bit-packed: 324 B) and a
//      32-byte ring of the lane's compressed input in LDS: 356 B per lane, 7 waves per CU.  The fast kernel decodes literal
//      RANKS instead of literal bytes (k_translate_literals maps them afterwards) and needs 176 B per lane and <= 128 VGPRs:
//      14 waves per CU, every wavefront of a chromosome resident at once; it issues ONE vector memory instruction per loop
//      iteration (what such an instruction costs).  Neither decoder touches the LZ77
//      window: they emit the literals (16-byte stores) and one 32-bit
//      entry {literals-before:8, distance-1:15, length:9} per match.  Nothing a lane loads depends on
//      anything it stored, so it never waits on the LZ77 window's memory latency.
//
//  K1b `lz77_resolve`    -- copying matches is data-parallel once positions are known, so the
//      mapping is ONE WAVE PER BGZF BLOCK: 64 entries at a time, output offsets by a DPP prefix sum.
//      The wave keeps the last 2-3 KiB of its output in a linear LDS window: NEAR matches -- the
//      record -> previous record -> ... chain that makes a BAM stream serial -- are resolved LDS -> LDS
//      in rounds (a match may start once everything below its source end is final, i.e. lies below
//      the start of the first unfinished match; self-overlapping matches are extended periodically),
//      FAR matches and literal runs come from global memory with every load of the batch in flight
//      before the first store, and the finished span is written to HBM once, contiguously.
//
// Roofline: both kernels are bound by instruction issue (K1b: the VALU port 91 % busy and 0.51 instructions per cycle and SIMD;
// K1a: one instruction per four cycles with three to four waves per SIMD); neither is HBM-bandwidth bound and their GB/s are
// reported for completeness (DESIGN.md sections 3 and 4).
#include <cstdlib>

#include "common.hpp"
#include "inflate2_core.hpp"
#include "lz77_copy.hpp"
#include "kernels.hpp"

namespace sbx {

namespace {

constexpr int kInfThreads = 64;              // K1a: one wavefront per workgroup, one lane per BGZF block
constexpr int kLaneLds = 356;                // bytes of LDS per lane (89 dwords: odd stride; 7 waves x 64 lanes fit 160 KiB)
constexpr int kLitSymOff = 0;                // 288 x 9 bits: the literal/length symbols in canonical order, bit-packed (324 bytes): one
                                             //          unaligned 16-bit read and one bit-field extract per symbol
constexpr int kLitSymBytes = 324;
constexpr int kRingOff = 324;                // u32[8]   input ring (32 bytes of this lane's compressed payload)
constexpr int kClLenOff = 32;                // u8[19]   code-length code lengths while a dynamic header is parsed (in the
                                             //          literal area, which is rebuilt afterwards; its symbols sit at 0..18)
constexpr int kLenTabBytes = 64, kDistTabBytes = 128;   // symbol -> base / extra bits, shared by the workgroup
constexpr int kLensScratch = 320;            // bytes of global scratch the general kernel uses per block: code lengths being built
constexpr int kScratchStride = inf2::kScratchBytes;     // scratch per block (the fast kernel's layout, inflate2_core.hpp)
constexpr int kGeneralLensOff = inf2::kScratchTabs;     // where the general kernel keeps its code lengths inside it (a block it decodes has no
                                                        // translation tables; the info words in front stay intact)
static_assert(kGeneralLensOff + kLensScratch <= kScratchStride, "scratch layout");

enum : uint32_t {
    INF_OK = 0,
    INF_BAD_BTYPE = 1,
    INF_BAD_STORED = 2,
    INF_BAD_CODELENS = 3,
    INF_BAD_SYMBOL = 4,
    INF_BAD_DISTANCE = 5,
    INF_OUTPUT_OVERRUN = 6,
    INF_INPUT_OVERRUN = 7,
    INF_SIZE_MISMATCH = 8,
};

__constant__ uint8_t kClOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// unaligned dword access (byte-aligned pointers: never cast to uint32_t*)
__device__ __forceinline__ uint32_t ldu32(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
__device__ __forceinline__ void stu32(uint8_t* p, uint32_t v) { __builtin_memcpy(p, &v, 4); }

// ---- token streams between K1a and K1b -----------------------------------------------------------
// literal stream of block b: bytes at lit[lit_off(b) ..], 64-byte aligned, capacity >= isize[b] + 49 (K1a notices an
// output overrun only at the end of a loop iteration, kLitPerIter literals late at most)
// entry stream of block b  : u32 at ent[ent_off(b) ..], 64-byte aligned, capacity >= isize/3 + isize/255 + 11
// Both offsets are pure functions of (out_off[b], b) so that no extra table is needed.
__host__ __device__ __forceinline__ uint64_t lit_off(uint64_t out_off_b, uint32_t b) { return inflate_lit_offset(out_off_b, b); }
__host__ __device__ __forceinline__ uint64_t ent_off(uint64_t out_off_b, uint32_t b) { return (out_off_b / 3 + out_off_b / 255 + 28ull * b + 15ull) & ~15ull; }

__device__ __forceinline__ uint32_t make_entry(uint32_t lit_run, uint32_t len, uint32_t dist) {
    return (lit_run << 24) | ((dist - 1) << 9) | len;      // len == 0: literal-run-only entry
}

typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

// ---- per-lane input: a 32-byte ring in LDS, refilled by wave-synchronous events --------------------
// A lane never waits for the word it needs next (it sits in a register, and the word after that is
// already being read from the LDS ring); the ring itself is topped up 16 bytes at a time by global
// loads that ALL lanes issue in the same iteration (`service`), and a chunk loaded at one event is
// only written to the ring at the next one -- so the vmcnt wait the compiler puts in front of that
// write finds the load long finished, and no other VMEM result is ever consumed in the decode loops.
constexpr int kLitPerIter = 2;       // literal/length symbols decoded per lane and loop iteration (K1a)
constexpr int kRingDwords = 8;
constexpr int kRingLow = 3;          // an event is triggered when some lane has <= this many dwords left

struct BitReader {
    const uint8_t* gp;    // next 16-byte chunk of this lane's payload to fetch from HBM (4-byte aligned)
    uint32_t* ring;       // this lane's 16-dword ring in LDS
    u32x4 pend;           // chunk loaded at the previous event, not yet in the ring
    uint32_t pend_valid;
    uint32_t rpos;        // ring index of the dword AFTER wnext
    uint32_t wpos;        // ring index where the next chunk goes (multiple of 4)
    uint32_t ravail;      // dwords available: wnext + unread ring dwords
    uint32_t wnext;       // next dword of the stream (already in a register)
    uint64_t buf;
    int cnt;              // valid bits in buf
    uint32_t staged;      // chunks put into the ring by service()
    uint32_t cnt0;        // valid bits in buf after init

    __device__ __forceinline__ static u32x4 load16(const uint8_t* p) {
        u32x4 v;
        __builtin_memcpy(&v, p, 16);
        return v;
    }
    __device__ __forceinline__ void put_chunk(u32x4 v) {
        ring[wpos] = v.x; ring[wpos + 1] = v.y; ring[wpos + 2] = v.z; ring[wpos + 3] = v.w;
        wpos = (wpos + 4) & (kRingDwords - 1);
    }
    __device__ __forceinline__ void init(const uint8_t* p, uint32_t* ring_) {
        ring = ring_;
        const int lead = (int)((uintptr_t)p & 3);
        const uint8_t* a = p - lead;
        wpos = 0;
        u32x4 c0 = load16(a), c1 = load16(a + 16);
        put_chunk(c0); put_chunk(c1);
        pend = load16(a + 32);
        pend_valid = 1;
        gp = a + 48;
        buf = (uint64_t)(ring[0] >> (8 * lead));
        cnt = 32 - 8 * lead;
        wnext = ring[1];
        rpos = 2;
        ravail = kRingDwords - 1;
        staged = 0;
        cnt0 = (uint32_t)cnt;
    }
    // wave-synchronous: call with all lanes of the wave converged, once per loop iteration
    __device__ __forceinline__ void service() {
        if (__any(ravail <= (uint32_t)kRingLow)) {
            if (pend_valid && ravail + 4 <= (uint32_t)kRingDwords) {
                put_chunk(pend);
                ravail += 4;
                ++staged;
                pend_valid = 0;
            }
            if (!pend_valid) {
                pend = load16(gp);
                gp += 16;
                pend_valid = 1;
            }
        }
    }
    __device__ __forceinline__ void refill() {   // guarantees cnt > 32 afterwards
        if (cnt <= 32) {
            buf |= (uint64_t)wnext << cnt;
            cnt += 32;
            wnext = ring[rpos];
            rpos = (rpos + 1) & (kRingDwords - 1);
            --ravail;
        }
    }
    __device__ __forceinline__ uint32_t peek(int n) const { return (uint32_t)buf & ((1u << n) - 1u); }
    __device__ __forceinline__ void drop(int n) { buf >>= n; cnt -= n; }
    // bits consumed so far: every refill moved 32 bits into buf, and refills = dwords taken from the ring
    __device__ __forceinline__ uint32_t consumed() const {
        const uint32_t refills = (uint32_t)(kRingDwords - 1) + 4u * staged - ravail;
        return cnt0 + 32u * refills - (uint32_t)cnt;
    }
    __device__ __forceinline__ uint32_t take(int n) { uint32_t v = peek(n); drop(n); return v; }
};

// Canonical code held in registers, two 16-bit lanes per VGPR:
//   lim1[j] = { limit[2j+1] - 1, limit[2j+2] - 1 }   limit[l] = left-justified (15-bit) exclusive upper
//                                                    bound of the codes of length <= l
//   dd[j]   = { D[2j+2] - D[2j+1], D[2j+3] - D[2j+2] } (mod 2^9) | 1 << 13,  D[l] = first symbol index
//                                                    of length l minus first code of length l
// For the 15-bit prefix v:  mask_l = (v >= limit[l]) ; len = 1 + sum mask_l ; D[len] = D[1] + sum mask_l*dd_l.
// Both sums come out of ONE accumulator: a symbol index needs 9 bits, so 16 terms stay below 2^13 and bit
// 13 of every dd counts the masks.  Per pair of lengths: v_pk_sub_i16, v_pk_lshrrev_b16, v_dot2_u32_u16 --
// no memory, no VCC chains, identical in every lane.
struct Code {
    s16x2 lim1[8];
    u16x2 dd[8];
    uint32_t d1;
};

// pairs [kFrom, kTo) of the sum above
template <int kFrom, int kTo>
__device__ __forceinline__ uint32_t decode_pairs(const Code& C, uint32_t v, uint32_t acc) {
    const s16x2 vv = {(short)v, (short)v};
    // all masks first, then the chain of dot products: back to back, a packed subtract, its shift and the dot product that reads it
    // need wait states (s_nop) that occupy the wave's issue slot as an instruction does
    u16x2 m[kTo - kFrom];
#pragma unroll
    for (int j = kFrom; j < kTo; ++j) m[j - kFrom] = (u16x2)(C.lim1[j] - vv);
#pragma unroll
    for (int j = kFrom; j < kTo; ++j) m[j - kFrom] = m[j - kFrom] >> 15;   // 1 where v >= limit
#pragma unroll
    for (int j = kFrom; j < kTo; ++j) acc = __builtin_amdgcn_udot2(m[j - kFrom], C.dd[j], acc, false);
    return acc;
}

// kShortPairs < 8 and `is_short` (wave-uniform): every lane's code is COMPLETE within 2 * kShortPairs bits -- the limit of
// that length is 2^15, no 15-bit prefix reaches any longer length, the remaining pairs would add nothing -- and they are
// skipped.  zlib's codes for a BAM block: the literal/length code reaches 14 bits, never 15, the distance code 12 at most
// (tools/token_stats.cpp).
template <int kShortPairs>
__device__ __forceinline__ void decode_len(const Code& C, uint32_t v, uint32_t is_short, int* len, uint32_t* delta) {
    uint32_t acc = decode_pairs<0, kShortPairs>(C, v, C.d1);
    // (the flag is re-read from its scalar register at every use: hoisted out of the loop, the comparison becomes a lane mask that
    // costs two vector instructions per test; and a real branch: the compiler would otherwise compute the pairs and select)
    asm volatile("" : "+s"(is_short));
    if (kShortPairs < 8 && is_short == 0u) {
        asm volatile("");
        acc = decode_pairs<kShortPairs, 8>(C, v, acc);
    }
    *len = 1 + (int)(acc >> 13);
    *delta = acc;                                        // low 9 bits: the caller masks
}

// 16 small counters of a lane (code length -> count / insert position) packed two per VGPR.  Indexed
// by a per-lane value, so every access is an unrolled compare-select over the 8 registers: ~25 VALU
// ops, used a few hundred times per deflate block -- and 32 bytes of LDS per lane saved, which is what
// lets a seventh wave onto the CU.
struct Pack16 {
    uint32_t r[8];
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = 0;
    }
    __device__ __forceinline__ uint32_t get(uint32_t l) const {
        uint32_t v = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) v = (l >> 1) == (uint32_t)j ? r[j] : v;
        return (v >> (16u * (l & 1u))) & 0xFFFFu;
    }
    __device__ __forceinline__ void add(uint32_t l, uint32_t x) {     // no carry between the halves: values < 2^16
        const uint32_t a = x << (16u * (l & 1u));
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] += (l >> 1) == (uint32_t)j ? a : 0u;
    }
    __device__ __forceinline__ void set(uint32_t l, uint32_t x) { add(l, x - get(l)); }
};

// The <= 30 distance symbols in canonical order, 5 bits each: twelve per 64-bit register pair, the last six in a dword.
struct DistSyms {
    uint64_t a, b;
    uint32_t c;
    __device__ __forceinline__ void clear() { a = b = 0; c = 0; }
    __device__ __forceinline__ void put(uint32_t idx, uint32_t sym) {   // idx < 30, slot still zero
        const uint32_t base = idx >= 24u ? 24u : idx >= 12u ? 12u : 0u;
        const uint64_t v = (uint64_t)sym << (5u * (idx - base));
        a |= idx < 12u ? v : 0ull;
        b |= idx >= 12u && idx < 24u ? v : 0ull;
        c |= idx >= 24u ? (uint32_t)v : 0u;
    }
    __device__ __forceinline__ uint32_t get(uint32_t idx) const {       // idx < 30
        const uint32_t base = idx >= 24u ? 24u : idx >= 12u ? 12u : 0u;
        const uint64_t v = idx >= 24u ? (uint64_t)c : idx >= 12u ? b : a;
        return (uint32_t)(v >> (5u * (idx - base))) & 31u;
    }
};

// Build one canonical code from `n` code lengths in lens[] (global scratch, 1 byte each): symbol
// permutation to LDS, limits/deltas to registers.  Returns false on an over-subscribed code.
// (Incomplete codes are accepted, as zlib accepts the single-code distance tree; an unused code
// decodes as "invalid symbol".)
// *complete_within = the code is complete using lengths <= kWithin only (limit[kWithin] == 2^15).
template <bool kIsLit, int kWithin>
__device__ __forceinline__ bool build_code(const uint8_t* lens, int n, uint8_t* lds, DistSyms& DS, Code& C, bool* complete_within) {
    Pack16 tmp;
    tmp.clear();
    for (int s = 0; s < n; ++s) tmp.add(lens[s] & 15u, 1u);
    uint32_t cnt[16];
#pragma unroll
    for (int l = 0; l < 16; ++l) cnt[l] = (tmp.r[l >> 1] >> (16 * (l & 1))) & 0xFFFFu;
    tmp.clear();
    uint32_t first = 0, offs = 0;
    int32_t left = 1;
    bool ok = true;
    uint32_t lim[17], D[18];
#pragma unroll
    for (int l = 1; l <= 15; ++l) {
        left = (left << 1) - (int32_t)cnt[l];
        if (left < 0) ok = false;
        lim[l] = (first + cnt[l]) << (15 - l);
        D[l] = offs - first;                   // delta of length l (mod 2^16 is all that matters)
        tmp.r[l >> 1] |= offs << (16 * (l & 1));   // running insert position during the sort below
        offs += cnt[l];
        first = (first + cnt[l]) << 1;
    }
    lim[16] = 32768;                           // sentinel: never reached by a 15-bit prefix
    D[16] = D[15];
    D[17] = D[15];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        C.lim1[j] = s16x2{(short)(lim[2 * j + 1] - 1), (short)(lim[2 * j + 2] - 1)};
        C.dd[j] = u16x2{(unsigned short)(((D[2 * j + 2] - D[2 * j + 1]) & 0x1FFu) | 0x2000u),
                        (unsigned short)(((D[2 * j + 3] - D[2 * j + 2]) & 0x1FFu) | 0x2000u)};
    }
    C.d1 = D[1] & 0x1FFu;
    *complete_within = lim[kWithin] == 32768u;
    if (!ok) return false;
    if (kIsLit) {
        for (int i = 0; i < kLitSymBytes / 4; ++i) ((uint32_t*)(lds + kLitSymOff))[i] = 0;
    }
    if (!kIsLit) DS.clear();
    for (int s = 0; s < n; ++s) {
        uint32_t l = lens[s] & 15u;
        if (l) {
            uint32_t idx = tmp.get(l);
            tmp.add(l, 1u);
            if (kIsLit) {
                const uint32_t bo = 9u * idx, v = (uint32_t)s << (bo & 7u);     // 9 + 7 bits: two bytes
                lds[kLitSymOff + (bo >> 3)] |= (uint8_t)v;
                lds[kLitSymOff + (bo >> 3) + 1] |= (uint8_t)(v >> 8);
            } else if (idx < 30u) {
                DS.put(idx, (uint32_t)s);
            }
        }
    }
    return true;
}

// Token emitter of one lane (K1a).  Scattered small stores from 64 lanes are 64 partial-line write requests each, and a
// line that is completed by eight separate 16-byte stores over ~50 loop iterations does not survive in L2 next to the
// 230,000 other open lines of the launch: it goes to HBM half-filled, several times (WRITE_SIZE 2.9x the token bytes,
// profiles/round1).  The emitter therefore keeps the last 16 literal bytes and the last 4 match entries in registers
// (byte / dword shift registers), parks every completed 16-byte group in a register FIFO of kDepth groups and writes a
// stream only in aligned bursts of kDepth back-to-back 16-byte stores (kDepth = 4: a whole 64-byte sector at once; 2: the
// 32 bytes of one write request; the FIFO is shifted with register moves, so a shorter one is fewer instructions).
// The stores are DEFERRED: flush(), which the decode loop calls right after the input-ring service, writes the bursts
// that are full.  gfx9 counts loads and stores in the same in-order vmcnt, so the wait in front of the ring service
// would otherwise also wait for token stores issued moments before -- a store round trip of stall in every iteration;
// this way every VMEM operation of an iteration is issued at its top and has a whole iteration to complete.
template <int kDepth, bool kStream>
struct Emitter {
    uint8_t* lit;         // 64-byte aligned literal stream of this block
    uint32_t* ent;        // 64-byte aligned entry stream of this block
    u32x4 la, ea;         // the last 20 literal bytes (la + lx: a group that has just been completed stays whole while up
    uint32_t lx;          //  to three more literals of the same loop iteration are pushed) / the entry group being filled
    uint32_t n_grp;       // literal groups parked so far
    u32x4 lq[kDepth];     // parked literal groups: the newest in lq[kDepth - 1], the oldest of lq_n in lq[kDepth - lq_n]
    u32x4 eq[kDepth];
    uint32_t lq_n, eq_n;  // parked groups
    uint32_t lq_at, eq_at;    // byte offset / entry index of the oldest parked group
    uint32_t n_lit, n_ent, run;
    // kStream: the token stores bypass the caches' allocation (nontemporal): a 16-byte store into a line L2 does not hold would
    // otherwise fetch the line first
    __device__ __forceinline__ static void store16(void* p, u32x4 v) {
        if (kStream) __builtin_nontemporal_store(v, (u32x4*)p);
        else *(u32x4*)p = v;
    }
    __device__ __forceinline__ void init(uint8_t* l, uint32_t* e) {
        lit = l; ent = e; n_lit = n_ent = run = 0;
        const u32x4 z = {0, 0, 0, 0};
        la = ea = z;
        lx = 0; n_grp = 0;
#pragma unroll
        for (int k = 0; k < kDepth; ++k) lq[k] = eq[k] = z;
        lq_n = eq_n = 0;
        lq_at = eq_at = 0;
    }
    __device__ __forceinline__ void burst_lit() {
#pragma unroll
        for (int k = 0; k < kDepth; ++k) store16(lit + lq_at + 16 * k, lq[k]);
        lq_n = 0;
    }
    __device__ __forceinline__ void burst_ent() {
#pragma unroll
        for (int k = 0; k < kDepth; ++k) store16(ent + eq_at + 4 * k, eq[k]);
        eq_n = 0;
    }
    __device__ __forceinline__ void flush() {
        if (lq_n == (uint32_t)kDepth) burst_lit();
        if (eq_n == (uint32_t)kDepth) burst_ent();
    }
    __device__ __forceinline__ void push_byte(uint32_t byte) {      // {lx, la} = ({lx, la} >> 8) | byte << 152
        la.x = __builtin_amdgcn_alignbit(la.y, la.x, 8);
        la.y = __builtin_amdgcn_alignbit(la.z, la.y, 8);
        la.z = __builtin_amdgcn_alignbit(la.w, la.z, 8);
        la.w = __builtin_amdgcn_alignbit(lx, la.w, 8);
        lx = (lx >> 8) | (byte << 24);
    }
    // Parks the literal group completed since the last call (at most one: call it at least every four literals).  The
    // decode loop has ONE park site per iteration instead of one per literal slot -- with 64 lanes some lane completes
    // a group in almost every slot, so a per-slot site is executed by the whole wave nearly every time.
    __device__ __forceinline__ void park_lits() {
        if ((n_lit >> 4) != n_grp) {
            const uint32_t k = n_lit & 15u;                    // literals already pushed behind the completed group (0..3)
            const uint32_t sh = (4u - k) & 3u;                 // the group starts 4 - k bytes above the bottom of {lx, la}
            u32x4 g;
            g.x = __builtin_amdgcn_alignbyte(la.y, la.x, sh);
            g.y = __builtin_amdgcn_alignbyte(la.z, la.y, sh);
            g.z = __builtin_amdgcn_alignbyte(la.w, la.z, sh);
            g.w = __builtin_amdgcn_alignbyte(lx, la.w, sh);
            if (k == 0u) { g.x = la.y; g.y = la.z; g.z = la.w; g.w = lx; }
            if (lq_n == (uint32_t)kDepth) burst_lit();         // (stored blocks: a group every 16 iterations, flush() keeps up)
            if (lq_n == 0u) lq_at = n_grp * 16u;
#pragma unroll
            for (int q = 0; q + 1 < kDepth; ++q) lq[q] = lq[q + 1];
            lq[kDepth - 1] = g;
            ++lq_n;
            ++n_grp;
        }
    }
    __device__ __forceinline__ void push_entry(uint32_t e) {
        ea.x = ea.y; ea.y = ea.z; ea.z = ea.w; ea.w = e;
        ++n_ent;
        if ((n_ent & 3u) == 0) {
            if (eq_n == (uint32_t)kDepth) burst_ent();         // (a second group within one iteration: split runs only)
            if (eq_n == 0u) eq_at = n_ent - 4;
#pragma unroll
            for (int q = 0; q + 1 < kDepth; ++q) eq[q] = eq[q + 1];
            eq[kDepth - 1] = ea;
            ++eq_n;
        }
    }
    __device__ __forceinline__ void literal(uint32_t byte) {
        push_byte(byte);
        ++n_lit;
        ++run;
    }
    // an entry carries at most 255 literals: longer runs are split here, not in the per-literal path
    __device__ __forceinline__ void split_run() {
        while (run > 255u) { push_entry(make_entry(255, 0, 1)); run -= 255u; }
    }
    __device__ __forceinline__ void match(uint32_t len, uint32_t dist) {
        split_run();
        push_entry(make_entry(run, len, dist));
        run = 0;
    }
    __device__ __forceinline__ void finish() {
        park_lits();
        split_run();
        if (run) { push_entry(make_entry(run, 0, 1)); run = 0; }
        // parked groups (an incomplete burst: the oldest sits in slot kDepth - n)
        {
            const uint32_t skip = (uint32_t)kDepth - lq_n;
#pragma unroll
            for (int k = 0; k < kDepth; ++k)
                if (skip <= (uint32_t)k) store16(lit + lq_at + 16u * ((uint32_t)k - skip), lq[k]);
            lq_n = 0;
        }
        {
            const uint32_t skip = (uint32_t)kDepth - eq_n;
#pragma unroll
            for (int k = 0; k < kDepth; ++k)
                if (skip <= (uint32_t)k) store16(ent + eq_at + 4u * ((uint32_t)k - skip), eq[k]);
            eq_n = 0;
        }
        const uint32_t rl = n_lit & 15u;
        if (rl) {
            for (uint32_t k = rl; k < 16; ++k) push_byte(0);
            store16(lit + (n_lit & ~15u), u32x4{la.y, la.z, la.w, lx});
        }
        const uint32_t re = n_ent & 3u;
        if (re) {
            const uint32_t keep = n_ent;
            for (uint32_t k = re; k < 4; ++k) { ea.x = ea.y; ea.y = ea.z; ea.z = ea.w; ea.w = 0; }
            store16(ent + (keep & ~3u), ea);
        }
    }
};

// One input refill per loop iteration covers both literal/length slots (2 x 15 bits <= the 32 a refill guarantees); the
// output position is not counted per literal but derived (literals + match bytes).
// kDepth: groups per token-store burst (Emitter).  The base value / extra-bit count of a length or distance symbol come from
// two small tables shared by the workgroup in LDS (29 x u16, 30 x u32: different entries lie in different banks, equal ones
// are broadcast) instead of ~12 VALU instructions of arithmetic each, and the canonical decode skips the pairs of code
// lengths no lane's code uses (decode_len).
template <int kDepth, bool kStream>
__global__ __launch_bounds__(kInfThreads) void k_huffman_decode(
    const uint8_t* __restrict__ comp, const uint64_t* __restrict__ comp_off, const uint32_t* __restrict__ comp_len,
    const uint32_t* __restrict__ isize, const uint64_t* __restrict__ out_off, uint32_t n_blocks, uint32_t block0,
    uint8_t* __restrict__ lit_stream, uint32_t* __restrict__ ent_stream, uint32_t* __restrict__ n_entries,
    uint8_t* __restrict__ lens_scratch, uint32_t* __restrict__ status, unsigned long long* __restrict__ tok_bytes, uint32_t only_flagged) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    // Lanes past the last block shadow block n_blocks-1 but stay inactive: every lane of the wave
    // must take part in the wave-synchronous input service.
    // only_flagged: this launch follows the fast kernel (k_huffman_decode2) and decodes the blocks that one handed over
    // (status kNeedsGeneral); a wavefront none of whose blocks is flagged ends here.
    const uint32_t b_raw = blockIdx.x * kInfThreads + threadIdx.x;
    const uint32_t b = b_raw < n_blocks ? b_raw : n_blocks - 1;
    const bool live = b_raw < n_blocks && (only_flagged == 0u || status[b] == inf2::kNeedsGeneral);
    if (only_flagged != 0u && !__any(live)) return;
    uint8_t* lds = smem + threadIdx.x * kLaneLds;
    uint8_t* lens = lens_scratch + (size_t)b * kScratchStride + kGeneralLensOff;
    DistSyms DS;
    DS.clear();
    // RFC 1951 3.2.5 as tables behind the lanes' areas: length symbol 257 + i -> base | extra bits << 9,
    // distance symbol i -> base | extra bits << 16 (entries 30, 31 exist so that an invalid symbol reads something)
    uint16_t* const len_tab = (uint16_t*)(smem + kInfThreads * kLaneLds);
    uint32_t* const dist_tab = (uint32_t*)(smem + kInfThreads * kLaneLds + kLenTabBytes);
    {
        const uint32_t i = threadIdx.x;
        if (i < 32u) {
            const uint32_t t = i - 4u;
            const bool direct = i < 8u || i >= 28u;
            const uint32_t le = direct ? 0u : t >> 2;
            const uint32_t lb = i < 8u ? i + 3u : i >= 28u ? 258u : ((4u + (t & 3u)) << le) + 3u;
            len_tab[i] = (uint16_t)(lb | le << 9);
            const uint32_t de = i < 4u ? 0u : ((i >> 1) - 1u) & 15u;
            const uint32_t db = i < 4u ? i + 1u : ((2u + (i & 1u)) << de) + 1u;
            dist_tab[i] = db | de << 16;
        }
        __syncthreads();
    }

    const uint8_t* in = comp + comp_off[b];
    const uint32_t in_bits = comp_len[b] * 8u;
    const uint32_t osize = isize[b];
    const uint64_t oo = out_off[b];
    uint32_t opos = 0;          // bytes produced by matches (literals and stored bytes are counted by the emitter)
    uint32_t err = INF_OK;

    Emitter<kDepth, kStream> em;   // (lanes past the last block stay inactive and never emit)
    em.init(lit_stream + lit_off(oo, block0 + b), ent_stream + ent_off(oo, block0 + b));
    BitReader br;
    br.init(in, (uint32_t*)(lds + kRingOff));
    Code CL, CD;
#pragma unroll
    for (int j = 0; j < 8; ++j) { CL.lim1[j] = s16x2{0, 0}; CL.dd[j] = u16x2{0, 0}; CD.lim1[j] = s16x2{0, 0}; CD.dd[j] = u16x2{0, 0}; }
    CL.d1 = CD.d1 = 0;

    // The control flow below keeps all 64 lanes inside the same loops (a lane that is done or has
    // failed idles with `active == false`) so that br.service() is always executed converged.
    bool active = live && !(osize == 0 && comp_len[b] == 0);   // nothing to do for an empty payload
    bool last = false;
    while (__any(active)) {
        // ---- block header ---------------------------------------------------------------
        uint32_t btype = 3;
        if (active) {
            br.refill();
            last = br.take(1) != 0;
            btype = br.take(2);
            if (btype == 3) { err = INF_BAD_BTYPE; active = false; }
        }
        br.service();
        // stored block: skip to the byte boundary, LEN, NLEN, raw bytes
        uint32_t stored_left = 0;
        if (active && btype == 0) {
            br.drop(br.cnt & 7);
            br.refill();
            uint32_t len = br.take(16);
            br.refill();
            uint32_t nlen = br.take(16);
            if ((len ^ 0xFFFFu) != nlen) { err = INF_BAD_STORED; active = false; }
            else if (opos + em.n_lit + len > osize) { err = INF_OUTPUT_OVERRUN; active = false; }
            else { stored_left = len; }
        }
        while (__any(stored_left != 0)) {
            if (stored_left) {
                br.refill();
                em.literal(br.take(8));
                --stored_left;
            }
            em.park_lits();
            br.service();
        }
        int nlit = 0, ndist = 0;
        const bool huff = active && btype != 0;
        if (active && btype == 1) {
            // fixed code (RFC 1951 3.2.6)
            for (int s = 0; s < 288; ++s) lens[s] = (uint8_t)(s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8);
            for (int s = 0; s < 30; ++s) lens[288 + s] = 5;
            nlit = 288;
            ndist = 30;
        }
        // dynamic code: code-length code, then nlit + ndist lengths
        uint32_t ccnt[8];
#pragma unroll
        for (int l = 0; l < 8; ++l) ccnt[l] = 0;
        uint8_t* cl_len = lds + kClLenOff;     // [19] parked in the literal symbol area (rebuilt below)
        uint8_t* cl_sym = lds + kLitSymOff;    // [19] symbols sorted by (length, value)
        int ncl_left = 0, cl_i = 0;
        bool dyn = active && btype == 2;
        if (dyn) {
            br.refill();
            nlit = (int)br.take(5) + 257;
            ndist = (int)br.take(5) + 1;
            ncl_left = (int)br.take(4) + 4;
            if (nlit > 286 || ndist > 30) { err = INF_BAD_CODELENS; active = false; dyn = false; ncl_left = 0; }
            for (int s = 0; s < 19; ++s) cl_len[s] = 0;
        }
        while (__any(ncl_left != 0)) {
            if (ncl_left) {
                br.refill();
                cl_len[kClOrder[cl_i++]] = (uint8_t)br.take(3);
                --ncl_left;
            }
            br.service();
        }
        if (dyn) {
            int k = 0;
            for (int l = 1; l <= 7; ++l)
                for (int s2 = 0; s2 < 19; ++s2)
                    if (cl_len[s2] == l) cl_sym[k++] = (uint8_t)s2;
            for (int s2 = 0; s2 < 19; ++s2) {
                uint32_t l = cl_len[s2];
#pragma unroll
                for (int q = 1; q < 8; ++q) ccnt[q] += (l == (uint32_t)q) ? 1u : 0u;
            }
            int32_t left = 1;
            bool ok = true;
#pragma unroll
            for (int l = 1; l <= 7; ++l) { left = (left << 1) - (int32_t)ccnt[l]; if (left < 0) ok = false; }
            if (!ok) { err = INF_BAD_CODELENS; active = false; dyn = false; }
        }
        {
            int i = 0;
            uint32_t prev = 0;
            const int total = nlit + ndist;
            bool more = dyn;
            while (__any(more)) {
                if (more) {
                    br.refill();
                    // canonical walk over lengths 1..7 with the counts in registers
                    uint32_t code = 0, first = 0, index = 0, sym = 0xFFFFFFFFu;
                    int used = 0;
#pragma unroll
                    for (int l = 1; l <= 7; ++l) {
                        code |= (uint32_t)(br.buf >> (l - 1)) & 1u;
                        uint32_t c = ccnt[l];
                        if (used == 0 && code < first + c) {
                            sym = cl_sym[index + (code - first)];
                            used = l;
                        }
                        index += c;
                        first = (first + c) << 1;
                        code <<= 1;
                    }
                    if (sym >= 19) { err = INF_BAD_CODELENS; more = false; }
                    else {
                        br.drop(used);
                        if (sym < 16) {
                            lens[i++] = (uint8_t)sym;
                            prev = sym;
                        } else {
                            uint32_t rep, val;
                            if (sym == 16) { val = prev; rep = 3 + br.take(2); if (i == 0) { err = INF_BAD_CODELENS; more = false; rep = 0; } }
                            else if (sym == 17) { val = 0; rep = 3 + br.take(3); }
                            else { val = 0; rep = 11 + br.take(7); }
                            if (i + (int)rep > total) { err = INF_BAD_CODELENS; more = false; }
                            else {
                                for (uint32_t k = 0; k < rep; ++k) lens[i++] = (uint8_t)val;
                                prev = val;   // (after 17/18 a following 16 repeats 0, as in zlib)
                            }
                        }
                        if (i >= total) more = false;
                    }
                }
                br.service();
            }
            if (dyn && err == INF_OK && lens[256] == 0) err = INF_BAD_CODELENS;   // no end-of-block code
            if (err != INF_OK) active = false;
        }
        bool sym_loop = huff && active;
        bool lit_in_14 = true, dist_in_12 = true;
        if (sym_loop) {
            if (!build_code<true, 14>(lens, nlit, lds, DS, CL, &lit_in_14)) { err = INF_BAD_CODELENS; active = false; sym_loop = false; }
            else if (!build_code<false, 12>(lens + nlit, ndist, lds, DS, CD, &dist_in_12)) { err = INF_BAD_CODELENS; active = false; sym_loop = false; }
        }
        // wave-uniform: every decoding lane's code is complete within 14 / 12 bits (decode_len)
        // (in scalar registers: a flag the compiler keeps as a lane mask costs two vector instructions per test)
        uint32_t short_lit = __builtin_amdgcn_readfirstlane(__all(!sym_loop || lit_in_14) ? 1u : 0u);
        uint32_t short_dist = __builtin_amdgcn_readfirstlane(__all(!sym_loop || dist_in_12) ? 1u : 0u);
        asm volatile("" : "+s"(short_lit), "+s"(short_dist));

        // ---- symbol loop -----------------------------------------------------------------
        // Under SIMT the (long) match path is paid by the whole wave in every iteration in which any
        // lane has a match, so each iteration first decodes up to kLitPerIter literal/length symbols
        // per lane -- literals are emitted on the spot, the first length symbol (or end-of-block) stops
        // the lane's run -- and then handles at most one match per lane.
        if (__any(sym_loop)) do {
            br.service();               // every VMEM operation of the iteration is issued here, at its top:
            em.flush();                 // one ring load and the token stores parked by the previous iteration
            uint32_t msym = 0;          // pending length symbol (257..285) of this lane, 0 = none
            uint32_t bad = INF_OK;
            static_assert(kLitPerIter * 15 <= 32, "one refill must cover the literal/length slots of an iteration");
            if (sym_loop) br.refill();
#pragma unroll
            for (int r = 0; r < kLitPerIter; ++r) {
                if (sym_loop && msym == 0 && bad == INF_OK) {
                    const uint32_t v = __brev((uint32_t)br.buf) >> 17;
                    int len;
                    uint32_t delta;
                    decode_len<7>(CL, v, short_lit, &len, &delta);
                    const int lc = len > 15 ? 15 : len;
                    const uint32_t idx0 = (delta + (v >> (15 - lc))) & 0x1FFu;
                    const uint32_t idx = idx0 > 287u ? 287u : idx0;
                    const uint32_t bo = 9u * idx;
                    uint16_t packed;
                    __builtin_memcpy(&packed, lds + kLitSymOff + (bo >> 3), 2);
                    const uint32_t sym = ((uint32_t)packed >> (bo & 7u)) & 0x1FFu;
                    br.drop(lc);
                    const bool ok = len <= 15 && idx0 <= 287u && sym <= 285u;
                    if (ok && sym < 256u) em.literal(sym);             // (overrun: checked once per iteration below)
                    sym_loop = !(ok && sym == 256u);
                    msym = ok && sym > 256u ? sym : 0u;
                    bad = ok ? INF_OK : INF_BAD_SYMBOL;
                }
            }
            em.park_lits();
            const uint32_t opos_now = opos + em.n_lit;
            if (opos_now > osize && bad == INF_OK) { bad = INF_OUTPUT_OVERRUN; msym = 0; }
            if (msym != 0) {
                // One refill covers the whole match: <= 5 length-extra + 15 code + 13 distance-extra bits.
                br.refill();
                // match length (RFC 1951 3.2.5): base and extra bits from the shared table
                const uint32_t lt = len_tab[msym - 257u];
                const uint32_t le = lt >> 9, lb = lt & 0x1FFu;
                const uint32_t mlen = lb + br.take((int)le);
                const uint32_t dv = __brev((uint32_t)br.buf) >> 17;
                int dl;
                uint32_t ddelta;
                decode_len<6>(CD, dv, short_dist, &dl, &ddelta);
                const int dc = dl > 15 ? 15 : dl;
                const uint32_t didx0 = (ddelta + (dv >> (15 - dc))) & 0x1FFu;
                const uint32_t dsym = DS.get(didx0 > 29u ? 29u : didx0);
                br.drop(dc);
                // distance (RFC 1951 3.2.5)
                const uint32_t dt = dist_tab[dsym];
                const uint32_t de = dt >> 16, db = dt & 0xFFFFu;
                const uint32_t dist = db + br.take((int)de);
                const bool code_ok = dl <= 15 && didx0 < 30u && dsym <= 29u && dist <= opos_now;
                const bool fits = opos_now + mlen <= osize;
                if (code_ok && fits) { opos += mlen; em.match(mlen, dist); }
                bad = !code_ok ? (uint32_t)INF_BAD_DISTANCE : !fits ? (uint32_t)INF_OUTPUT_OVERRUN : bad;
            }
            if (bad != INF_OK) { err = bad; active = false; sym_loop = false; }
        } while (__any(sym_loop));
        if (active && last) active = false;
    }
    em.finish();
    if (err == INF_OK && opos + em.n_lit != osize) err = INF_SIZE_MISMATCH;
    if (err == INF_OK && br.consumed() > in_bits) err = INF_INPUT_OVERRUN;
    if (live) {
        status[b] = err;
        n_entries[b] = em.n_ent;
    }
    if (tok_bytes) {
        // bytes of the two token streams (accounting): one atomic per wave into one of 64 accumulators
        unsigned long long t = live ? (unsigned long long)em.n_lit + 4ull * em.n_ent : 0ull;
        for (int d = 32; d >= 1; d >>= 1) t += __shfl_xor(t, d, 64);
        if (threadIdx.x == 0) atomicAdd(tok_bytes + (blockIdx.x & 63u), t);
    }
}


// ---- K1a, round 4: the fast kernel -------------------------------------------------------------------------------------
// One lane per BGZF block, the lane program of inflate2_core.hpp (literal ranks instead of literal bytes, windows instead of a
// bit buffer, build state in LDS): 176 bytes of LDS per lane and <= 128 VGPRs, so that 14 wavefronts share a CU (round 3's
// kernel: 7) and the 3,379 wavefronts of a chromosome-sized file are resident at once.  Blocks it does not decode are flagged
// kNeedsGeneral and decoded by k_huffman_decode in a second launch.
constexpr int kInf2Threads = 64;
__global__ __launch_bounds__(kInf2Threads) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_huffman_decode2(
    const uint8_t* __restrict__ comp, const uint64_t* __restrict__ comp_off, const uint32_t* __restrict__ comp_len,
    const uint32_t* __restrict__ isize, const uint64_t* __restrict__ out_off, uint32_t n_blocks, uint32_t block0,
    uint8_t* __restrict__ lit_stream, uint32_t* __restrict__ ent_stream, uint32_t* __restrict__ n_entries,
    uint8_t* __restrict__ scratch, uint32_t* __restrict__ status, unsigned long long* __restrict__ tok_bytes) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t lane = threadIdx.x;
    const uint32_t b_raw = blockIdx.x * kInf2Threads + lane;
    const bool live = b_raw < n_blocks;
    const uint32_t b = live ? b_raw : n_blocks - 1;
    uint16_t* const len_tab = (uint16_t*)(smem + inf2::kWaveLds);
    uint32_t* const dist_tab = (uint32_t*)(smem + inf2::kWaveLds + inf2::kLenTabBytes);
    if (lane < 32u) inf2::rfc_tables_entry(lane, &len_tab[lane], &dist_tab[lane]);
    __syncthreads();
    const uint64_t oo = out_off[b];
    inf2::LaneIo io;
    io.in = comp + comp_off[b];
    io.in_bits = comp_len[b] * 8u;
    io.osize = isize[b];
    io.lit = lit_stream + lit_off(oo, block0 + b);
    io.ent = ent_stream + ent_off(oo, block0 + b);
    io.scratch = scratch + (size_t)b * kScratchStride;
    io.live = live;
    inf2::Lane L;
    const inf2::LaneResult R = L.run(io, smem, lane, len_tab, dist_tab);
    if (live) {
        status[b] = R.status;
        n_entries[b] = R.n_ent;
    }
    if (tok_bytes) {
        unsigned long long t = live && R.status == 0u ? (unsigned long long)R.n_lit + 4ull * R.n_ent : 0ull;
        for (int d = 32; d >= 1; d >>= 1) t += __shfl_xor(t, d, 64);
        if (lane == 0) atomicAdd(tok_bytes + (blockIdx.x & 63u), t);
    }
}

// Literal ranks -> literal bytes, in place in the literal stream, between K1a and K1b: one wavefront per BGZF block, the block's
// (<= kMaxSeg) tables of 256 bytes in LDS -- a table spans exactly the 64 banks, so the 64 lanes' byte lookups never conflict --
// every lane translates 16-byte groups.  A group that straddles two deflate blocks picks the table per byte.
__global__ __launch_bounds__(64) void k_translate_literals(
    uint8_t* __restrict__ lit_stream, const uint64_t* __restrict__ out_off, uint32_t n_blocks, uint32_t block0,
    const uint8_t* __restrict__ scratch, const uint32_t* __restrict__ status) {
    __shared__ __attribute__((aligned(16))) uint8_t tabs[inf2::kMaxSeg * 256];
    __shared__ uint32_t seg_start[inf2::kMaxSeg + 1];
    const uint32_t b = blockIdx.x, lane = threadIdx.x;
    if (status[b] != INF_OK) return;
    const uint8_t* sc = scratch + (size_t)b * kScratchStride;
    const uint32_t* info = (const uint32_t*)(sc + inf2::kScratchInfo);
    const uint32_t n_seg = info[0], n_lit = info[1];
    if (n_seg == 0u || n_lit == 0u) return;
    for (uint32_t s = 0; s < n_seg; ++s) ((uint32_t*)tabs)[64u * s + lane] = ((const uint32_t*)(sc + inf2::kScratchTabs))[64u * s + lane];
    if (lane <= n_seg) seg_start[lane] = lane < n_seg ? info[2 + lane] : n_lit;
    __syncthreads();
    uint8_t* lit = lit_stream + lit_off(out_off[b], block0 + b);
    const uint32_t n_grp = (n_lit + 15u) >> 4;
    for (uint32_t g = lane; g < n_grp; g += 64) {
        u32x4 v = *(const u32x4*)(lit + 16u * g);
        const uint32_t p0 = 16u * g;
        // segment of the group's first byte (segments are few: a linear search)
        uint32_t s0 = 0;
        while (s0 + 1u < n_seg && seg_start[s0 + 1u] <= p0) ++s0;
        const uint32_t next = seg_start[s0 + 1u];
        uint32_t w[4] = {v.x, v.y, v.z, v.w};
        if (next >= p0 + 16u || s0 + 1u >= n_seg) {
            const uint8_t* t = tabs + 256u * s0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t x = w[k];
                w[k] = (uint32_t)t[x & 0xFFu] | (uint32_t)t[(x >> 8) & 0xFFu] << 8 | (uint32_t)t[(x >> 16) & 0xFFu] << 16 | (uint32_t)t[x >> 24] << 24;
            }
        } else {
            for (uint32_t i = 0; i < 16; ++i) {
                const uint32_t p = p0 + i;
                uint32_t sg = s0;
                while (sg + 1u < n_seg && seg_start[sg + 1u] <= p) ++sg;
                const uint32_t x = (w[i >> 2] >> (8u * (i & 3u))) & 0xFFu;
                const uint32_t y = tabs[256u * sg + x];
                w[i >> 2] = (w[i >> 2] & ~(0xFFu << (8u * (i & 3u)))) | y << (8u * (i & 3u));
            }
        }
        *(u32x4*)(lit + 16u * g) = u32x4{w[0], w[1], w[2], w[3]};
    }
}

// ---- K1b -----------------------------------------------------------------------------------------
// One wave per BGZF block, four blocks per workgroup.  The wave keeps the most recent kHist..kCap
// bytes of its output in a linear LDS buffer (offset = position - base; the buffer is slid down every
// ~1 KiB of output, so there is no wrap-around anywhere):
//   * literal runs and FAR matches (source below `base`, i.e. output this wave flushed to HBM at least
//     a batch ago -- final, no dependency) are fetched from global memory, all loads of a batch in
//     flight together: one memory round trip per batch;
//   * NEAR matches -- the ones that form the long dependency chain record -> previous record ->
//     ... of a BAM stream -- are resolved LDS -> LDS in rounds that cost an LDS round trip each;
//   * the finished span of the batch is written to HBM once, contiguous and coalesced.
constexpr int kResThreads = 256;
// Window geometry (template parameters of the kernel): kHist = bytes of history guaranteed to be in LDS, kSpanMax = output
// bytes per batch (a batch is <= 64 entries AND <= this), kCap = LDS bytes per wave = kHist + slide hysteresis (1 KiB) +
// kSpanMax.  A sweep on config 2 (profiles/round2/README.md): 1 KiB / 1 KiB, 1 / 1.5, 2 / 1 and 2 / 1.5 KiB are within
// 0.1 ms of each other (23.9-24.0 ms); anything larger costs occupancy (2 / 2: 25.3, 3 / 1.5: 25.5, 4 / 2 KiB: 29.8 ms) --
// the kernel is bound by VALU issue, not by where its bytes come from.
constexpr uint32_t kHistDefault = 2048, kSpanDefault = 1536;

// inclusive prefix sum over the 64 lanes: four row-shift steps inside each row of 16 lanes, then the row
// totals are broadcast across rows (DPP row_bcast15 / row_bcast31) -- VALU only, no LDS crossbar trips
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, uint32_t lane) {
    (void)lane;
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);    // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);    // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);    // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);    // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);   // row_bcast:15 -> rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);   // row_bcast:31 -> rows 2, 3
    return v;
}

// (the short copy of a lane -- Short16 -- and the expansion of short periodic matches live in lz77_copy.hpp: the CPU harness runs them too)
using lz::Short16;

// Long copy tasks (16 < n <= 512) of a wave, done by all 64 lanes, 8 bytes per lane, four tasks in
// flight (loads of all four before the first store).  Lane r of `m` owns a task: n bytes from
// sbase + s to buf + d.  Source and destination of a task never overlap.
// (four tasks per turn of the loop; two or one per turn -- a batch of a BAM stream holds 1.6 long far matches and 0.2 long literal runs,
// tools/token_stats.cpp -- execute 8 % fewer vector instructions and are not faster: profiles/round5/README.md)
__device__ __forceinline__ void coop_copy(uint64_t m, const uint8_t* sbase, uint32_t s, uint8_t* buf, uint32_t d, uint32_t n, uint32_t lane) {
    while (m) {
        uint32_t S[4], D[4], N[4], wa[4], wb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            N[j] = 0; S[j] = 0; D[j] = 0;
            if (m) {
                const int r = __builtin_ctzll(m);
                m &= m - 1;
                S[j] = __builtin_amdgcn_readlane(s, r);
                D[j] = __builtin_amdgcn_readlane(d, r);
                N[j] = __builtin_amdgcn_readlane(n, r);
            }
        }
        const uint32_t off = 8 * lane;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (off < N[j]) {
                const uint32_t a = off + 4 <= N[j] ? off : N[j] - 4, c = off + 8 <= N[j] ? off + 4 : N[j] - 4;
                wa[j] = ldu32(sbase + S[j] + a);
                wb[j] = ldu32(sbase + S[j] + c);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (off < N[j]) {
                const uint32_t a = off + 4 <= N[j] ? off : N[j] - 4, c = off + 8 <= N[j] ? off + 4 : N[j] - 4;
                stu32(buf + D[j] + a, wa[j]);
                stu32(buf + D[j] + c, wb[j]);
            }
        }
    }
}

// the whole wave copies N bytes inside LDS (wave-uniform arguments, ranges disjoint)
__device__ __forceinline__ void wave_copy(uint8_t* d, const uint8_t* s, uint32_t N, uint32_t lane) {
    if (N >= 4) {
        for (uint32_t off = 4 * lane; off < N; off += 256) {
            const uint32_t o2 = off + 4 <= N ? off : N - 4;
            stu32(d + o2, ldu32(s + o2));
        }
    } else if (lane < N) {
        d[lane] = s[lane];
    }
}

// kOwn32 (variant 1): what the token statistics of BAM streams say a batch spends its instructions on (tools/token_stats.cpp):
// 99.5 % of the near matches longer than 16 bytes are at most 32 bytes long, and so are most far ones -- they no longer go
// through the cooperative copy loop (three readlanes per task, two dwords per lane, 3 to 19 of 64 lanes busy) but are copied
// by their own lane with a second 16-byte step; and a short self-overlapping match (a run, a dinucleotide repeat: 0.8 per
// batch, 84 % with a period of at most 8) is no longer expanded byte by byte but with four byte permutes of the period
// (v_perm_b32, selectors per period from a 128-byte table in LDS).
// (Round 4's candidate -- near matches resolved per OUTPUT BYTE through origin pointers, `k_lz77_resolve_jump` -- ran on the device in
// round 5: correct, and 37 % SLOWER than this kernel (30.6 against 22.4 ms on config 2; more instructions of every kind, not fewer:
// profiles/round5/README.md).  It is gone; what it left is the lesson that the rounds of phase B are cheap -- see kExact below.)
// (kAblate: tools/k1_lab compiles parts of the batch loop out to time them; 0 in the product)
// (kTail16: the second 16 bytes of a 17 .. 32-byte own-lane copy as ONE 16-byte word at offset n - 16 -- it overlaps what the first step
// copied -- instead of a second Short16 step: 3.16 -> 3.05 ms at 40 Mbp, profiles/round6/call_j_k1b_tail16_40Mbp.jsonl; false: the A/B partner)
template <uint32_t kHist, uint32_t kSpanMax, bool kOwn32, bool kExact, uint32_t kAblate = 0, bool kTail16 = true>
__device__ __forceinline__ void lz77_resolve_body(
    const uint8_t* __restrict__ lit_stream, const uint32_t* __restrict__ ent_stream, const uint32_t* __restrict__ n_entries,
    const uint64_t* __restrict__ out_off, const uint32_t* __restrict__ isize, uint32_t n_blocks, uint32_t block0,
    uint8_t* out, const uint32_t* __restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    // (measured, profiles/round3: fewer instructions do not make this kernel faster -- telling the compiler that the wave index is
    // uniform (readfirstlane: the block's state and the batch loop in scalar registers, 879 -> 791 vector instructions) costs 6 %,
    // copy offsets as min(4 k, n - 4) instead of compare + select (772 instructions, 15 instead of 73 s_nop) cost 3.5 %, per-batch
    // scalar flags for the rare kinds of matches change nothing)
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t b = blockIdx.x * (kResThreads / 64) + wv;
    if (kOwn32) {
        uint32_t* tab = (uint32_t*)(smem + (kResThreads / 64) * (kHist + 1024u + kSpanMax + 16u));
        if (threadIdx.x < 32u) {
            const uint32_t d = (threadIdx.x >> 2) + 1u, j = threadIdx.x & 3u;
            tab[threadIdx.x] = lz::period_selector(d, j);
        }
        __syncthreads();
    }
    if (b >= n_blocks) return;
    if (status[b] != INF_OK) return;
    constexpr uint32_t kCap = kHist + 1024u + kSpanMax, kWaveLds = kCap + 16u;
    uint8_t* buf = smem + wv * kWaveLds;
    // byte selectors of a period: sel[d - 1][j] = the four bytes ((4 j + i) mod d), i = 0..3 -- output dword j of a match of period d
    const uint32_t* per_sel = (const uint32_t*)(smem + (kResThreads / 64) * kWaveLds);
    const uint64_t oo = out_off[b];
    const uint8_t* lit = lit_stream + lit_off(oo, block0 + b);
    const uint32_t* ent = ent_stream + ent_off(oo, block0 + b);
    uint8_t* o = out + oo;
    const uint32_t ne = n_entries[b];
    uint32_t opos = 0, lpos = 0;   // wave-uniform running positions (block-relative)
    uint32_t base = 0;             // position held at buf[0]; multiple of 16

    uint32_t e_next = lane < ne ? ent[lane] : 0u;
    for (uint32_t e0 = 0; e0 < ne;) {
        uint32_t e = e_next;
        const uint32_t in_batch = ne - e0 < 64u ? ne - e0 : 64u;
        // ---- positions: one packed scan (64 * 513 < 2^16), batch cut at kSpanMax bytes of output ----------
        uint32_t lr = e >> 24, len = e & 511u;
        const uint32_t dist = ((e >> 9) & 0x7FFFu) + 1u;
        uint32_t tot = lr + len;
        const uint32_t pk = wave_incl_scan(tot | (lr << 16), lane);
        const uint32_t take = __popcll(__ballot(lane < in_batch && (pk & 0xFFFFu) <= kSpanMax));   // >= 1: an entry is <= 513 bytes
        if (lane >= take) { lr = 0; len = 0; tot = 0; }
        const uint32_t pk_last = __builtin_amdgcn_readlane(pk, take - 1);
        const uint32_t span = pk_last & 0xFFFFu, lspan = pk_last >> 16;
        e0 += take;
        e_next = (e0 + lane < ne) ? ent[e0 + lane] : 0u;            // next batch (wherever this one was cut)
        // ---- slide the window down when this batch might not fit -----------------------------------------
        if (!(kAblate & 32u) && opos - base + kSpanMax > kCap) {
            const uint32_t nb = (opos - kHist) & ~15u, delta = nb - base, keep = opos - nb;
            for (uint32_t i = 16 * lane; i < keep; i += 1024) {
                const u32x4 v = *(const u32x4*)(buf + delta + i);
                *(u32x4*)(buf + i) = v;
            }
            base = nb;
        }
        const uint32_t eo = opos + (pk & 0xFFFFu) - tot;     // first output byte of this entry (its literal run)
        const uint32_t el = lpos + (pk >> 16) - lr;          // its first literal in the block's literal stream
        const uint32_t dst = eo + lr;                        // first byte of the match
        const uint32_t src = dst - dist;
        const bool far = len != 0 && src < base;             // (dist <= dst for every entry K1a emitted)
        // ---- phase A: everything that comes from global memory, all loads before the first store ---------
        {
            constexpr uint32_t kOwn = kOwn32 ? 32u : 16u;       // bytes a lane copies itself
            // own-lane copies in steps of 16 bytes: one step, or two when some item of the batch is 17 .. 32 bytes long (the same code
            // and registers for both steps -- what matters is that the kernel keeps its 8 waves per SIMD)
            const uint32_t own_l = (kAblate & 1u) ? 0u : lr <= kOwn ? lr : 0u, own_f = (kAblate & 2u) ? 0u : far && len <= kOwn ? len : 0u;
            const uint32_t steps = (kAblate & 16u) ? 1u : kOwn32 && __any((own_l | own_f) > 16u) ? 2u : 1u;
            for (uint32_t h = 0; h < (kTail16 ? 1u : steps); ++h) {
                const uint32_t n_l = own_l > 16u * h ? (own_l - 16u * h < 16u ? own_l - 16u * h : 16u) : 0u;
                const uint32_t n_f = own_f > 16u * h ? (own_f - 16u * h < 16u ? own_f - 16u * h : 16u) : 0u;
                Short16 rl, rf;
                rl.load(lit + el + 16u * h, n_l);
                rf.load(o + src + 16u * h, n_f);
                rl.store(buf + (eo - base) + 16u * h, n_l);
                rf.store(buf + (dst - base) + 16u * h, n_f);
            }
            if (kTail16 && steps > 1u) {
                // 17 .. 32 bytes: the last 16 of them as one word (it overlaps what the first step copied)
                // (a lane has a long literal run or a long far match, rarely both: ONE word per lane, the literal run first)
                const bool tl = own_l > 16u, tf = !tl && own_f > 16u;
                const uint8_t* ts = tl ? lit + el + own_l - 16u : o + src + own_f - 16u;
                const uint32_t td = tl ? (eo - base) + own_l - 16u : (dst - base) + own_f - 16u;
                u32x4 tv;
                if (tl || tf) { __builtin_memcpy(&tv, ts, 16); __builtin_memcpy(buf + td, &tv, 16); }
                if (__any(tl && own_f > 16u)) {
                    if (tl && own_f > 16u) { __builtin_memcpy(&tv, o + src + own_f - 16u, 16); __builtin_memcpy(buf + (dst - base) + own_f - 16u, &tv, 16); }
                }
            }
            if (!(kAblate & (1u | 64u))) coop_copy(__ballot(lr > kOwn), lit, el, buf, eo - base, lr, lane);
            if (!(kAblate & (2u | 64u))) coop_copy(__ballot(far && len > kOwn), o, src, buf, dst - base, len, lane);
        }
        // ---- phase B: near matches, LDS -> LDS --------------------------------------------------------------
        // A match may start once everything below its source end is final.  Matches start in entry order,
        // so "final" is everything below the start of the first unfinished match (the frontier F); that
        // match itself always qualifies: its source ends at or before its own start (a self-overlapping
        // match reads [src, dst) only and is extended periodically).  (The exact rule -- a match may start once
        // no unfinished match writes into its source range: a 64-bit mask per lane from two binary searches over
        // the lanes, tested against the ballot of pending lanes -- needs 3.3 rounds per batch instead of 5.6
        // (tools/token_stats.cpp) and was measured at the same kernel time: the rounds are not what the wave
        // waits for, the global loads of phase A are.)
        const uint32_t s_hi = src + (len < dist ? len : dist);
        const uint32_t dsto = dst - base, srco = src - base;
        bool pending = !(kAblate & 4u) && len != 0 && !far;
        // kExact (variant 3): the exact readiness rule.  A match may start once no UNFINISHED match writes into its source range
        // [src, s_hi).  Entries are in position order, so the matches whose destination meets that range are a contiguous range of
        // lanes [jlo, jhi] -- found once per batch by two binary searches over the lanes' {start, end} of the match (window offsets,
        // 16 bits each, one dword per lane in LDS) -- and "unfinished" is the ballot of pending lanes: ready = no pending lane in my
        // range.  3.3 rounds per batch instead of 5.6 on a BAM stream (tools/token_stats.cpp).  Measured (profiles/round5): scalar
        // instructions -22 %, vector instructions -4 % (a round is ~25 vector and ~55 scalar instructions; the two searches cost 36 vector
        // instructions per batch), kernel time -2.5 % (22.2 -> 21.7 ms on config 2): the kernel follows its VECTOR instruction count.
        uint32_t dep_lo = 0, dep_hi = 0;
        if (kExact) {
            uint32_t* rng = (uint32_t*)(smem + (kResThreads / 64) * kWaveLds + 128u) + wv * 64u;
            // (a lane without a match: start = end = where its match would be -- the arrays stay monotone, the lane is never pending)
            rng[lane] = dsto | ((dsto + len) << 16);
            uint32_t jlo = 0, jhi = 0;                       // lanes with end <= srco | lanes with start < s_hi - base
            const uint32_t s_hio = s_hi - base;
#pragma unroll
            for (uint32_t step = 32; step; step >>= 1) {
                const uint32_t a = rng[jlo + step - 1], b2 = rng[jhi + step - 1];
                jlo += (a >> 16) <= srco ? step : 0u;
                jhi += (b2 & 0xFFFFu) < s_hio ? step : 0u;
            }
            // (lane 63 is never counted by the search -- 63 steps at most -- and never matters: a range ends below its own lane)
            if (pending && jhi > jlo) {
                const uint64_t m = (jhi >= 64u ? ~0ull : (1ull << jhi) - 1ull) & ~((1ull << jlo) - 1ull);
                dep_lo = (uint32_t)m; dep_hi = (uint32_t)(m >> 32);
            }
        }
        for (uint64_t pm = __ballot(pending); pm; pm = __ballot(pending)) {
            bool ready;
            if (kExact) {
                ready = pending && ((dep_lo & (uint32_t)pm) | (dep_hi & (uint32_t)(pm >> 32))) == 0u;
            } else {
                const uint32_t F = __builtin_amdgcn_readlane(dst, __builtin_ctzll(pm));
                ready = pending && s_hi <= F;
            }
            pending = pending && !ready;
            const bool plain = ready && dist >= len;
            {
                constexpr uint32_t kOwn = kOwn32 ? 32u : 16u;
                const uint32_t own_s = plain && len <= kOwn ? len : 0u;       // (source and destination of a plain match do not overlap)
                const uint32_t steps = kOwn32 && __any(own_s > 16u) ? 2u : 1u;
                for (uint32_t h = 0; h < (kTail16 ? 1u : steps); ++h) {
                    const uint32_t n_s = own_s > 16u * h ? (own_s - 16u * h < 16u ? own_s - 16u * h : 16u) : 0u;
                    Short16 rs;
                    rs.load(buf + srco + 16u * h, n_s);
                    rs.store(buf + dsto + 16u * h, n_s);
                }
                if (kTail16 && steps > 1u && own_s > 16u) {
                    u32x4 ts;
                    __builtin_memcpy(&ts, buf + srco + own_s - 16u, 16);
                    __builtin_memcpy(buf + dsto + own_s - 16u, &ts, 16);
                }
                coop_copy(__ballot(plain && len > kOwn), buf, srco, buf, dsto, len, lane);
            }
            // self-overlapping matches: byte k is src[k mod dist].  Short ones in their own lane (all
            // loads first: the bytes read lie in [src, dst)), long ones by doubling: the period, then
            // 1, 2, 4 ... periods copied from the match's own output.
            const bool per = ready && dist < len;
            const bool per_perm = kOwn32 && per && len <= 16 && dist <= 8;
            if (kOwn32 && __any(per_perm)) {
                // output byte k = period[k mod dist]: the period sits in the first (<= 8) bytes at src, output dword j is one byte
                // permute of them; the tail words are cut out of neighbouring dwords
                Short16 ws;
                uint32_t n_p = 0;
                if (per_perm) {
                    const lz::W2 xx = lz::ld64(buf + srco);
                    const u32x4 sel = *(const u32x4*)(per_sel + 4u * (dist - 1u));
                    n_p = len;
                    lz::periodic16(xx.x, xx.y, len, sel.x, sel.y, sel.z, sel.w, &ws);
                }
                ws.store(buf + dsto, n_p);
            }
            if (per && len <= 16 && !per_perm) {
                uint32_t lo = 0, hi = 0, m = 0;      // 16 bytes in two 64-bit halves would need 4 regs; len <= 16
                uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
                for (uint32_t k = 0; k < 16; ++k) {
                    if (k < len) {
                        w[k >> 2] |= (uint32_t)buf[srco + m] << (8 * (k & 3));
                        if (++m == dist) m = 0;
                    }
                }
                (void)lo; (void)hi;
#pragma unroll
                for (uint32_t k = 0; k < 16; ++k)
                    if (k < len) buf[dsto + k] = (uint8_t)(w[k >> 2] >> (8 * (k & 3)));
            }
            for (uint64_t lm = __ballot(per && len > 16); lm; lm &= lm - 1) {
                const int q = __builtin_ctzll(lm);
                const uint32_t D = __builtin_amdgcn_readlane(dsto, q), P = __builtin_amdgcn_readlane(dist, q), N = __builtin_amdgcn_readlane(len, q);
                wave_copy(buf + D, buf + D - P, P, lane);
                for (uint32_t done = P; done < N;) {
                    const uint32_t n2 = done < N - done ? done : N - done;
                    wave_copy(buf + D + done, buf + D, n2, lane);
                    done += n2;
                }
            }
        }
        // ---- phase C: the finished span goes to HBM, contiguous ------------------------------------------------
        {
            const uint8_t* sp = buf + (opos - base);
            uint8_t* dp = o + opos;
            for (uint32_t i = 16 * lane; !(kAblate & 8u) && i < span; i += 1024) {
                if (i + 16 <= span) {
                    u32x4 v;
                    __builtin_memcpy(&v, sp + i, 16);
                    __builtin_memcpy(dp + i, &v, 16);
                } else {
                    for (uint32_t k = i; k < span; ++k) dp[k] = sp[k];
                }
            }
        }
        opos += span;
        lpos += lspan;
    }
}


#define SBX_LZ77_ARGS const uint8_t* __restrict__ lit_stream, const uint32_t* __restrict__ ent_stream, const uint32_t* __restrict__ n_entries, \
                      const uint64_t* __restrict__ out_off, const uint32_t* __restrict__ isize, uint32_t n_blocks, uint32_t block0, uint8_t* out, \
                      const uint32_t* __restrict__ status
#define SBX_LZ77_PASS lit_stream, ent_stream, n_entries, out_off, isize, n_blocks, block0, out, status
// k_lz77_resolve_exact: own-lane copies up to 32 bytes in two steps of two 8-byte words, byte-permute expansion of short periodic matches,
// the exact readiness rule (kExact above; LDS per wave: window + 256 bytes of match ranges).  Compiled for 8 waves per SIMD (64 VGPRs; the
// attribute is worth 0.4 ms).  (Round 3's k_lz77_resolve_o32 -- the frontier rule -- was the A/B partner until round 6; profiles/round5.)
template <uint32_t kHist, uint32_t kSpanMax>
__global__ __launch_bounds__(kResThreads) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_lz77_resolve_exact(SBX_LZ77_ARGS) {
    lz77_resolve_body<kHist, kSpanMax, true, true>(SBX_LZ77_PASS);
}

}  // namespace

size_t inflate_scratch_bytes(uint32_t n_blocks) { return (size_t)n_blocks * kScratchStride + 64; }

// sizes of the two token streams for n_blocks blocks producing `total` output bytes
size_t inflate_lit_bytes(uint64_t total, uint32_t n_blocks) { return (size_t)(lit_off(total, n_blocks) + 65536 + 64); }
size_t inflate_ent_words(uint64_t total, uint32_t n_blocks) { return (size_t)(ent_off(total, n_blocks) + 65536 / 3 + 65536 / 255 + 64); }   // (slack covers the padded last slice)

namespace {

struct InflateArgs {
    const uint8_t* comp; const uint64_t* comp_off; const uint32_t* comp_len; const uint32_t* isize; const uint64_t* out_off; uint8_t* out;
    uint32_t n_blocks, block0;
    uint8_t* scratch; uint8_t* lit; uint32_t* ent; uint32_t* nent; uint32_t* status; unsigned long long* tok;
    // the blocks [first, first + n) of these arrays
    InflateArgs slice(uint32_t first, uint32_t n) const {
        InflateArgs a = *this;
        a.comp_off += first; a.comp_len += first; a.isize += first; a.out_off += first; a.nent += first; a.status += first;
        a.scratch += (size_t)first * kScratchStride;
        a.block0 += first;
        a.n_blocks = n;
        return a;
    }
};

// K1a of a range of blocks: the fast kernel, the general one for the blocks that one flagged, the literal translation
void launch_k1a(const InflateArgs& a, hipStream_t stream) {
    if (a.n_blocks == 0) return;
    // SBX_K1A=1: round 3's kernel alone (the general kernel: every kind of block); default: the fast kernel, then the general one for
    // the blocks the fast one flagged (a wavefront without a flagged block ends at once), then the literal translation
    static const int k1a = [] { const char* e = getenv("SBX_K1A"); return e ? atoi(e) : 2; }();
    if (k1a != 1) {
        dim3 grid((a.n_blocks + kInf2Threads - 1) / kInf2Threads), block(kInf2Threads);
        // (SBX_K1A_LDS_PAD: extra LDS per workgroup -- an occupancy experiment)
        static const size_t pad = [] { const char* e = getenv("SBX_K1A_LDS_PAD"); return e ? (size_t)atoi(e) : (size_t)0; }();
        const size_t lds = (size_t)inf2::kWaveLds + inf2::kLenTabBytes + inf2::kDistTabBytes + pad;
        hipLaunchKernelGGL(k_huffman_decode2, grid, block, lds, stream, a.comp, a.comp_off, a.comp_len, a.isize, a.out_off, a.n_blocks, a.block0,
                           a.lit, a.ent, a.nent, a.scratch, a.status, a.tok);
        SBX_HIP(hipGetLastError());
    }
    {
        dim3 grid((a.n_blocks + kInfThreads - 1) / kInfThreads), block(kInfThreads);
        const size_t lds = (size_t)kInfThreads * kLaneLds + kLenTabBytes + kDistTabBytes;
        const uint32_t only_flagged = k1a != 1 ? 1u : 0u;
        // (one 16-byte group per token store: the kernel is a template over the burst depth and the cache policy of its stores --
        // rounds 3 / 4 measured 2 and 4 groups per burst and nontemporal stores, profiles/round3, profiles/round4; only this form is built)
        hipLaunchKernelGGL((k_huffman_decode<1, false>), grid, block, lds, stream, a.comp, a.comp_off, a.comp_len, a.isize, a.out_off,
                           a.n_blocks, a.block0, a.lit, a.ent, a.nent, a.scratch, a.status, a.tok, only_flagged);
        SBX_HIP(hipGetLastError());
    }
    if (k1a != 1) {
        hipLaunchKernelGGL(k_translate_literals, dim3(a.n_blocks), dim3(64), 0, stream, a.lit, a.out_off, a.n_blocks, a.block0, a.scratch, a.status);
        SBX_HIP(hipGetLastError());
    }
}

void launch_k1b(const InflateArgs& a, hipStream_t stream) {
    if (a.n_blocks == 0) return;
    const uint32_t per = kResThreads / 64;
    dim3 grid((a.n_blocks + per - 1) / per), block(kResThreads);
    const size_t lds = (size_t)(kResThreads / 64) * (kHistDefault + 1024u + kSpanDefault + 16u);
    hipLaunchKernelGGL((k_lz77_resolve_exact<kHistDefault, kSpanDefault>), grid, block, lds + 128 + (size_t)(kResThreads / 64) * 256, stream, a.lit,
                       a.ent, a.nent, a.out_off, a.isize, a.n_blocks, a.block0, a.out, a.status);
    SBX_HIP(hipGetLastError());
}

}  // namespace

void launch_bgzf_inflate(const uint8_t* d_comp, const uint64_t* d_comp_off, const uint32_t* d_comp_len,
                         const uint32_t* d_isize, const uint64_t* d_out_off, uint8_t* d_out, uint32_t n_blocks,
                         uint32_t block0, uint8_t* d_scratch, uint8_t* d_lit, uint32_t* d_ent, uint32_t* d_nent,
                         uint32_t* d_status, hipStream_t stream, hipEvent_t ev_mid, unsigned long long* d_tok_bytes) {
    if (n_blocks == 0) { if (ev_mid) SBX_HIP(hipEventRecord(ev_mid, stream)); return; }
    const InflateArgs all{d_comp, d_comp_off, d_comp_len, d_isize, d_out_off, d_out, n_blocks, block0, d_scratch, d_lit, d_ent, d_nent, d_status, d_tok_bytes};
    // (measured and dropped, profiles/round4/README.md: K1a of the second half of the blocks on a second stream next to K1b of the first half --
    // the two kernels do not fill each other's issue slots, the inflate takes 51-54 ms instead of 48)
    launch_k1a(all, stream);
    if (ev_mid) SBX_HIP(hipEventRecord(ev_mid, stream));
    launch_k1b(all, stream);
}

const char* inflate_status_string(uint32_t s) {
    switch (s) {
        case INF_OK: return "ok";
        case INF_BAD_BTYPE: return "invalid block type";
        case INF_BAD_STORED: return "invalid stored block lengths";
        case INF_BAD_CODELENS: return "invalid code lengths set";
        case INF_BAD_SYMBOL: return "invalid literal/length code";
        case INF_BAD_DISTANCE: return "invalid distance";
        case INF_OUTPUT_OVERRUN: return "output exceeds ISIZE";
        case INF_INPUT_OVERRUN: return "deflate stream runs past the end of the block";
        case INF_SIZE_MISMATCH: return "inflated size differs from ISIZE";
        default: return "unknown";
    }
}

}  // namespace sbx
