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
//  K1a `huffman_decode`  -- entropy decoding is a serial dependency chain per stream, so the
//      mapping is ONE LANE PER BGZF BLOCK: 64 independent decoders per wavefront, every VALU
//      instruction doing useful work in all lanes.  No lookup tables in memory for the code
//      lengths: the 15 left-justified limits of the literal/length and the distance code live in
//      VGPRs and the code length is 1 + sum_l (peek >= limit[l]) -- 14 compares, branch-free and
//      identical for every lane.  The per-lane symbol permutations (288 + 32 entries) sit in LDS
//      at a 105-dword lane stride (odd => conflict-free for equal offsets).  The decoder does NOT
//      touch the LZ77 window: it emits the literal bytes (packed 4 per store) and one 32-bit
//      entry {literals-before:8, distance-1:15, length:9} per match.  Nothing it loads depends on
//      anything it stored, so the lane never waits on the LZ77 window's memory latency.
//
//  K1b `lz77_resolve`    -- copying matches is data-parallel once positions are known, so the
//      mapping is ONE WAVE PER BGZF BLOCK: 64 entries at a time, output offsets by a wave prefix
//      sum, literals and match bytes copied by 8 lanes per entry (coalesced within an entry),
//      periodic extension (src + k mod dist) removes the intra-match dependency, and matches whose
//      source overlaps a still-pending match of the same batch wait for the next round (bitmask
//      test; the first pending entry is always ready).  No LDS window: the sliding window is the
//      wave's own earlier output in HBM/L2, and 32 waves per CU hide its latency.
//
// Roofline: K1a is bound by the serial decode chain (VALU + LDS latency), K1b by memory latency of
// scattered short copies; neither is HBM-bandwidth bound and their GB/s are reported separately
// from the HBM-bound accumulate kernel.
#include "common.hpp"
#include "kernels.hpp"

namespace sbx {

namespace {

constexpr int kInfThreads = 64;              // K1a: one wavefront per workgroup, one lane per BGZF block
constexpr int kLaneLds = 420;                // bytes of LDS per lane (105 dwords: odd stride)
constexpr int kLitSymOff = 0;                // u8[288]  low 8 bits of literal/length symbols, canonical order
constexpr int kLitHiOff = 288;               // u8[36]   bit 8 of those symbols, bit-packed
constexpr int kDistSymOff = 324;             // u8[32]   distance symbols, canonical order
constexpr int kLitDeltaOff = 356;            // i16[16]  symbol-index delta per code length (lit/len)
constexpr int kDistDeltaOff = 388;           // i16[16]  same for distances
constexpr int kLensScratch = 320;            // bytes of global scratch per lane: code lengths being built

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

// ---- token streams between K1a and K1b -----------------------------------------------------------
// literal stream of block b: bytes at lit[lit_off(b) ..], 16-byte aligned, capacity isize[b] (+pad)
// entry stream of block b  : u32 at ent[ent_off(b) ..], capacity isize/3 + isize/255 + 4
// Both offsets are pure functions of (out_off[b], b) so that no extra table is needed.
__host__ __device__ __forceinline__ uint64_t lit_off(uint64_t out_off_b, uint32_t b) { return (out_off_b + 16ull * b + 15ull) & ~15ull; }
__host__ __device__ __forceinline__ uint64_t ent_off(uint64_t out_off_b, uint32_t b) { return out_off_b / 3 + out_off_b / 255 + 8ull * b; }

__device__ __forceinline__ uint32_t make_entry(uint32_t lit_run, uint32_t len, uint32_t dist) {
    return (lit_run << 24) | ((dist - 1) << 9) | len;      // len == 0: literal-run-only entry
}

struct BitReader {
    const uint8_t* cp;    // next 16-byte chunk to prefetch (4-byte aligned)
    u32x4 cur, nxt;       // words being consumed / prefetched chunk
    int widx;             // next word of `cur` (0..3)
    uint64_t buf;
    int cnt;              // valid bits in buf
    uint32_t consumed;    // bits consumed so far (relative to payload start)

    __device__ __forceinline__ static u32x4 load16(const uint8_t* p) {
        u32x4 v;
        __builtin_memcpy(&v, p, 16);
        return v;
    }
    __device__ __forceinline__ uint32_t next_word() {
        uint32_t w = widx == 0 ? cur.x : widx == 1 ? cur.y : widx == 2 ? cur.z : cur.w;
        if (++widx == 4) {
            cur = nxt;
            widx = 0;
            nxt = load16(cp);
            cp += 16;
        }
        return w;
    }
    __device__ __forceinline__ void init(const uint8_t* p) {
        int lead = (int)((uintptr_t)p & 3);
        const uint8_t* a = p - lead;
        cur = load16(a);
        nxt = load16(a + 16);
        cp = a + 32;
        widx = 0;
        uint32_t w0 = next_word();
        buf = (uint64_t)(w0 >> (8 * lead));
        cnt = 32 - 8 * lead;
        consumed = 0;
    }
    __device__ __forceinline__ void refill() {   // guarantees cnt > 32 afterwards
        if (cnt <= 32) {
            buf |= (uint64_t)next_word() << cnt;
            cnt += 32;
        }
    }
    __device__ __forceinline__ uint32_t peek(int n) const { return (uint32_t)buf & ((1u << n) - 1u); }
    __device__ __forceinline__ void drop(int n) { buf >>= n; cnt -= n; consumed += (uint32_t)n; }
    __device__ __forceinline__ uint32_t take(int n) { uint32_t v = peek(n); drop(n); return v; }
};

// Per-code decode state held in registers: left-justified (15-bit) exclusive upper limits.
struct Limits {
    uint32_t lim[16];   // lim[l], l = 1..15 ; lim[0] unused
};

// length of the canonical code whose left-justified 15-bit prefix is v (1..16; 16 = invalid)
__device__ __forceinline__ int code_length(const Limits& L, uint32_t v) {
    int len = 1;
#pragma unroll
    for (int l = 1; l <= 15; ++l) len += (v >= L.lim[l]) ? 1 : 0;
    return len;
}

// Build one canonical code from `n` code lengths in lens[] (global scratch, 1 byte each): symbol
// permutation + per-length deltas to LDS, limits to L.  Returns false on an over-subscribed code.
// (Incomplete codes are accepted, as zlib accepts the single-code distance tree; an unused code
// decodes as "invalid symbol".)
template <bool kIsLit>
__device__ __forceinline__ bool build_code(const uint8_t* lens, int n, uint8_t* lds, Limits& L) {
    uint16_t* tmp = (uint16_t*)(lds + (kIsLit ? kLitDeltaOff : kDistDeltaOff));
#pragma unroll
    for (int l = 0; l < 16; ++l) tmp[l] = 0;
    for (int s = 0; s < n; ++s) tmp[lens[s]] += 1;
    uint32_t cnt[16];
#pragma unroll
    for (int l = 0; l < 16; ++l) cnt[l] = tmp[l];
    uint32_t first = 0, offs = 0;
    int32_t left = 1;
    bool ok = true;
    uint32_t firstc[16];
#pragma unroll
    for (int l = 1; l <= 15; ++l) {
        left = (left << 1) - (int32_t)cnt[l];
        if (left < 0) ok = false;
        firstc[l] = first;
        L.lim[l] = (first + cnt[l]) << (15 - l);
        tmp[l] = (uint16_t)offs;            // running insert position during the sort below
        offs += cnt[l];
        first = (first + cnt[l]) << 1;
    }
    L.lim[0] = 0;
    if (!ok) return false;
    if (kIsLit) {
#pragma unroll
        for (int i = 0; i < 9; ++i) ((uint32_t*)(lds + kLitHiOff))[i] = 0;
    }
    for (int s = 0; s < n; ++s) {
        uint32_t l = lens[s];
        if (l) {
            uint32_t idx = tmp[l];
            tmp[l] = (uint16_t)(idx + 1);
            if (kIsLit) {
                lds[kLitSymOff + idx] = (uint8_t)s;
                if (s & 256) lds[kLitHiOff + (idx >> 3)] |= (uint8_t)(1u << (idx & 7));
            } else {
                lds[kDistSymOff + idx] = (uint8_t)s;
            }
        }
    }
    // delta[l] = (start index of length-l symbols) - (first code of length l)
#pragma unroll
    for (int l = 1; l <= 15; ++l) {
        uint32_t start = (uint32_t)tmp[l] - cnt[l];
        tmp[l] = (uint16_t)(start - firstc[l]);
    }
    return true;
}

// Token emitter of one lane (K1a): literal bytes packed four per store, one u32 per match.
struct Emitter {
    uint8_t* lit;         // 16-byte aligned literal stream of this block
    uint32_t* ent;
    uint32_t n_lit, n_ent, acc, run;
    __device__ __forceinline__ void init(uint8_t* l, uint32_t* e) { lit = l; ent = e; n_lit = n_ent = acc = run = 0; }
    __device__ __forceinline__ void literal(uint32_t byte) {
        acc |= byte << (8u * (n_lit & 3u));
        ++n_lit;
        if ((n_lit & 3u) == 0) { *(uint32_t*)(lit + n_lit - 4) = acc; acc = 0; }
        if (++run == 255) { ent[n_ent++] = make_entry(255, 0, 1); run = 0; }
    }
    __device__ __forceinline__ void match(uint32_t len, uint32_t dist) {
        ent[n_ent++] = make_entry(run, len, dist);
        run = 0;
    }
    __device__ __forceinline__ void finish() {
        if (n_lit & 3u) *(uint32_t*)(lit + (n_lit & ~3u)) = acc;
        if (run) { ent[n_ent++] = make_entry(run, 0, 1); run = 0; }
    }
};

__global__ __launch_bounds__(kInfThreads) void k_huffman_decode(
    const uint8_t* __restrict__ comp, const uint64_t* __restrict__ comp_off, const uint32_t* __restrict__ comp_len,
    const uint32_t* __restrict__ isize, const uint64_t* __restrict__ out_off, uint32_t n_blocks, uint32_t block0,
    uint8_t* __restrict__ lit_stream, uint32_t* __restrict__ ent_stream, uint32_t* __restrict__ n_entries,
    uint8_t* __restrict__ lens_scratch, uint32_t* __restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t b = blockIdx.x * kInfThreads + threadIdx.x;
    if (b >= n_blocks) return;
    uint8_t* lds = smem + threadIdx.x * kLaneLds;
    uint8_t* lens = lens_scratch + (size_t)b * kLensScratch;

    const uint8_t* in = comp + comp_off[b];
    const uint32_t in_bits = comp_len[b] * 8u;
    const uint32_t osize = isize[b];
    const uint64_t oo = out_off[b];
    uint32_t opos = 0;
    uint32_t err = INF_OK;

    Emitter em;
    em.init(lit_stream + lit_off(oo, block0 + b), ent_stream + ent_off(oo, block0 + b));
    BitReader br;
    br.init(in);
    Limits LL, LD;
#pragma unroll
    for (int l = 0; l < 16; ++l) { LL.lim[l] = 0; LD.lim[l] = 0; }

    bool last = (osize == 0 && comp_len[b] == 0);   // nothing to do for an empty payload
    while (!last && err == INF_OK) {
        br.refill();
        last = br.take(1) != 0;
        uint32_t btype = br.take(2);
        if (btype == 0) {
            // stored block: skip to the byte boundary, LEN, NLEN, raw bytes
            br.drop(br.cnt & 7);
            br.refill();
            uint32_t len = br.take(16);
            br.refill();
            uint32_t nlen = br.take(16);
            if ((len ^ 0xFFFFu) != nlen) { err = INF_BAD_STORED; break; }
            if (opos + len > osize) { err = INF_OUTPUT_OVERRUN; break; }
            for (uint32_t i = 0; i < len; ++i) {
                br.refill();
                em.literal(br.take(8));
            }
            opos += len;
            continue;
        }
        if (btype == 3) { err = INF_BAD_BTYPE; break; }
        int nlit, ndist;
        if (btype == 1) {
            // fixed code (RFC 1951 3.2.6)
            for (int s = 0; s < 288; ++s) lens[s] = (uint8_t)(s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8);
            for (int s = 0; s < 30; ++s) lens[288 + s] = 5;
            nlit = 288;
            ndist = 30;
        } else {
            br.refill();
            nlit = (int)br.take(5) + 257;
            ndist = (int)br.take(5) + 1;
            int ncl = (int)br.take(4) + 4;
            if (nlit > 286 || ndist > 30) { err = INF_BAD_CODELENS; break; }
            // code-length code: 19 symbols, 3-bit lengths, in the RFC's permuted order.  Its
            // lengths (19 bytes) and canonical symbol order (19 bytes) are parked in the lane's
            // LDS symbol areas, which are rebuilt right after the header anyway.
            uint8_t* cl_len = lds + kDistSymOff;   // [19]
            uint8_t* cl_sym = lds + kLitSymOff;    // [19] symbols sorted by (length, value)
            for (int s = 0; s < 19; ++s) cl_len[s] = 0;
            for (int i = 0; i < ncl; ++i) {
                br.refill();
                cl_len[kClOrder[i]] = (uint8_t)br.take(3);
            }
            uint32_t ccnt[8];
#pragma unroll
            for (int l = 0; l < 8; ++l) ccnt[l] = 0;
            {
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
                if (!ok) { err = INF_BAD_CODELENS; break; }
            }
            // decode nlit + ndist code lengths
            int i = 0;
            uint32_t prev = 0;
            const int total = nlit + ndist;
            while (i < total && err == INF_OK) {
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
                if (sym >= 19) { err = INF_BAD_CODELENS; break; }
                br.drop(used);
                if (sym < 16) {
                    lens[i++] = (uint8_t)sym;
                    prev = sym;
                } else {
                    uint32_t rep, val;
                    if (sym == 16) {
                        if (i == 0) { err = INF_BAD_CODELENS; break; }
                        val = prev;
                        rep = 3 + br.take(2);
                    } else if (sym == 17) {
                        val = 0;
                        rep = 3 + br.take(3);
                    } else {
                        val = 0;
                        rep = 11 + br.take(7);
                    }
                    if (i + (int)rep > total) { err = INF_BAD_CODELENS; break; }
                    for (uint32_t k = 0; k < rep; ++k) lens[i++] = (uint8_t)val;
                    prev = val;   // (after 17/18 a following 16 repeats 0, as in zlib)
                }
            }
            if (err != INF_OK) break;
            if (lens[256] == 0) { err = INF_BAD_CODELENS; break; }   // no end-of-block code
        }
        if (!build_code<true>(lens, nlit, lds, LL)) { err = INF_BAD_CODELENS; break; }
        if (!build_code<false>(lens + nlit, ndist, lds, LD)) { err = INF_BAD_CODELENS; break; }
        const int16_t* ldelta = (const int16_t*)(lds + kLitDeltaOff);
        const int16_t* ddelta = (const int16_t*)(lds + kDistDeltaOff);

        // ---- symbol loop -----------------------------------------------------------------
        for (;;) {
            br.refill();
            uint32_t v = __brev(br.peek(15)) >> 17;
            int len = code_length(LL, v);
            if (len > 15) { err = INF_BAD_SYMBOL; break; }
            uint32_t idx = (uint32_t)((int32_t)ldelta[len] + (int32_t)(v >> (15 - len))) & 0x1FFu;
            if (idx >= 288) { err = INF_BAD_SYMBOL; break; }
            uint32_t sym = (uint32_t)lds[kLitSymOff + idx] | (((uint32_t)lds[kLitHiOff + (idx >> 3)] >> (idx & 7)) & 1u) << 8;
            br.drop(len);
            if (sym < 256) {
                if (opos >= osize) { err = INF_OUTPUT_OVERRUN; break; }
                ++opos;
                em.literal(sym);
                continue;
            }
            if (sym == 256) break;
            if (sym > 285) { err = INF_BAD_SYMBOL; break; }
            // match length (RFC 1951 3.2.5), computed arithmetically
            uint32_t mlen;
            if (sym < 265) mlen = sym - 254;
            else if (sym == 285) mlen = 258;
            else {
                uint32_t e = (sym - 261) >> 2;
                mlen = ((4 + ((sym - 261) & 3)) << e) + 3 + br.take((int)e);
            }
            br.refill();
            uint32_t dv = __brev(br.peek(15)) >> 17;
            int dl = code_length(LD, dv);
            if (dl > 15) { err = INF_BAD_DISTANCE; break; }
            uint32_t didx = (uint32_t)((int32_t)ddelta[dl] + (int32_t)(dv >> (15 - dl))) & 0x1FFu;
            if (didx >= 30) { err = INF_BAD_DISTANCE; break; }
            uint32_t dsym = lds[kDistSymOff + didx];
            br.drop(dl);
            if (dsym > 29) { err = INF_BAD_DISTANCE; break; }
            uint32_t dist;
            if (dsym < 4) dist = dsym + 1;
            else {
                uint32_t e = (dsym >> 1) - 1;
                dist = ((2 + (dsym & 1)) << e) + 1 + br.take((int)e);
            }
            if (dist > opos) { err = INF_BAD_DISTANCE; break; }
            if (opos + mlen > osize) { err = INF_OUTPUT_OVERRUN; break; }
            opos += mlen;
            em.match(mlen, dist);
        }
    }
    em.finish();
    if (err == INF_OK && opos != osize) err = INF_SIZE_MISMATCH;
    if (err == INF_OK && br.consumed > in_bits) err = INF_INPUT_OVERRUN;
    status[b] = err;
    n_entries[b] = em.n_ent;
}

// ---- K1b -----------------------------------------------------------------------------------------
constexpr int kResThreads = 256;   // 4 waves per workgroup, one BGZF block per wave

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, uint32_t lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(v, d, 64);
        if ((int)lane >= d) v += t;
    }
    return v;
}

__global__ __launch_bounds__(kResThreads) void k_lz77_resolve(
    const uint8_t* __restrict__ lit_stream, const uint32_t* __restrict__ ent_stream, const uint32_t* __restrict__ n_entries,
    const uint64_t* __restrict__ out_off, const uint32_t* __restrict__ isize, uint32_t n_blocks, uint32_t block0,
    uint8_t* out, const uint32_t* __restrict__ status) {
    __shared__ uint32_t s_end[kResThreads / 64][64];   // per wave: end offset (exclusive) of every entry's output
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t b = blockIdx.x * (kResThreads / 64) + wv;
    if (b >= n_blocks) return;
    if (status[b] != INF_OK) return;
    const uint64_t oo = out_off[b];
    const uint8_t* lit = lit_stream + lit_off(oo, block0 + b);
    const uint32_t* ent = ent_stream + ent_off(oo, block0 + b);
    uint8_t* o = out + oo;
    const uint32_t ne = n_entries[b];
    uint32_t opos = 0, lpos = 0;   // wave-uniform running positions
    uint32_t* end_arr = s_end[wv];
    const uint32_t grp = lane >> 3, k8 = lane & 7u;

    for (uint32_t e0 = 0; e0 < ne; e0 += 64) {
        const uint32_t e = (e0 + lane < ne) ? ent[e0 + lane] : 0u;
        const uint32_t lr = e >> 24, len = e & 511u, dist = ((e >> 9) & 0x7FFFu) + 1u;
        const uint32_t tot = lr + len;
        const uint32_t incl = wave_incl_scan(tot, lane);
        const uint32_t lincl = wave_incl_scan(lr, lane);
        const uint32_t eo = opos + incl - tot;        // first output byte of this entry (its literal run)
        const uint32_t el = lpos + lincl - lr;        // first literal of this entry in the literal stream
        const uint32_t dst = eo + lr;                 // first byte of the match
        const uint32_t src = dst - dist;
        end_arr[lane] = eo + tot;
        // ---- literal runs: 8 lanes per entry, no dependencies (source = literal stream) ----
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const int x = g * 8 + (int)grp;
            const uint32_t lr_x = __shfl(lr, x, 64), eo_x = __shfl(eo, x, 64), el_x = __shfl(el, x, 64);
            for (uint32_t kk = k8; kk < lr_x; kk += 8) o[eo_x + kk] = lit[el_x + kk];
        }
        // ---- which earlier entries of this batch does my source range touch? -------------
        // entry i occupies [end[i-1], end[i]); ends are non-decreasing.  lo = lowest i with
        // end[i] > src ; hi = 1 + lowest i with end[i] >= src + min(len, dist)  (the source bytes
        // actually read are [src, src + min(len, dist)) thanks to the periodic extension below).
        uint64_t dep = 0;
        const uint32_t s_hi = src + (len < dist ? len : dist);
        if (len && s_hi > opos) {                     // (sources entirely before this batch are final)
            uint32_t lo, hi;
            {
                uint32_t a = 0, c = lane;
                while (a < c) { uint32_t m = (a + c) >> 1; if (end_arr[m] > src) c = m; else a = m + 1; }
                lo = a;
            }
            {
                uint32_t a = lo, c = lane;
                while (a < c) { uint32_t m = (a + c) >> 1; if (end_arr[m] >= s_hi) c = m; else a = m + 1; }
                hi = a < lane ? a + 1 : lane;         // exclusive
            }
            if (hi > lo) dep = (hi - lo >= 64 ? ~0ULL : ((1ULL << (hi - lo)) - 1ULL)) << lo;
        }
        // ---- match rounds ---------------------------------------------------------------
        uint64_t pending = __ballot(len != 0);
        while (pending) {
            const bool ready = len != 0 && ((pending >> lane) & 1ULL) && (dep & pending) == 0ULL;
            const uint64_t rmask = __ballot(ready);
            // 8 lanes per entry; byte k of a match is out[src + k mod dist] (periodic extension:
            // every byte of an entry only depends on bytes before the entry)
            uint32_t v0[8], v1[8];
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const int x = g * 8 + (int)grp;
                const uint32_t len_x = __shfl(len, x, 64), src_x = __shfl(src, x, 64), dist_x = __shfl(dist, x, 64);
                const bool on = (rmask >> x) & 1ULL;
                v0[g] = 0; v1[g] = 0;
                if (on && k8 < len_x) v0[g] = o[src_x + (k8 < dist_x ? k8 : k8 % dist_x)];
                if (on && k8 + 8 < len_x) v1[g] = o[src_x + (k8 + 8 < dist_x ? k8 + 8 : (k8 + 8) % dist_x)];
            }
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const int x = g * 8 + (int)grp;
                const uint32_t len_x = __shfl(len, x, 64), dst_x = __shfl(dst, x, 64);
                const bool on = (rmask >> x) & 1ULL;
                if (on && k8 < len_x) o[dst_x + k8] = (uint8_t)v0[g];
                if (on && k8 + 8 < len_x) o[dst_x + k8 + 8] = (uint8_t)v1[g];
            }
            // tails of long matches (> 16 bytes)
            uint64_t lm = __ballot(ready && len > 16);
            while (lm) {
                const int x = __builtin_ctzll(lm);
                lm &= lm - 1;
                const uint32_t len_x = __shfl(len, x, 64), src_x = __shfl(src, x, 64), dist_x = __shfl(dist, x, 64),
                               dst_x = __shfl(dst, x, 64);
                for (uint32_t kk = 16 + lane; kk < len_x; kk += 64) o[dst_x + kk] = o[src_x + (kk < dist_x ? kk : kk % dist_x)];
            }
            pending &= ~rmask;
        }
        opos += __shfl(incl, 63, 64);
        lpos += __shfl(lincl, 63, 64);
    }
}

}  // namespace

size_t inflate_scratch_bytes(uint32_t n_blocks) { return (size_t)n_blocks * kLensScratch; }

// sizes of the two token streams for n_blocks blocks producing `total` output bytes
size_t inflate_lit_bytes(uint64_t total, uint32_t n_blocks) { return (size_t)(lit_off(total, n_blocks) + 65536 + 64); }
size_t inflate_ent_words(uint64_t total, uint32_t n_blocks) { return (size_t)(ent_off(total, n_blocks) + 65536 / 3 + 65536 / 255 + 64); }

void launch_bgzf_inflate(const uint8_t* d_comp, const uint64_t* d_comp_off, const uint32_t* d_comp_len,
                         const uint32_t* d_isize, const uint64_t* d_out_off, uint8_t* d_out, uint32_t n_blocks,
                         uint32_t block0, uint8_t* d_scratch, uint8_t* d_lit, uint32_t* d_ent, uint32_t* d_nent,
                         uint32_t* d_status, hipStream_t stream, hipEvent_t ev_mid) {
    if (n_blocks == 0) { if (ev_mid) SBX_HIP(hipEventRecord(ev_mid, stream)); return; }
    {
        dim3 grid((n_blocks + kInfThreads - 1) / kInfThreads), block(kInfThreads);
        size_t lds = (size_t)kInfThreads * kLaneLds;
        hipLaunchKernelGGL(k_huffman_decode, grid, block, lds, stream, d_comp, d_comp_off, d_comp_len, d_isize, d_out_off,
                           n_blocks, block0, d_lit, d_ent, d_nent, d_scratch, d_status);
        SBX_HIP(hipGetLastError());
    }
    if (ev_mid) SBX_HIP(hipEventRecord(ev_mid, stream));
    {
        const uint32_t per = kResThreads / 64;
        dim3 grid((n_blocks + per - 1) / per), block(kResThreads);
        hipLaunchKernelGGL(k_lz77_resolve, grid, block, 0, stream, d_lit, d_ent, d_nent, d_out_off, d_isize, n_blocks, block0,
                           d_out, d_status);
        SBX_HIP(hipGetLastError());
    }
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
