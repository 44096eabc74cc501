// inflate.hip -- K1 `bgzf_inflate`: raw-DEFLATE (RFC 1951) decode of BGZF payloads on gfx950.
//
// Replaces decompressBgzfBlock (BioD/bio/core/bgzf/block.d:127-216: zlib inflateInit2(-15) +
// inflate(Z_FINISH), one call per <=64 KiB block, run as std.parallelism tasks from
// BgzfInputStream.fillNextBlock, inputstream.d:414-417).  As in the reference's release build the
// CRC32 trailer is not verified (block.d:187).
//
// Design (MI355X-first, not a zlib translation): a BAM is hundreds of thousands of *independent*
// small deflate streams, so the unit of parallelism is the BGZF block and the mapping is one
// lane per block -- 64 independent decoders per wavefront, ~100 k blocks in flight per chip.
// Huffman decoding is a serial dependency chain per stream; lane-per-stream keeps all 64 lanes
// of every VALU instruction busy where a wave-per-stream decoder would use one.
//  * Canonical-code decode without lookup tables in memory: the 15 left-justified code-length
//    limits of the literal/length and distance codes live in VGPRs; the code length is
//    1 + sum_l (peek >= limit[l]) -- 14 compares, branch-free, identical for every lane.
//  * Per-lane symbol permutation tables (288 + 32 entries) and 2 x 16 per-length deltas sit in
//    LDS at a 105-dword lane stride (odd => all lanes hit distinct banks for equal offsets).
//  * Input is read as aligned dwords, one word prefetched ahead of use; output literals and
//    matches are written straight to the lane's slice of the inflated stream in HBM (the LZ77
//    window is the lane's own earlier output, served from L2/MALL).
// Roofline: this kernel is bound by the serial decode chain (ALU + LDS latency), not by HBM;
// its achieved GB/s is reported separately from the HBM-bound accumulate kernel.
#include "common.hpp"
#include "kernels.hpp"

namespace sbx {

namespace {

constexpr int kInfThreads = 64;              // one wavefront per workgroup
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

// unaligned dword access (byte-aligned pointers: never cast to uint32_t*)
__device__ __forceinline__ uint32_t ldu32(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
__device__ __forceinline__ void stu32(uint8_t* p, uint32_t v) { __builtin_memcpy(p, &v, 4); }

struct BitReader {
    const uint32_t* wp;   // next aligned word to fetch
    uint32_t nxt;         // prefetched word *wp[-1+1]
    uint64_t buf;
    int cnt;              // valid bits in buf
    int64_t consumed;     // bits consumed so far (relative to payload start)

    __device__ __forceinline__ void init(const uint8_t* p) {
        uintptr_t a = (uintptr_t)p;
        const uint32_t* w = (const uint32_t*)(a & ~(uintptr_t)3);
        int lead = (int)(a & 3);
        uint32_t w0 = w[0];
        buf = (uint64_t)(w0 >> (8 * lead));
        cnt = 32 - 8 * lead;
        nxt = w[1];
        wp = w + 2;
        consumed = 0;
    }
    __device__ __forceinline__ void refill() {   // guarantees cnt > 32 afterwards
        if (cnt <= 32) {
            buf |= (uint64_t)nxt << cnt;
            cnt += 32;
            nxt = *wp++;
        }
    }
    __device__ __forceinline__ uint32_t peek(int n) const { return (uint32_t)buf & ((1u << n) - 1u); }
    __device__ __forceinline__ void drop(int n) { buf >>= n; cnt -= n; consumed += n; }
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

// Build one canonical code from `n` code lengths in lens[] (global scratch, 1 byte each).
// Writes the symbol permutation through put_sym(index, symbol), deltas to LDS, limits to L.
// Returns false on an over-subscribed code.  (Incomplete codes are accepted as zlib does for
// the single-code distance tree; an unused code simply decodes as "invalid symbol".)
template <bool kIsLit>
__device__ __forceinline__ bool build_code(const uint8_t* lens, int n, uint8_t* lds, Limits& L) {
    uint16_t* tmp = (uint16_t*)(lds + (kIsLit ? kLitDeltaOff : kDistDeltaOff));
#pragma unroll
    for (int l = 0; l < 16; ++l) tmp[l] = 0;
    for (int s = 0; s < n; ++s) tmp[lens[s]] += 1;
    uint32_t cnt[16];
#pragma unroll
    for (int l = 0; l < 16; ++l) cnt[l] = tmp[l];
    // limits / first codes / offsets
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

__global__ __launch_bounds__(kInfThreads) void k_bgzf_inflate(
    const uint8_t* __restrict__ comp, const uint64_t* __restrict__ comp_off, const uint32_t* __restrict__ comp_len,
    const uint32_t* __restrict__ isize, const uint64_t* __restrict__ out_off, uint8_t* __restrict__ out,
    uint32_t n_blocks, uint8_t* __restrict__ lens_scratch, uint32_t* __restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t b = blockIdx.x * kInfThreads + threadIdx.x;
    if (b >= n_blocks) return;
    uint8_t* lds = smem + threadIdx.x * kLaneLds;
    uint8_t* lens = lens_scratch + (size_t)b * kLensScratch;

    const uint8_t* in = comp + comp_off[b];
    const int64_t in_bits = (int64_t)comp_len[b] * 8;
    uint8_t* const obase = out + out_off[b];
    const uint32_t osize = isize[b];
    uint32_t opos = 0;
    uint32_t err = INF_OK;

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
                obase[opos++] = (uint8_t)br.take(8);
            }
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
                    prev = val;   // (for 17/18 prev becomes 0, as in zlib: a following 16 repeats 0)
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
                obase[opos++] = (uint8_t)sym;
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
            uint8_t* dst = obase + opos;
            const uint8_t* src = dst - dist;
            opos += mlen;
            if (dist >= mlen) {
                // no overlap: unaligned dword moves, loads issued ahead of the stores
                uint32_t i = 0;
                for (; i + 16 <= mlen; i += 16) {
                    uint32_t a0 = ldu32(src + i), a1 = ldu32(src + i + 4), a2 = ldu32(src + i + 8), a3 = ldu32(src + i + 12);
                    stu32(dst + i, a0); stu32(dst + i + 4, a1); stu32(dst + i + 8, a2); stu32(dst + i + 12, a3);
                }
                for (; i + 4 <= mlen; i += 4) stu32(dst + i, ldu32(src + i));
                for (; i < mlen; ++i) dst[i] = src[i];
            } else {
                for (uint32_t i = 0; i < mlen; ++i) dst[i] = src[i];
            }
        }
    }
    if (err == INF_OK && opos != osize) err = INF_SIZE_MISMATCH;
    if (err == INF_OK && br.consumed > in_bits) err = INF_INPUT_OVERRUN;
    status[b] = err;
}

}  // namespace

size_t inflate_scratch_bytes(uint32_t n_blocks) { return (size_t)n_blocks * kLensScratch; }

void launch_bgzf_inflate(const uint8_t* d_comp, const uint64_t* d_comp_off, const uint32_t* d_comp_len,
                         const uint32_t* d_isize, const uint64_t* d_out_off, uint8_t* d_out, uint32_t n_blocks,
                         uint8_t* d_scratch, uint32_t* d_status, hipStream_t stream) {
    if (n_blocks == 0) return;
    dim3 grid((n_blocks + kInfThreads - 1) / kInfThreads), block(kInfThreads);
    size_t lds = (size_t)kInfThreads * kLaneLds;
    hipLaunchKernelGGL(k_bgzf_inflate, grid, block, lds, stream, d_comp, d_comp_off, d_comp_len, d_isize, d_out_off,
                       d_out, n_blocks, d_scratch, d_status);
    SBX_HIP(hipGetLastError());
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
