// depth.hip -- K3 `decode_accumulate`: per-position coverage counters from BAM records.
//
// Replaces the hot loop of `sambamba depth`: PileupRange.popFront / PileupRead.incrementPosition
// (BioD/bio/std/hts/bam/pileup.d:195-222,345-397), which advances every active read by one
// reference base per column, and PerBasePrinter.writeColumn's per-read classification
// (sambamba/depth.d:506-518,522-532).  Per SURVEY.md F5 / Appendix A the per-position output
// of that sweep line is a pure histogram over reads:
//     cnt[pos][sample][code] += 1  for every M/=/X base with qual >= min_bq,
//                                  code = Base5(base): A0 C1 G2 T3, everything else 4
//                                  (bio/core/base.d:85,163-186; 4-bit nibble, high nibble first,
//                                  read.d:364-383),
//     cnt[pos][sample][5]   += 1  for every position inside a D operation,
//     cnt[pos][sample][6]   += 1  for every position inside an N operation,
// so the GPU formulation is a scatter of read bases into position tiles:
//   * one workgroup owns one tile of T = 1024 / n_samples reference positions; its counters live in
//     LDS (T x n_samples x 7 u32 = 28.7 KB => 5 workgroups per CU) and are written to HBM exactly
//     once with coalesced stores -- no global atomics, no zero-fill pass over HBM;
//   * the records of a tile are a contiguous range [lo,hi) of the descriptor array (K2);
//     each wavefront pulls 64 descriptors with one coalesced load and ballots the ones that
//     overlap the tile; reads with a single run of aligned bases go three per pass (21 lanes x 8
//     bases each: one 8-byte quality load, one 8-byte packed-sequence load, 8 LDS atomics per
//     lane), general CIGARs are walked one read at a time across the 64 lanes;
//   * reads straddling a tile edge are visited by both tiles and clipped.
// Roofline: HBM.  Algorithmic bytes per read = record bytes + 28 B per covered position
// per sample written once.
#include <cstdlib>

#include "common.hpp"
#include "kernels.hpp"

namespace sbx {

namespace {

constexpr int kAccThreads = 256;
constexpr uint32_t kCigarType = 0x3C1A7u;   // cigar.d:116

__device__ __forceinline__ uint32_t ld32u(const uint8_t* p) {
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}

// Base5 internal code of a BAM 4-bit base code ("=ACMGRSVTWYHKDBN" -> A0 C1 G2 T3 else 4)
__device__ __forceinline__ uint32_t base5_of_nibble(uint32_t nib) {
    // 16 x 3-bit LUT packed into 48 bits
    const uint64_t lut = (4ULL << 0) | (0ULL << 3) | (1ULL << 6) | (4ULL << 9) | (2ULL << 12) | (4ULL << 15) |
                         (4ULL << 18) | (4ULL << 21) | (3ULL << 24) | (4ULL << 27) | (4ULL << 30) | (4ULL << 33) |
                         (4ULL << 36) | (4ULL << 39) | (4ULL << 42) | (4ULL << 45);
    return (uint32_t)(lut >> (nib * 3)) & 7u;
}

// LDS layout of a tile: positions are split by (p & 3) into 4 sub-arrays so that a lane owning 4
// consecutive positions (the dword fast path) and a lane owning 1 position (D/N runs) both spread
// over all 32 banks: dword index of position p = (p & 3) * sub_dw + (p >> 2) * (S * 7),
// sub_dw = (T / 4) * S * 7 + 8.
// (24-bit multiplies: full rate on CDNA, and p < 2^12, sub_dw < 2^12 -- the 32-bit v_mul_lo_u32 is quarter rate)
__device__ __forceinline__ uint32_t pos_dw(uint32_t p, uint32_t sub_dw, uint32_t s7) { return __umul24(p & 3u, sub_dw) + __umul24(p >> 2, s7); }

struct RecU {   // wave-uniform view of one record (values live in SGPRs)
    const uint8_t* seq;
    const uint8_t* qual;
    const uint8_t* cig;
    int32_t pos, end;
    uint32_t l_seq, n_cigar, kind, q_start, sample;
};

template <bool kSpan>
__global__ __launch_bounds__(kAccThreads) void k_accumulate(
    const uint8_t* __restrict__ U, const RecDesc* __restrict__ desc, const uint32_t* __restrict__ tile_lo,
    const uint32_t* __restrict__ tile_hi, const uint32_t* __restrict__ active, const uint32_t* __restrict__ tile_base,
    int32_t n_ref, uint32_t T, uint32_t S, uint32_t min_bq, uint32_t deep_thr, uint32_t* __restrict__ counters,
    uint32_t* __restrict__ span_out) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    {   // only the tiles k_accumulate16 left alone
        const uint32_t t = active[blockIdx.x];
        if (tile_hi[t] - tile_lo[t] < deep_thr) return;
    }
    const uint32_t s7 = S * 7;
    const uint32_t sub_dw = (T / 4) * s7 + 8;
    uint32_t* cnt = lds;                  // 4 * sub_dw dwords
    uint32_t* spn = lds + 4 * sub_dw;     // [T] (only when kSpan)
    const uint32_t tile = active[blockIdx.x];
    for (uint32_t i = threadIdx.x; i < 4 * sub_dw + (kSpan ? T : 0u); i += kAccThreads) lds[i] = 0;

    // which contig does this tile belong to?  (binary search in tile_base[0..n_ref])
    int lo_r = 0, hi_r = n_ref;   // invariant: tile_base[lo_r] <= tile < tile_base[hi_r]
    while (hi_r - lo_r > 1) {
        int mid = (lo_r + hi_r) >> 1;
        if (tile_base[mid] <= tile) lo_r = mid; else hi_r = mid;
    }
    const int32_t ts = (int32_t)((tile - tile_base[lo_r]) * T);   // first position of the tile in its contig
    const int32_t te = ts + (int32_t)T;
    const uint32_t r_lo = tile_lo[tile], r_hi = tile_hi[tile];
    __syncthreads();

    const uint32_t lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // (wave-uniform for the compiler too)

    // 4 bases of a match run held by this lane: bases q..q+nb-1 of the read at tile offsets p0..
    auto add4 = [&](uint32_t qw, uint32_t sw, uint32_t q, uint32_t nb, uint32_t p0, uint32_t sample) {
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) {
            if (k < nb) {
                const uint32_t qi = (q & 1u) + k;                        // nibble index inside sw
                const uint32_t byte = (sw >> (8u * (qi >> 1))) & 0xFFu;
                const uint32_t nib = (qi & 1u) ? (byte & 15u) : (byte >> 4);
                const uint32_t ql = (qw >> (8u * k)) & 0xFFu;
                if (ql >= min_bq) atomicAdd(&cnt[pos_dw(p0 + k, sub_dw, s7) + sample * 7 + base5_of_nibble(nib)], 1u);
            }
        }
    };
    // clipped range of a run [rp, rp+len) of aligned bases whose first base is query offset qp
    auto clip = [&](int32_t rp, uint32_t qp, uint32_t len, uint32_t l_seq, int32_t* i0, int32_t* i1) {
        *i0 = ts > rp ? ts - rp : 0;
        int32_t e = (int32_t)len < te - rp ? (int32_t)len : te - rp;
        if ((int64_t)qp + (int64_t)e > (int64_t)l_seq) e = (int32_t)l_seq - (int32_t)qp;   // malformed record guard
        *i1 = e;
    };
    // passes of a match run beyond what was prefetched (or all passes when first == 0)
    auto match_run = [&](const RecU& R, int32_t rp, uint32_t qp, uint32_t len, int32_t first) {
        int32_t i0, i1;
        clip(rp, qp, len, R.l_seq, &i0, &i1);
        for (int32_t i = i0 + first + 4 * (int32_t)lane; i < i1; i += 256) {
            const uint32_t q = qp + (uint32_t)i;
            const uint32_t nb = (uint32_t)(i1 - i) < 4u ? (uint32_t)(i1 - i) : 4u;
            // 4 quality bytes + the <= 3 sequence bytes holding bases q..q+3; the few bytes read past
            // the run stay inside the record (tags / next record) or the stream's 64-byte padding
            add4(ld32u(R.qual + q), ld32u(R.seq + (q >> 1)), q, nb, (uint32_t)(rp + i - ts), R.sample);
        }
    };
    auto gap_run = [&](const RecU& R, int32_t rp, uint32_t len, uint32_t code) {
        int32_t i0 = ts > rp ? ts - rp : 0;
        int64_t room = (int64_t)te - rp;
        int32_t i1 = (int64_t)len < room ? (int32_t)len : (int32_t)(room < 0 ? 0 : room);
        for (int32_t i = i0 + (int32_t)lane; i < i1; i += 64)
            atomicAdd(&cnt[pos_dw((uint32_t)(rp + i - ts), sub_dw, s7) + R.sample * 7 + code], 1u);
    };

    for (uint32_t c = r_lo + wave * 64u; c < r_hi; c += (kAccThreads / 64) * 64u) {
        const uint32_t ri = c + lane;
        RecDesc d;
        d.kind = 0;
        d.pos = 0; d.end = 0; d.rec_off = 0; d.l_seq = 0; d.n_cigar = 0; d.l_name = 0; d.q_start = 0; d.sample = 0;
        if (ri < r_hi) d = desc[ri];
        const bool take = d.kind != 0 && d.pos < te && d.end > ts;
        uint64_t mask = __ballot(take);
        // wave-uniform view of record r of this batch
        auto view = [&](int r) {
            RecU R;
            const uint32_t off_lo = __builtin_amdgcn_readlane((uint32_t)d.rec_off, r);
            const uint32_t off_hi = __builtin_amdgcn_readlane((uint32_t)(d.rec_off >> 32), r);
            R.pos = (int32_t)__builtin_amdgcn_readlane((uint32_t)d.pos, r);
            R.end = (int32_t)__builtin_amdgcn_readlane((uint32_t)d.end, r);
            R.l_seq = __builtin_amdgcn_readlane(d.l_seq, r);
            const uint32_t misc = __builtin_amdgcn_readlane((uint32_t)d.n_cigar | ((uint32_t)d.l_name << 16) | ((uint32_t)d.kind << 24), r);
            const uint32_t misc2 = __builtin_amdgcn_readlane((uint32_t)d.q_start | ((uint32_t)d.sample << 16), r);
            R.n_cigar = misc & 0xFFFFu;
            R.kind = misc >> 24;
            R.q_start = misc2 & 0xFFFFu;
            R.sample = (S > 1) ? (misc2 >> 16) : 0u;
            const uint8_t* rec = U + (((uint64_t)off_hi << 32) | off_lo);
            R.cig = rec + 36 + ((misc >> 16) & 0xFFu);
            R.seq = R.cig + 4 * R.n_cigar;
            R.qual = R.seq + ((R.l_seq + 1) >> 1);
            return R;
        };
        auto consume_general = [&](const RecU& R) {
            {
                // General CIGAR walk.  Every reference-consuming op occupies max(len, 1) columns (a
                // zero-length op still shows for one column in the reference's cursor, pileup.d:195-205)
                // and the read leaves the pileup at end = pos + sum(len) (read.d:1380-1383), which
                // truncates whatever is left.
                int32_t rp = R.pos;
                uint32_t qp = 0;
                for (uint32_t k = 0; k < R.n_cigar; ++k) {
                    uint32_t op = ld32u(R.cig + 4 * k);
                    uint32_t ty = (kCigarType >> ((op & 15u) * 2u)) & 3u, len = op >> 4;
                    if (ty & 2u) {
                        if (len == 0) len = 1;
                        const int32_t room = R.end - rp;
                        if ((int64_t)len > (int64_t)room) len = (uint32_t)(room > 0 ? room : 0);
                    }
                    if (ty == 3) {
                        match_run(R, rp, qp, len, 0);
                        rp += (int32_t)len;
                        qp += len;
                    } else if (ty == 2) {
                        gap_run(R, rp, len, (op & 15u) == 2u ? 5u : 6u);   // D -> DEL, otherwise (N) -> REFSKIP
                        rp += (int32_t)len;
                    } else if (ty == 1) {
                        qp += len;
                    }
                    if (rp >= te || rp >= R.end) break;
                }
            }
        };
        // ---- fast path: reads with a single run of aligned bases (kind 1), three per pass ----------
        // 21 lanes x 8 bases cover a 150-base read, so a pass handles three reads with 63 lanes busy:
        // per lane one 8-byte load of qualities, one of packed sequence, 8 LDS atomics.  The lane that
        // holds the descriptor precomputes the clipped run; the worker lanes fetch it by cross-lane
        // reads instead of 7 wave-uniform broadcasts per read.
        int32_t fi0 = 0, fi1 = 0;
        const bool fast = take && d.kind == 1;
        if (fast) clip(d.pos, d.q_start, (uint32_t)(d.end - d.pos), d.l_seq, &fi0, &fi1);
        const uint32_t f_n = fast && fi1 > fi0 ? (uint32_t)(fi1 - fi0) : 0u;         // bases of the run inside the tile
        const uint32_t f_q0 = (uint32_t)d.q_start + (uint32_t)fi0;                  // query offset of the first of them
        const uint32_t f_t0 = (uint32_t)(d.pos + fi0 - ts);                          // its tile offset
        const uint64_t f_seq = d.rec_off + 36u + d.l_name + 4u * (uint32_t)d.n_cigar;  // offset of the packed sequence in U
        const uint32_t f_qd = (d.l_seq + 1u) >> 1;                                   // qualities follow the sequence
        const uint32_t f_smp = S > 1 ? (uint32_t)d.sample : 0u;
        if (kSpan) {
            uint64_t ms = mask;
            while (ms) {
                const int r = __builtin_ctzll(ms);
                ms &= ms - 1;
                const int32_t rp = (int32_t)__builtin_amdgcn_readlane((uint32_t)d.pos, r);
                const int32_t re = (int32_t)__builtin_amdgcn_readlane((uint32_t)d.end, r);
                const int32_t a = rp > ts ? rp : ts, b2 = re < te ? re : te;
                for (int32_t p = a + (int32_t)lane; p < b2; p += 64) atomicAdd(&spn[p - ts], 1u);
            }
        }
        uint64_t m1 = __ballot(f_n != 0);
        const uint32_t grp = lane / 21u, sub = lane - grp * 21u;     // lane 63: grp 3 = idle
        while (m1) {
            int r0 = __builtin_ctzll(m1); m1 &= m1 - 1;
            int r1 = -1, r2 = -1;
            if (m1) { r1 = __builtin_ctzll(m1); m1 &= m1 - 1; }
            if (m1) { r2 = __builtin_ctzll(m1); m1 &= m1 - 1; }
            const int src = grp == 0 ? r0 : grp == 1 ? r1 : grp == 2 ? r2 : -1;
            const int sl = src < 0 ? 0 : src;
            const uint32_t n_src = __shfl(f_n, sl, 64);     // every lane takes part: a masked-off source lane reads as 0
            const uint32_t n = src < 0 ? 0u : n_src;
            const uint32_t q0 = __shfl(f_q0, sl, 64), t0 = __shfl(f_t0, sl, 64), qd = __shfl(f_qd, sl, 64);
            const uint32_t s_lo = __shfl((uint32_t)f_seq, sl, 64), s_hi = __shfl((uint32_t)(f_seq >> 32), sl, 64);
            const uint32_t smp = S > 1 ? __shfl(f_smp, sl, 64) : 0u;
            const uint8_t* seq = U + (((uint64_t)s_hi << 32) | s_lo);
            const uint8_t* qual = seq + qd;
            for (uint32_t j = 8u * sub; __any(j < n); j += 168u) {
                if (j < n) {
                    const uint32_t q = q0 + j;
                    const uint32_t nb = n - j < 8u ? n - j : 8u;
                    // 8 quality bytes and the <= 5 sequence bytes holding bases q..q+7 (reads past the run stay
                    // inside the record / the stream's padding)
                    const uint32_t qw0 = ld32u(qual + q), qw1 = ld32u(qual + q + 4);
                    const uint32_t sw0 = ld32u(seq + (q >> 1)), sw1 = ld32u(seq + (q >> 1) + 4);
                    const uint64_t qw = (uint64_t)qw0 | ((uint64_t)qw1 << 32);
                    const uint64_t sw = (uint64_t)sw0 | ((uint64_t)sw1 << 32);
#pragma unroll
                    for (uint32_t k = 0; k < 8; ++k) {
                        if (k < nb) {
                            const uint32_t qi = (q & 1u) + k;                            // nibble index inside sw
                            const uint32_t byte = (uint32_t)(sw >> (8u * (qi >> 1))) & 0xFFu;
                            const uint32_t nib = (qi & 1u) ? (byte & 15u) : (byte >> 4);
                            const uint32_t ql = (uint32_t)(qw >> (8u * k)) & 0xFFu;
                            if (ql >= min_bq) atomicAdd(&cnt[pos_dw(t0 + j + k, sub_dw, s7) + __umul24(smp, 7u) + base5_of_nibble(nib)], 1u);
                        }
                    }
                }
            }
        }
        // ---- general CIGARs (kind 2): one read at a time, wave-uniform CIGAR walk ------------------------
        uint64_t m2 = __ballot(take && d.kind == 2);
        while (m2) {
            const int r = __builtin_ctzll(m2);
            m2 &= m2 - 1;
            const RecU R = view(r);
            consume_general(R);
        }
    }
    __syncthreads();
    // write the tile once, coalesced (undoing the (p & 3) split)
    const uint32_t n_cnt = T * s7;
    uint32_t* out = counters + (size_t)blockIdx.x * n_cnt;
    // i / s7 by reciprocal multiplication (exact for i < 2^16; one real division per thread instead of 28)
    const uint32_t inv_s7 = 0xFFFFFFFFu / s7 + 1u;
    for (uint32_t i = threadIdx.x; i < n_cnt; i += kAccThreads) {
        const uint32_t p = __umulhi(i, inv_s7), k = i - __umul24(p, s7);
        out[i] = cnt[pos_dw(p, sub_dw, s7) + k];
    }
    if (kSpan) {
        uint32_t* so = span_out + (size_t)blockIdx.x * T;
        for (uint32_t i = threadIdx.x; i < T; i += kAccThreads) so[i] = spn[i];
    }
}


// ---- K3, 16-bit LDS counters ----------------------------------------------------------------------------
// The tile kernel above keeps 32-bit counters in LDS (28.7 KB per tile => 5 workgroups per CU) and is bound by how
// many bytes a CU has in flight, not by HBM.  No counter of a tile can exceed the number of records whose range
// [lo,hi) K2 marked for it, so every tile with fewer than 2^16 of them -- all of a WGS, any amplicon stack below
// 65,535x -- is accumulated in 16-bit counters packed two per dword:
//   * 16 bytes of LDS per (position, sample): {A|C}, {G|T}, {other|DEL}, {REFSKIP|-}; ds_add_u32 of 1 or 1<<16; a tile
//     takes 16 KB and eight workgroups (32 waves, the CU's limit) are resident;
//   * position p sits in slot p ^ ((p >> 4) & 7): the ten lanes of a read, 16 positions apart, fall into different
//     bank groups instead of one;
//   * fast path: 16 bases per lane, ten lanes per read, six reads per pass -- 12 bytes of packed sequence per lane,
//     plus 16 bytes of qualities only when a base-quality threshold is set (-q 0, the default, never touches the
//     quality strings: 150 of a record's ~283 bytes stay in HBM); the loads of the next pass are issued before the
//     LDS atomics of the current one;
//   * span counts (kSpan) are kept as a difference array (+1 at the first, -1 behind the last covered position of
//     a read) and integrated once when the tile is written;
//   * blockIdx -> tile slot is XCD-aware: workgroup b runs on XCD b % 8, so XCD x takes the x-th eighth of the
//     active tiles in order and the reads straddling two tiles are found in the L2 that fetched them.
// Tiles with deep_thr (2^16) or more records are left to k_accumulate (launched only when tile_compact counted any).
constexpr uint32_t kGrpLanes = 10;       // lanes per read in the fast path
constexpr uint32_t kGrpBases = 160;      // 16 bases per lane

__device__ __forceinline__ uint32_t swz(uint32_t p) { return p ^ ((p >> 4) & 7u); }

__device__ __forceinline__ uint32_t nibble_swap(uint32_t x) { return ((x & 0x0F0F0F0Fu) << 4) | ((x >> 4) & 0x0F0F0F0Fu); }

template <bool kSpan, bool kQual, bool kOneSample>
__global__ __launch_bounds__(kAccThreads) void k_accumulate16(
    const uint8_t* __restrict__ U, const RecDesc* __restrict__ desc, const uint32_t* __restrict__ tile_lo,
    const uint32_t* __restrict__ tile_hi, const uint32_t* __restrict__ active, uint32_t n_active,
    const uint32_t* __restrict__ tile_base, int32_t n_ref, uint32_t T, uint32_t S_arg, uint32_t min_bq, uint32_t deep_thr,
    uint32_t* __restrict__ counters, uint32_t* __restrict__ span_out, uint32_t compact) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const uint32_t S = kOneSample ? 1u : S_arg;      // (one sample -- a single-sample BAM or --combined -- is the common case:
                                                     //  the per-base address arithmetic loses its multiplications)
    // XCD-aware slot: the grid has 8 * per workgroups
    const uint32_t per = gridDim.x >> 3;
    const uint32_t slot = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
    if (slot >= n_active) return;
    const uint32_t tile = active[slot];
    const uint32_t r_lo = tile_lo[tile], r_hi = tile_hi[tile];
    if (r_hi - r_lo >= deep_thr) return;                // left to the 32-bit kernel
    uint32_t* cnt = lds;                                  // [T][S][4] dwords
    int32_t* spn = (int32_t*)(lds + T * S * 4);           // [T + 1] difference array (kSpan), then 4 wave totals
    const uint32_t n_lds = T * S * 4 + (kSpan ? T + 1 + 4 : 0u);
    for (uint32_t i = threadIdx.x; i < n_lds; i += kAccThreads) lds[i] = 0;

    int lo_r = 0, hi_r = n_ref;   // invariant: tile_base[lo_r] <= tile < tile_base[hi_r]
    while (hi_r - lo_r > 1) {
        int mid = (lo_r + hi_r) >> 1;
        if (tile_base[mid] <= tile) lo_r = mid; else hi_r = mid;
    }
    const int32_t ts = (int32_t)((tile - tile_base[lo_r]) * T);
    const int32_t te = ts + (int32_t)T;
    __syncthreads();

    const uint32_t lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // (wave-uniform for the compiler too)
    // counter dword of (tile offset p, sample, code) and its increment
    auto add_code = [&](uint32_t p, uint32_t sample, uint32_t code) {
        atomicAdd(&cnt[(__umul24(swz(p), S) + sample) * 4u + (code >> 1)], 1u << ((code & 1u) << 4));
    };
    auto clip = [&](int32_t rp, uint32_t qp, uint32_t len, uint32_t l_seq, int32_t* i0, int32_t* i1) {
        *i0 = ts > rp ? ts - rp : 0;
        int32_t e = (int32_t)len < te - rp ? (int32_t)len : te - rp;
        if ((int64_t)qp + (int64_t)e > (int64_t)l_seq) e = (int32_t)l_seq - (int32_t)qp;   // malformed record guard
        *i1 = e;
    };
    // general path helpers: a run of aligned bases / of D or N positions of one read, all 64 lanes
    auto match_run = [&](const RecU& R, int32_t rp, uint32_t qp, uint32_t len) {
        int32_t i0, i1;
        clip(rp, qp, len, R.l_seq, &i0, &i1);
        for (int32_t i = i0 + 4 * (int32_t)lane; i < i1; i += 256) {
            const uint32_t q = qp + (uint32_t)i;
            const uint32_t nb = (uint32_t)(i1 - i) < 4u ? (uint32_t)(i1 - i) : 4u;
            const uint32_t qw = kQual ? ld32u(R.qual + q) : 0u, sw = ld32u(R.seq + (q >> 1));
#pragma unroll
            for (uint32_t k = 0; k < 4; ++k) {
                if (k < nb) {
                    const uint32_t qi = (q & 1u) + k;
                    const uint32_t byte = (sw >> (8u * (qi >> 1))) & 0xFFu;
                    const uint32_t nib = (qi & 1u) ? (byte & 15u) : (byte >> 4);
                    const uint32_t ql = (qw >> (8u * k)) & 0xFFu;
                    if (!kQual || ql >= min_bq) add_code((uint32_t)(rp + i - ts) + k, R.sample, base5_of_nibble(nib));
                }
            }
        }
    };
    auto gap_run = [&](const RecU& R, int32_t rp, uint32_t len, uint32_t code) {
        int32_t i0 = ts > rp ? ts - rp : 0;
        int64_t room = (int64_t)te - rp;
        int32_t i1 = (int64_t)len < room ? (int32_t)len : (int32_t)(room < 0 ? 0 : room);
        for (int32_t i = i0 + (int32_t)lane; i < i1; i += 64) add_code((uint32_t)(rp + i - ts), R.sample, code);
    };

    for (uint32_t c = r_lo + wave * 64u; c < r_hi; c += (kAccThreads / 64) * 64u) {
        const uint32_t ri = c + lane;
        RecDesc d;
        d.kind = 0;
        d.pos = 0; d.end = 0; d.rec_off = 0; d.l_seq = 0; d.n_cigar = 0; d.l_name = 0; d.q_start = 0; d.sample = 0;
        if (ri < r_hi) d = desc[ri];
        const bool take = d.kind != 0 && d.pos < te && d.end > ts;
        if (kSpan && take) {
            const int32_t a = d.pos > ts ? d.pos : ts, b2 = d.end < te ? d.end : te;
            atomicAdd(&spn[a - ts], 1);
            atomicAdd(&spn[b2 - ts], -1);
        }
        // ---- fast path: one run of aligned bases with at most 160 of them inside the tile ------------------
        int32_t fi0 = 0, fi1 = 0;
        const bool one_run = take && d.kind == 1;
        if (one_run) clip(d.pos, d.q_start, (uint32_t)(d.end - d.pos), d.l_seq, &fi0, &fi1);
        const uint32_t run_n = one_run && fi1 > fi0 ? (uint32_t)(fi1 - fi0) : 0u;
        const bool fast = run_n != 0 && run_n <= kGrpBases;
        const uint32_t f_n = fast ? run_n : 0u;                                       // bases of the run inside the tile
        const uint32_t f_q0 = (uint32_t)d.q_start + (uint32_t)fi0;                  // query offset of the first of them
        const uint32_t f_t0 = (uint32_t)(d.pos + fi0 - ts);                          // its tile offset
        const uint64_t f_seq = d.rec_off + 36u + d.l_name + 4u * (uint32_t)d.n_cigar;  // offset of the packed sequence in U
        const uint32_t f_qd = (d.l_seq + 1u) >> 1;                                   // qualities follow the sequence
        const uint32_t f_smp = S > 1 ? (uint32_t)d.sample : 0u;
        const uint32_t grp = lane / kGrpLanes, sub = lane - grp * kGrpLanes;         // lanes 60..63: grp 6 = idle
        uint64_t m1 = __ballot(fast);

        struct Pass { uint32_t q[4]; uint32_t s[3]; uint32_t nb, par, t, smp; };
        auto fetch = [&](uint64_t& m) {
            int r[6];
#pragma unroll
            for (int g = 0; g < 6; ++g) {
                r[g] = -1;
                if (m) { r[g] = __builtin_ctzll(m); m &= m - 1; }
            }
            const int src = grp == 0 ? r[0] : grp == 1 ? r[1] : grp == 2 ? r[2] : grp == 3 ? r[3] : grp == 4 ? r[4] : grp == 5 ? r[5] : -1;
            const int sl = src < 0 ? 0 : src;
            const uint32_t n_src = __shfl(f_n, sl, 64);     // every lane takes part
            const uint32_t n = src < 0 ? 0u : n_src;
            const uint32_t q0 = __shfl(f_q0, sl, 64), t0 = __shfl(f_t0, sl, 64);
            const uint32_t s_lo = __shfl((uint32_t)f_seq, sl, 64), s_hi = __shfl((uint32_t)(f_seq >> 32), sl, 64);
            Pass P;
            P.smp = S > 1 ? __shfl(f_smp, sl, 64) : 0u;
            const uint32_t qd = kQual ? __shfl(f_qd, sl, 64) : 0u;
            const uint32_t j = 16u * sub;
            P.nb = j < n ? (n - j < 16u ? n - j : 16u) : 0u;
            const uint32_t q = q0 + j;
            P.par = q & 1u;
            P.t = t0 + j;
            P.q[0] = P.q[1] = P.q[2] = P.q[3] = 0; P.s[0] = P.s[1] = P.s[2] = 0;
            if (P.nb) {
                // the <= 9 sequence bytes holding bases q..q+15 and their 16 qualities (reads past the run stay inside
                // the record / the stream's 64-byte padding)
                const uint8_t* seq = U + (((uint64_t)s_hi << 32) | s_lo);
                const uint8_t* sp = seq + (q >> 1);
                P.s[0] = ld32u(sp); P.s[1] = ld32u(sp + 4); P.s[2] = ld32u(sp + 8);
                if (kQual) {
                    const uint8_t* qp = seq + qd + q;
                    P.q[0] = ld32u(qp); P.q[1] = ld32u(qp + 4); P.q[2] = ld32u(qp + 8); P.q[3] = ld32u(qp + 12);
                }
            }
            return P;
        };
        auto consume = [&](const Pass& P) {
            if (P.nb == 0) return;
            // nibbles in stream order (high nibble of a byte first), the lane's first base at bit 0
            const uint32_t y0 = nibble_swap(P.s[0]), y1 = nibble_swap(P.s[1]), y2 = nibble_swap(P.s[2]);
            const uint32_t sh = P.par * 4u;
            const uint32_t b0 = __builtin_amdgcn_alignbit(y1, y0, sh), b1 = __builtin_amdgcn_alignbit(y2, y1, sh);
            const uint32_t base_dw = __umul24(P.smp, 4u);
#pragma unroll
            for (uint32_t k = 0; k < 16; ++k) {
                if (k < P.nb) {
                    const uint32_t nib = ((k < 8 ? b0 : b1) >> (4u * (k & 7u))) & 15u;
                    const uint32_t ql = kQual ? (P.q[k >> 2] >> (8u * (k & 3u))) & 0xFFu : 0u;
                    if (!kQual || ql >= min_bq) {
                        const uint32_t code = base5_of_nibble(nib);
                        atomicAdd(&cnt[__umul24(swz(P.t + k), S * 4u) + base_dw + (code >> 1)], 1u << ((code & 1u) << 4));
                    }
                }
            }
        };
        if (m1) {
            Pass cur = fetch(m1);
            for (;;) {
                const bool more = m1 != 0;
                Pass nxt;
                if (more) nxt = fetch(m1);
                consume(cur);
                if (!more) break;
                cur = nxt;
            }
        }
        // ---- everything else (general CIGARs, runs longer than 160 bases): one read at a time, 64 lanes -------
        uint64_t m2 = __ballot(take && !fast && (d.kind == 2 || run_n != 0));
        while (m2) {
            const int r = __builtin_ctzll(m2);
            m2 &= m2 - 1;
            RecU R;
            const uint32_t off_lo = __builtin_amdgcn_readlane((uint32_t)d.rec_off, r);
            const uint32_t off_hi = __builtin_amdgcn_readlane((uint32_t)(d.rec_off >> 32), r);
            R.pos = (int32_t)__builtin_amdgcn_readlane((uint32_t)d.pos, r);
            R.end = (int32_t)__builtin_amdgcn_readlane((uint32_t)d.end, r);
            R.l_seq = __builtin_amdgcn_readlane(d.l_seq, r);
            const uint32_t misc = __builtin_amdgcn_readlane((uint32_t)d.n_cigar | ((uint32_t)d.l_name << 16) | ((uint32_t)d.kind << 24), r);
            const uint32_t misc2 = __builtin_amdgcn_readlane((uint32_t)d.q_start | ((uint32_t)d.sample << 16), r);
            R.n_cigar = misc & 0xFFFFu;
            R.kind = misc >> 24;
            R.q_start = misc2 & 0xFFFFu;
            R.sample = (S > 1) ? (misc2 >> 16) : 0u;
            const uint8_t* rec = U + (((uint64_t)off_hi << 32) | off_lo);
            R.cig = rec + 36 + ((misc >> 16) & 0xFFu);
            R.seq = R.cig + 4 * R.n_cigar;
            R.qual = R.seq + ((R.l_seq + 1) >> 1);
            // CIGAR walk as in k_accumulate: a reference-consuming op occupies max(len, 1) columns (pileup.d:195-205) and the
            // read leaves the pileup at end = pos + sum(len) (read.d:1380-1383)
            int32_t rp = R.pos;
            uint32_t qp = 0;
            for (uint32_t k = 0; k < R.n_cigar; ++k) {
                uint32_t op = ld32u(R.cig + 4 * k);
                uint32_t ty = (kCigarType >> ((op & 15u) * 2u)) & 3u, len = op >> 4;
                if (ty & 2u) {
                    if (len == 0) len = 1;
                    const int32_t room = R.end - rp;
                    if ((int64_t)len > (int64_t)room) len = (uint32_t)(room > 0 ? room : 0);
                }
                if (ty == 3) {
                    match_run(R, rp, qp, len);
                    rp += (int32_t)len;
                    qp += len;
                } else if (ty == 2) {
                    gap_run(R, rp, len, (op & 15u) == 2u ? 5u : 6u);   // D -> DEL, otherwise (N) -> REFSKIP
                    rp += (int32_t)len;
                } else if (ty == 1) {
                    qp += len;
                }
                if (rp >= te || rp >= R.end) break;
            }
        }
    }
    __syncthreads();
    // ---- write the tile once: 16-bit pairs -> u32[T][S][7], coalesced 16-byte stores --------------------------
    const uint32_t s7 = S * 7;
    const uint32_t n_cnt = T * s7;                  // multiple of 4 (T >= 16)
    if (compact) {
        // region / window statistics need two numbers per position and sample -- the bases counted (codes 0..4) and the depth (all 7:
        // D / N count as quality 255), depth.d:661-698,760-845 -- not the seven counters: one word {bases : 16 | depth : 16} (a tile of
        // this kernel holds fewer than 2^16 records), 4 bytes per position instead of 28 (these modes print O(windows))
        const uint32_t n_ps = T * S;                // multiple of 4
        uint32_t* outc = counters + (size_t)slot * n_ps;
        const uint32_t inv_s = 0xFFFFFFFFu / S + 1u;
        for (uint32_t i4 = threadIdx.x * 4u; i4 < n_ps; i4 += kAccThreads * 4u) {
            uint32_t v[4];
#pragma unroll
            for (uint32_t e = 0; e < 4; ++e) {
                const uint32_t i = i4 + e;
                const uint32_t p = kOneSample ? i : __umulhi(i, inv_s), smp = kOneSample ? 0u : i - __umul24(p, S);
                const uint32_t* w = cnt + (__umul24(swz(p), S) + smp) * 4u;
                const uint32_t m = (w[0] & 0xFFFFu) + (w[0] >> 16) + (w[1] & 0xFFFFu) + (w[1] >> 16) + (w[2] & 0xFFFFu);
                v[e] = m | ((m + (w[2] >> 16) + (w[3] & 0xFFFFu)) << 16);
            }
            *(uint4*)(outc + i4) = make_uint4(v[0], v[1], v[2], v[3]);
        }
    }
    uint32_t* out = counters + (size_t)slot * n_cnt;
    const uint32_t inv_s7 = 0xFFFFFFFFu / s7 + 1u;      // i / s7 by reciprocal multiplication (exact for i < 2^16)
    const uint32_t inv_7 = 0xFFFFFFFFu / 7u + 1u;
    for (uint32_t i4 = threadIdx.x * 4u; !compact && i4 < n_cnt; i4 += kAccThreads * 4u) {
        uint32_t v[4];
#pragma unroll
        for (uint32_t e = 0; e < 4; ++e) {
            const uint32_t i = i4 + e;
            const uint32_t p = __umulhi(i, inv_s7), rem = i - __umul24(p, s7);
            const uint32_t smp = __umulhi(rem, inv_7), k = rem - smp * 7u;
            const uint32_t w = cnt[(__umul24(swz(p), S) + smp) * 4u + (k >> 1)];
            v[e] = (k & 1u) ? (w >> 16) : (w & 0xFFFFu);
        }
        *(uint4*)(out + i4) = make_uint4(v[0], v[1], v[2], v[3]);
    }
    if (kSpan) {
        // integrate the difference array: every thread owns `chunk` consecutive positions
        const uint32_t chunk = (T + kAccThreads - 1) / kAccThreads;
        const uint32_t p0 = threadIdx.x * chunk;
        int32_t loc = 0;
        for (uint32_t k = 0; k < chunk; ++k) if (p0 + k < T) loc += spn[p0 + k];
        int32_t incl = loc;
#pragma unroll
        for (int dlt = 1; dlt < 64; dlt <<= 1) {
            const int32_t o = __shfl_up(incl, dlt, 64);
            if ((int)lane >= dlt) incl += o;
        }
        int32_t* wtot = spn + T + 1;
        if (lane == 63) wtot[wave] = incl;
        __syncthreads();
        int32_t run = incl - loc;
        for (uint32_t w = 0; w < wave; ++w) run += wtot[w];
        uint32_t* so = span_out + (size_t)slot * T;
        for (uint32_t k = 0; k < chunk; ++k) {
            if (p0 + k < T) { run += spn[p0 + k]; so[p0 + k] = (uint32_t)run; }
        }
    }
}


// ---- K3, variant 2: one sample, no base-quality threshold, no span counts -- the headline path -------------------------------
// k_accumulate16 above spends ~18 VALU instructions per base on what is, per base, "pick a 16-bit counter and add 1": nibble,
// code, swizzled address, increment, range test.  Here every lane owns an ALIGNED block of 16 tile positions of one read:
//   * LDS layout: position p at dword 4 p + (p >> 4) -- linear, one pad dword per 16 positions, so that lanes 16 positions apart
//     fall on consecutive banks without a per-position swizzle, and base k of a lane is `block address + code offset` with 16 k
//     in the instruction's offset field;
//   * the counter dword and the half it increments come from two 32-bit lookup constants indexed by 2 x nibble (v_bfe_u32),
//     the range test is one bit of a per-lane validity mask folded into the increment (an out-of-range base adds 0):
//     8 VALU instructions + one ds_add_u32 per base;
//   * a read covers 10 or 11 aligned blocks depending on where it starts, so reads are packed into lanes by a prefix sum of
//     their block counts (a byte map lane-slot -> read in LDS, built once per 64 descriptors) instead of fixed groups of ten:
//     64 blocks per pass, no idle lanes but in the last pass of a batch;
//   * general CIGARs and long runs go one read at a time, but through the same block routine (all 64 lanes, wave-uniform
//     parameters) and the same address arithmetic.
// bits [2n, 2n + 1]: counter dword of 4-bit base code n -- A (1), C (2) -> 0; G (4), T (8) -> 1; everything else -> 2 ("other")
#define SBX_LUT_OFF_ENTRY(n) (((n) == 1 || (n) == 2 ? 0u : (n) == 4 || (n) == 8 ? 1u : 2u) << (2 * (n)))
#define kLutOff (SBX_LUT_OFF_ENTRY(0) | SBX_LUT_OFF_ENTRY(1) | SBX_LUT_OFF_ENTRY(2) | SBX_LUT_OFF_ENTRY(3) | SBX_LUT_OFF_ENTRY(4) | SBX_LUT_OFF_ENTRY(5) | \
                 SBX_LUT_OFF_ENTRY(6) | SBX_LUT_OFF_ENTRY(7) | SBX_LUT_OFF_ENTRY(8) | SBX_LUT_OFF_ENTRY(9) | SBX_LUT_OFF_ENTRY(10) | SBX_LUT_OFF_ENTRY(11) | \
                 SBX_LUT_OFF_ENTRY(12) | SBX_LUT_OFF_ENTRY(13) | SBX_LUT_OFF_ENTRY(14) | SBX_LUT_OFF_ENTRY(15))
#define kLutHalf ((1u << 4) | (1u << 16))          // bit 2n: code n counts in the high half of its dword (C, T)

__device__ __forceinline__ uint32_t blk_dw(uint32_t p) { return 4u * p + (p >> 4); }      // dword index of tile position p

// one aligned block: positions pb .. pb + 15 of the tile, validity mask vm, 16 nibbles (stream order) in b_lo | b_hi
__device__ __forceinline__ void add_block(uint32_t* cnt, uint32_t pb, uint32_t vm, uint32_t b_lo, uint32_t b_hi) {
    uint32_t* base = cnt + (pb >> 4) * 65u;
#pragma unroll
    for (uint32_t k = 0; k < 16; ++k) {
        const uint32_t w = k < 8 ? b_lo : b_hi, sh = 4u * (k & 7u);
        const uint32_t nib2 = sh ? (w >> (sh - 1u)) & 0x1Eu : (w << 1) & 0x1Eu;
        const uint32_t off = (kLutOff >> nib2) & 3u;
        const uint32_t half = ((kLutHalf >> nib2) & 1u) << 4;
        const uint32_t inc = ((vm >> k) & 1u) << half;
        atomicAdd(base + off + 4u * k, inc);
    }
}

// the 16 nibbles of tile positions pb .. pb + 15 of a run whose first base (tile offset t0) is nibble `par` of byte *seq0
__device__ __forceinline__ void load_block(const uint8_t* seq0, uint32_t par, uint32_t t0, uint32_t pb, uint32_t* b_lo, uint32_t* b_hi) {
    const int32_t qi = (int32_t)par + (int32_t)pb - (int32_t)t0;      // >= -15: the block may start before the run
    const uint8_t* sp = seq0 + (qi >> 1);                               // (reads a few bytes before the sequence: inside the record)
    const uint32_t y0 = nibble_swap(ld32u(sp)), y1 = nibble_swap(ld32u(sp + 4)), y2 = nibble_swap(ld32u(sp + 8));
    const uint32_t sh = ((uint32_t)qi & 1u) * 4u;
    *b_lo = __builtin_amdgcn_alignbit(y1, y0, sh);
    *b_hi = __builtin_amdgcn_alignbit(y2, y1, sh);
}

constexpr uint32_t kMapBytes = 704;      // 64 reads x 11 blocks


// variant 3 of K3: variant 2 plus a second stage that sends the runs of general CIGARs through the packed block routine too
typedef uint32_t u32x4c __attribute__((ext_vector_type(4)));
constexpr uint32_t kLaneOps = 8;          // CIGARs of up to this many operations are walked by their lane
constexpr uint32_t kRunItems = 22;        // run items of a wave per 64 descriptors (16 bytes each: 352 bytes) ...
constexpr uint32_t kRunBlocks = 352;      // ... and the lane map of their blocks: together the 704 bytes of the first stage's map
constexpr uint32_t kGapItems = 8;
constexpr uint32_t kWaveBytesC = kMapBytes + kGapItems * 8u;


__global__ __launch_bounds__(kAccThreads) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_accumulate16c(
    const uint8_t* __restrict__ U, const RecDesc* __restrict__ desc, const uint32_t* __restrict__ tile_lo,
    const uint32_t* __restrict__ tile_hi, const uint32_t* __restrict__ active, uint32_t n_active,
    const uint32_t* __restrict__ tile_base, int32_t n_ref, uint32_t T, uint32_t deep_thr, uint32_t* __restrict__ counters, uint32_t compact) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const uint32_t per = gridDim.x >> 3;
    const uint32_t slot = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);      // XCD-aware, as k_accumulate16
    if (slot >= n_active) return;
    const uint32_t tile = active[slot];
    const uint32_t r_lo = tile_lo[tile], r_hi = tile_hi[tile];
    if (r_hi - r_lo >= deep_thr) return;                // left to the 32-bit kernel
    const uint32_t n_cnt_dw = 4u * T + (T >> 4);
    uint32_t* cnt = lds;
    const uint32_t lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // (wave-uniform for the compiler too)
    // per wave: 704 bytes -- the lane map of the single-run reads, then (second stage) 22 run items of 16 bytes + the lane map
    // of their blocks -- and 8 gap items of 8 bytes
    uint8_t* map = (uint8_t*)(lds + ((n_cnt_dw + 3u) & ~3u)) + wave * kWaveBytesC;
    u32x4c* runq = (u32x4c*)map;                         // [kRunItems] {byte offset lo, hi | parity << 31, t0 | n << 16, first block}
    uint8_t* map2 = map + kRunItems * 16u;              // [kRunBlocks]
    uint32_t* gapq = (uint32_t*)(map + kMapBytes);      // [kGapItems][2] {t0 | n << 16, code}
    for (uint32_t i = threadIdx.x; i < n_cnt_dw; i += kAccThreads) lds[i] = 0;

    int lo_r = 0, hi_r = n_ref;   // invariant: tile_base[lo_r] <= tile < tile_base[hi_r]
    while (hi_r - lo_r > 1) {
        int mid = (lo_r + hi_r) >> 1;
        if (tile_base[mid] <= tile) lo_r = mid; else hi_r = mid;
    }
    const int32_t ts = (int32_t)((tile - tile_base[lo_r]) * T);
    const int32_t te = ts + (int32_t)T;
    __syncthreads();

    for (uint32_t c = r_lo + wave * 64u; c < r_hi; c += (kAccThreads / 64) * 64u) {
        const uint32_t ri = c + lane;
        RecDesc d;
        d.kind = 0;
        d.pos = 0; d.end = 0; d.rec_off = 0; d.l_seq = 0; d.n_cigar = 0; d.l_name = 0; d.q_start = 0; d.sample = 0;
        if (ri < r_hi) d = desc[ri];
        const bool take = d.kind != 0 && d.pos < te && d.end > ts;
        // ---- reads with one run of aligned bases: clipped run [t0, t0 + n) of the tile, first base = query offset q0 ----------
        int32_t i0 = 0, i1 = 0;
        const bool one_run = take && d.kind == 1;
        if (one_run) {
            const int32_t len = d.end - d.pos;
            i0 = ts > d.pos ? ts - d.pos : 0;
            i1 = len < te - d.pos ? len : te - d.pos;
            if ((int64_t)d.q_start + (int64_t)i1 > (int64_t)d.l_seq) i1 = (int32_t)d.l_seq - (int32_t)d.q_start;   // malformed record guard
        }
        const uint32_t n_run = one_run && i1 > i0 ? (uint32_t)(i1 - i0) : 0u;
        const uint32_t t0 = (uint32_t)(d.pos + i0 - ts);
        const uint32_t nblk_all = n_run ? ((t0 + n_run - 1u) >> 4) - (t0 >> 4) + 1u : 0u;
        const bool fast = n_run != 0 && nblk_all <= 11u;
        const uint32_t nblk = fast ? nblk_all : 0u;
        const uint32_t q0 = (uint32_t)d.q_start + (uint32_t)i0;
        const uint64_t a64 = d.rec_off + 36u + d.l_name + 4u * (uint32_t)d.n_cigar + (q0 >> 1);     // byte of the run's first base in U
        // packed per-read parameters, fetched by the worker lanes with cross-lane reads
        uint32_t incl = nblk;
#pragma unroll
        for (int dlt = 1; dlt < 64; dlt <<= 1) {
            const uint32_t o = __shfl_up(incl, dlt, 64);
            if ((int)lane >= dlt) incl += o;
        }
        const uint32_t start = incl - nblk;
        const uint32_t total = __builtin_amdgcn_readlane(incl, 63);
        const uint32_t p_lo = (uint32_t)a64;
        const uint32_t p_hi = (uint32_t)(a64 >> 32) | (start << 16) | ((q0 & 1u) << 31);      // 48-bit offset | start (<= 704) | parity
        const uint32_t p_tn = t0 | (n_run << 16);
        for (uint32_t j = 0; j < 11u; ++j)
            if (j < nblk) map[start + j] = (uint8_t)lane;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (uint32_t g0 = 0; g0 < total; g0 += 64u) {
            const uint32_t g = g0 + lane;
            const bool on = g < total;
            const uint32_t r = on ? (uint32_t)map[g] : 0u;
            const uint32_t w_lo = __shfl(p_lo, r, 64), w_hi = __shfl(p_hi, r, 64), w_tn = __shfl(p_tn, r, 64);
            if (on) {
                const uint32_t rt0 = w_tn & 0xFFFFu, rn = w_tn >> 16;
                const uint32_t j = g - ((w_hi >> 16) & 0x7FFFu);
                const uint32_t pb = ((rt0 >> 4) + j) << 4;
                const uint32_t k0 = pb < rt0 ? rt0 - pb : 0u;
                const uint32_t k1 = rt0 + rn - pb < 16u ? rt0 + rn - pb : 16u;
                const uint32_t vm = ((1u << k1) - 1u) & ~((1u << k0) - 1u);
                const uint8_t* seq0 = U + (((uint64_t)(w_hi & 0xFFFFu) << 32) | w_lo);
                uint32_t b_lo, b_hi;
                load_block(seq0, w_hi >> 31, rt0, pb, &b_lo, &b_hi);
                add_block(cnt, pb, vm, b_lo, b_hi);
            }
        }
        __builtin_amdgcn_wave_barrier();      // the map is rewritten by the next batch
        // ---- second stage: general CIGARs.  Every lane walks its own read's CIGAR (the lanes stay in one loop over the operation
        // index); runs of aligned bases that touch the tile become items {first base, tile offset, length} in the wave's queue, D / N
        // runs go to the gap queue; the items then flow through the same packed block routine as above.  A batch whose items do
        // not fit the queues (a wave of long reads) falls back to the one-read-at-a-time loop below.
        const bool k2 = take && d.kind == 2 && d.n_cigar <= kLaneOps;
        uint32_t nq = 0, ng = 0, qblocks = 0;
        bool overflow = false;
        if (__any(k2)) {
            const uint8_t* cig = U + d.rec_off + 36u + d.l_name;
            const uint64_t seq_off = d.rec_off + 36u + d.l_name + 4u * (uint32_t)d.n_cigar;
            int32_t rp = d.pos;
            uint32_t qp = 0;
            bool live = k2;
            const uint64_t lt = (1ull << lane) - 1ull;
            for (uint32_t k = 0; __any(live && k < d.n_cigar); ++k) {
                const bool act = live && k < d.n_cigar;
                const uint32_t op = act ? ld32u(cig + 4 * k) : 0u;
                const uint32_t ty = (kCigarType >> ((op & 15u) * 2u)) & 3u;
                uint32_t len = op >> 4;
                if (ty & 2u) {
                    if (len == 0) len = 1;
                    const int32_t room = d.end - rp;
                    if ((int64_t)len > (int64_t)room) len = (uint32_t)(room > 0 ? room : 0);
                }
                // clipped to the tile
                const int32_t c0 = ts > rp ? ts - rp : 0;
                const int64_t room_t = (int64_t)te - rp;
                int32_t c1 = (int64_t)len < room_t ? (int32_t)len : (int32_t)(room_t < 0 ? 0 : room_t);
                const bool is_m = act && ty == 3u;
                if (is_m && (int64_t)qp + (int64_t)c1 > (int64_t)d.l_seq) c1 = (int32_t)d.l_seq - (int32_t)qp;   // malformed record guard
                const bool has_m = is_m && c1 > c0, has_g = act && ty == 2u && c1 > c0;
                const uint64_t mb = __ballot(has_m), gb = __ballot(has_g);
                if (has_m) {
                    const uint32_t slot_q = nq + (uint32_t)__popcll(mb & lt);
                    const uint32_t rt0 = (uint32_t)(rp + c0 - ts), rn = (uint32_t)(c1 - c0), rq0 = qp + (uint32_t)c0;
                    const uint64_t a64r = seq_off + (rq0 >> 1);
                    if (slot_q < kRunItems) runq[slot_q] = u32x4c{(uint32_t)a64r, (uint32_t)(a64r >> 32) | ((rq0 & 1u) << 31), rt0 | (rn << 16), 0u};
                }
                if (has_g) {
                    const uint32_t slot_g = ng + (uint32_t)__popcll(gb & lt);
                    if (slot_g < kGapItems) {
                        gapq[2 * slot_g] = (uint32_t)(rp + c0 - ts) | ((uint32_t)(c1 - c0) << 16);
                        gapq[2 * slot_g + 1] = (op & 15u) == 2u ? 5u : 6u;      // D -> DEL, otherwise (N) -> REFSKIP
                    }
                }
                nq += (uint32_t)__popcll(mb);
                ng += (uint32_t)__popcll(gb);
                if (act) {
                    if (ty & 2u) rp += (int32_t)len;
                    if (ty == 3u || ty == 1u) qp += len;
                    if (rp >= te || rp >= d.end) live = false;
                }
            }
            overflow = nq > kRunItems || ng > kGapItems;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (!overflow && nq) {
                // blocks of the items: lane q owns item q
                u32x4c it = u32x4c{0, 0, 0, 0};
                if (lane < nq) it = runq[lane];
                const uint32_t it0 = it.z & 0xFFFFu, itn = it.z >> 16;
                const uint32_t inb = lane < nq ? ((it0 + itn - 1u) >> 4) - (it0 >> 4) + 1u : 0u;
                uint32_t inc2 = inb;
#pragma unroll
                for (int dlt = 1; dlt < 64; dlt <<= 1) {
                    const uint32_t o = __shfl_up(inc2, dlt, 64);
                    if ((int)lane >= dlt) inc2 += o;
                }
                qblocks = __builtin_amdgcn_readlane(inc2, 63);
                if (qblocks > kRunBlocks) overflow = true;
                else {
                    const uint32_t st2 = inc2 - inb;
                    if (lane < nq) runq[lane].w = st2;
                    for (uint32_t j = 0; __any(j < inb); ++j)
                        if (j < inb) map2[st2 + j] = (uint8_t)lane;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    for (uint32_t g0 = 0; g0 < qblocks; g0 += 64u) {
                        const uint32_t g = g0 + lane;
                        if (g < qblocks) {
                            const u32x4c e = runq[map2[g]];
                            const uint32_t rt0 = e.z & 0xFFFFu, rn = e.z >> 16;
                            const uint32_t pb = ((rt0 >> 4) + (g - e.w)) << 4;
                            const uint32_t k0 = pb < rt0 ? rt0 - pb : 0u;
                            const uint32_t k1 = rt0 + rn - pb < 16u ? rt0 + rn - pb : 16u;
                            const uint8_t* seq0 = U + (((uint64_t)(e.y & 0xFFFFu) << 32) | e.x);
                            uint32_t b_lo, b_hi;
                            load_block(seq0, e.y >> 31, rt0, pb, &b_lo, &b_hi);
                            add_block(cnt, pb, ((1u << k1) - 1u) & ~((1u << k0) - 1u), b_lo, b_hi);
                        }
                    }
                }
            }
            if (!overflow) {
                for (uint32_t gi = 0; gi < ng; ++gi) {
                    const uint32_t w0 = gapq[2 * gi], code = gapq[2 * gi + 1];
                    const uint32_t g0p = w0 & 0xFFFFu, gn = w0 >> 16;
                    for (uint32_t i = lane; i < gn; i += 64u)
                        atomicAdd(cnt + blk_dw(g0p + i) + (code >> 1), 1u << ((code & 1u) << 4));
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        // ---- whatever is left (CIGARs of more than kLaneOps operations, single runs of more than eleven blocks, batches that
        // overflowed the queues): one read at a time, 64 lanes ------------------------------------------------------------------
        uint64_t m2 = __ballot(take && !fast && ((d.kind == 2 && (!k2 || overflow)) || (d.kind != 2 && n_run != 0)));
        while (m2) {
            const int r = __builtin_ctzll(m2);
            m2 &= m2 - 1;
            const uint32_t off_lo = __builtin_amdgcn_readlane((uint32_t)d.rec_off, r);
            const uint32_t off_hi = __builtin_amdgcn_readlane((uint32_t)(d.rec_off >> 32), r);
            const int32_t R_pos = (int32_t)__builtin_amdgcn_readlane((uint32_t)d.pos, r);
            const int32_t R_end = (int32_t)__builtin_amdgcn_readlane((uint32_t)d.end, r);
            const uint32_t R_lseq = __builtin_amdgcn_readlane(d.l_seq, r);
            const uint32_t misc = __builtin_amdgcn_readlane((uint32_t)d.n_cigar | ((uint32_t)d.l_name << 16), r);
            const uint32_t R_ncig = misc & 0xFFFFu;
            const uint8_t* rec = U + (((uint64_t)off_hi << 32) | off_lo);
            const uint8_t* cig = rec + 36 + ((misc >> 16) & 0xFFu);
            const uint8_t* seq = cig + 4 * R_ncig;
            // CIGAR walk as in k_accumulate: a reference-consuming op occupies max(len, 1) columns (pileup.d:195-205) and the
            // read leaves the pileup at end = pos + sum(len) (read.d:1380-1383)
            int32_t rp = R_pos;
            uint32_t qp = 0;
            for (uint32_t k = 0; k < R_ncig; ++k) {
                const uint32_t op = ld32u(cig + 4 * k);
                const uint32_t ty = (kCigarType >> ((op & 15u) * 2u)) & 3u;
                uint32_t len = op >> 4;
                if (ty & 2u) {
                    if (len == 0) len = 1;
                    const int32_t room = R_end - rp;
                    if ((int64_t)len > (int64_t)room) len = (uint32_t)(room > 0 ? room : 0);
                }
                if (ty == 3) {
                    int32_t a0 = ts > rp ? ts - rp : 0;
                    int32_t a1 = (int32_t)len < te - rp ? (int32_t)len : te - rp;
                    if ((int64_t)qp + (int64_t)a1 > (int64_t)R_lseq) a1 = (int32_t)R_lseq - (int32_t)qp;   // malformed record guard
                    if (a1 > a0) {
                        const uint32_t rt0 = (uint32_t)(rp + a0 - ts), rn = (uint32_t)(a1 - a0), rq0 = qp + (uint32_t)a0;
                        const uint8_t* seq0 = seq + (rq0 >> 1);
                        const uint32_t nb = ((rt0 + rn - 1u) >> 4) - (rt0 >> 4) + 1u;
                        for (uint32_t j = lane; j < nb; j += 64u) {
                            const uint32_t pb = ((rt0 >> 4) + j) << 4;
                            const uint32_t k0 = pb < rt0 ? rt0 - pb : 0u;
                            const uint32_t k1 = rt0 + rn - pb < 16u ? rt0 + rn - pb : 16u;
                            uint32_t b_lo, b_hi;
                            load_block(seq0, rq0 & 1u, rt0, pb, &b_lo, &b_hi);
                            add_block(cnt, pb, ((1u << k1) - 1u) & ~((1u << k0) - 1u), b_lo, b_hi);
                        }
                    }
                    rp += (int32_t)len;
                    qp += len;
                } else if (ty == 2) {
                    const uint32_t code = (op & 15u) == 2u ? 5u : 6u;      // D -> DEL, otherwise (N) -> REFSKIP
                    const int32_t g0 = ts > rp ? ts - rp : 0;
                    const int64_t room = (int64_t)te - rp;
                    const int32_t g1 = (int64_t)len < room ? (int32_t)len : (int32_t)(room < 0 ? 0 : room);
                    for (int32_t i = g0 + (int32_t)lane; i < g1; i += 64)
                        atomicAdd(cnt + blk_dw((uint32_t)(rp + i - ts)) + (code >> 1), 1u << ((code & 1u) << 4));
                    rp += (int32_t)len;
                } else if (ty == 1) {
                    qp += len;
                }
                if (rp >= te || rp >= R_end) break;
            }
        }
    }
    __syncthreads();
    // ---- write the tile once: 16-bit pairs -> u32[T][7], coalesced 16-byte stores -------------------------------------------------
    const uint32_t n_out = T * 7u;                  // multiple of 4 (T >= 16)
    if (compact) {       // region / window modes: {bases counted : 16 | depth : 16} per position (see k_accumulate16)
        uint32_t* outc = counters + (size_t)slot * T;
        for (uint32_t i4 = threadIdx.x * 4u; i4 < T; i4 += kAccThreads * 4u) {
            uint32_t v[4];
#pragma unroll
            for (uint32_t e = 0; e < 4; ++e) {
                const uint32_t* w = cnt + blk_dw(i4 + e);
                const uint32_t m = (w[0] & 0xFFFFu) + (w[0] >> 16) + (w[1] & 0xFFFFu) + (w[1] >> 16) + (w[2] & 0xFFFFu);
                v[e] = m | ((m + (w[2] >> 16) + (w[3] & 0xFFFFu)) << 16);
            }
            *(uint4*)(outc + i4) = make_uint4(v[0], v[1], v[2], v[3]);
        }
        return;
    }
    uint32_t* out = counters + (size_t)slot * n_out;
    const uint32_t inv_7 = 0xFFFFFFFFu / 7u + 1u;
    for (uint32_t i4 = threadIdx.x * 4u; i4 < n_out; i4 += kAccThreads * 4u) {
        uint32_t v[4];
#pragma unroll
        for (uint32_t e = 0; e < 4; ++e) {
            const uint32_t i = i4 + e;
            const uint32_t p = __umulhi(i, inv_7), k = i - p * 7u;
            const uint32_t w = cnt[blk_dw(p) + (k >> 1)];
            v[e] = (k & 1u) ? (w >> 16) : (w & 0xFFFFu);
        }
        *(uint4*)(out + i4) = make_uint4(v[0], v[1], v[2], v[3]);
    }
}

}  // namespace

void launch_accumulate(const uint8_t* d_U, const RecDesc* d_desc, const uint32_t* d_tile_lo, const uint32_t* d_tile_hi,
                       const uint32_t* d_active, uint32_t n_active, uint32_t n_deep, uint32_t deep_thr, const uint32_t* d_tile_base,
                       int32_t n_ref, uint32_t tile_pos, uint32_t n_samples, uint32_t min_bq, uint32_t* d_counters, uint32_t* d_span,
                       hipStream_t stream, bool compact) {
    if (!n_active) return;
    if (compact && n_deep) throw Error(SBX_EINVAL, "internal error: compact counters with deep tiles");
    const uint32_t cflag = compact ? 1u : 0u;
    {
        const size_t lds = (size_t)tile_pos * n_samples * 16 + (d_span ? ((size_t)tile_pos + 5) * 4 : 0);
        const dim3 grid(((n_active + 7) / 8) * 8), block(kAccThreads);
        auto go = [&](auto kern) {
            SBX_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(kern, grid, block, lds, stream, d_U, d_desc, d_tile_lo, d_tile_hi, d_active, n_active, d_tile_base, n_ref,
                               tile_pos, n_samples, min_bq, deep_thr, d_counters, d_span, cflag);
        };
        if (n_samples == 1 && !d_span && !min_bq) {
            const size_t lds3 = (((size_t)4 * tile_pos + (tile_pos >> 4) + 3) & ~(size_t)3) * 4 + (size_t)(kAccThreads / 64) * kWaveBytesC;
            SBX_HIP(hipFuncSetAttribute((const void*)k_accumulate16c, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3));
            hipLaunchKernelGGL(k_accumulate16c, grid, block, lds3, stream, d_U, d_desc, d_tile_lo, d_tile_hi, d_active, n_active, d_tile_base,
                               n_ref, tile_pos, deep_thr, d_counters, cflag);
        } else if (n_samples == 1) {
            if (d_span) { if (min_bq) go(k_accumulate16<true, true, true>); else go(k_accumulate16<true, false, true>); }
            else { if (min_bq) go(k_accumulate16<false, true, true>); else go(k_accumulate16<false, false, true>); }
        } else {
            if (d_span) { if (min_bq) go(k_accumulate16<true, true, false>); else go(k_accumulate16<true, false, false>); }
            else { if (min_bq) go(k_accumulate16<false, true, false>); else go(k_accumulate16<false, false, false>); }
        }
        SBX_HIP(hipGetLastError());
    }
    if (!n_deep) return;
    size_t lds = ((size_t)(tile_pos / 4) * n_samples * 7 + 8) * 16 + (d_span ? (size_t)tile_pos * 4 : 0);
    if (d_span) {
        SBX_HIP(hipFuncSetAttribute((const void*)k_accumulate<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_accumulate<true>, dim3(n_active), dim3(kAccThreads), lds, stream, d_U, d_desc, d_tile_lo,
                           d_tile_hi, d_active, d_tile_base, n_ref, tile_pos, n_samples, min_bq, deep_thr, d_counters, d_span);
    } else {
        SBX_HIP(hipFuncSetAttribute((const void*)k_accumulate<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_accumulate<false>, dim3(n_active), dim3(kAccThreads), lds, stream, d_U, d_desc, d_tile_lo,
                           d_tile_hi, d_active, d_tile_base, n_ref, tile_pos, n_samples, min_bq, deep_thr, d_counters, d_span);
    }
    SBX_HIP(hipGetLastError());
}

}  // namespace sbx
