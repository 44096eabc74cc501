// format.hip -- K6 `format_base_rows`: the text of `sambamba depth base` written on the device.
//
// Replaces PerBasePrinter.writeColumn (sambamba/depth.d:534-555) and the zero-coverage rows of
// writeEmptyColumns (depth.d:452-487) for a position range of one contig:
// without it the end-to-end rate is bounded by the host formatting ~35 bytes for each of 10^8..10^9
// positions.  A row is
//     REF \t POS \t COV \t A \t C \t G \t T \t DEL \t REFSKIP [\t SAMPLE] [\t y|n] \n
// per sample, in sample order; the loop over samples RETURNS at the first sample whose COV is outside
// [min_cov, max_cov] unless rows are annotated (depth.d:540-541).  A position takes part iff a pileup
// column exists there (>= 1 admitted read spans it) or min_cov == 0, in which case positions without a
// column print all-zero rows -- the same text writeEmptyColumns produces from its precomputed tails.
//
// Two passes over the counter tiles with the same emitter: pass 1 only adds up the bytes of every
// 256-position chunk, a scan turns them into offsets, pass 2 builds each chunk's text in LDS (one
// position per lane; rows assembled in registers and stored eight bytes at a time, format_core.hpp)
// and streams it to HBM 16 bytes per lane, contiguous.  HBM-bound: 28 B of counters read twice +
// ~35 B of text written per position and sample.
#include "common.hpp"
#include "format_core.hpp"
#include "kernels.hpp"

namespace sbx {

namespace {

constexpr int kFmtThreads = 256;

// The counters of a position: where they are and whether a pileup column exists there (>= 1 admitted read spans it).
struct PosInfo {
    const uint32_t* c;
    bool column, prints;          // prints: the position has rows at all (a column, or all-zero rows when min_cov == 0)
};
__device__ __forceinline__ PosInfo pos_lookup(const FormatArgs& a, uint32_t pos) {
    const uint32_t tile = a.tile_first + pos / a.T;
    const uint32_t slot = tile < a.tile_end ? a.slot_of[tile] : 0xFFFFFFFFu;
    const uint32_t in_tile = pos & (a.T - 1u);
    PosInfo pi;
    pi.c = slot != 0xFFFFFFFFu ? a.counters + ((size_t)slot * a.T + in_tile) * a.S * 7u : nullptr;
    pi.column = false;
    if (pi.c) {
        if (a.span) pi.column = a.span[(size_t)slot * a.T + in_tile] != 0;
        else {
            uint32_t any = 0;
            for (uint32_t k = 0; k < a.S * 7u; ++k) any |= pi.c[k];
            pi.column = any != 0;
        }
    }
    pi.prints = pi.column || a.zero_fill;
    return pi;
}

// One row -- REF \t POS \t COV \t A \t C \t G \t T \t DEL \t REFSKIP [\t SAMPLE] [\t y|n] \n  (the other bases are counted in COV only) --
// in two steps, so that the writing kernel, which needs the length of every row of a chunk before it can place the first one, does not
// read and count a position twice: row_prepare loads the counters of sample s and counts the digits; row_emit appends the row to a
// RowSink (format_core.hpp: rows assembled in a 64-bit register, eight bytes per store, four digits per dword of arithmetic).
struct RowPrep {
    uint32_t w[6], nd[6];
    uint64_t total;
    uint32_t nd_tot, len;
    bool ok, small;
};
// false: the loop over the samples of the position ends here (coverage out of range and rows are not annotated: depth.d:540-541 returns)
__device__ __forceinline__ bool row_prepare(const FormatArgs& a, const PosInfo& pi, uint32_t s, uint32_t nd_pos, RowPrep& r) {
    uint32_t v[7] = {0, 0, 0, 0, 0, 0, 0};
    if (pi.column) __builtin_memcpy(v, pi.c + s * 7u, 28);
    r.total = (uint64_t)v[0] + v[1] + v[2] + v[3] + v[4] + v[5] + v[6];
    r.ok = r.total >= a.lo && r.total <= a.hi;
    r.len = 0;
    if (!r.ok && !a.annotate) return false;
    r.small = r.total < 10000ull;                  // then every counter has at most four digits
    r.w[0] = v[0]; r.w[1] = v[1]; r.w[2] = v[2]; r.w[3] = v[3]; r.w[4] = v[5]; r.w[5] = v[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) r.nd[k] = r.small ? fmt::n_digits4(r.w[k]) : fmt::n_digits32(r.w[k]);
    r.nd_tot = r.small ? fmt::n_digits4((uint32_t)r.total) : fmt::n_digits64(r.total);
    const uint32_t sl = a.combined ? 0u : a.sample_off[s + 1] - a.sample_off[s];
    r.len = a.ref_name_len + 1u + nd_pos + 1u + r.nd_tot + 6u + r.nd[0] + r.nd[1] + r.nd[2] + r.nd[3] + r.nd[4] + r.nd[5] +
            (a.combined ? 0u : 1u + sl) + (a.annotate ? 2u : 0u) + 1u;
    return true;
}
__device__ __forceinline__ void row_emit(const FormatArgs& a, uint32_t pos, uint32_t nd_pos, uint32_t s, const RowPrep& r, fmt::RowSink& o) {
    o.str(a.names, a.ref_name_len);
    o.sep_num32('\t', pos, nd_pos);
    if (r.small) o.sep_num32('\t', (uint32_t)r.total, r.nd_tot); else o.sep_num64('\t', r.total);
#pragma unroll
    for (int k = 0; k < 6; ++k) o.sep_num32('\t', r.w[k], r.nd[k]);
    if (!a.combined) {
        o.put((uint64_t)'\t', 1u);
        o.str(a.names + a.sample_off[s], a.sample_off[s + 1] - a.sample_off[s]);
    }
    if (a.annotate) o.put((uint64_t)'\t' | (uint64_t)(r.ok ? 'y' : 'n') << 8 | (uint64_t)'\n' << 16, 3u);
    else o.put((uint64_t)'\n', 1u);
}

// One position: its rows (one per sample) appended to `dst` when kWrite, their length either way.
template <bool kWrite>
__device__ __forceinline__ uint32_t emit_position(const FormatArgs& a, uint32_t pos, uint8_t* dst) {
    const PosInfo pi = pos_lookup(a, pos);
    if (!pi.prints) return 0;
    fmt::RowSink o;
    o.init(dst);
    uint32_t n = 0;
    const uint32_t nd_pos = fmt::n_digits32(pos);
    for (uint32_t s = 0; s < a.S; ++s) {
        RowPrep r;
        if (!row_prepare(a, pi, s, nd_pos, r)) break;
        n += r.len;
        if (kWrite) row_emit(a, pos, nd_pos, s, r, o);
    }
    if (kWrite) o.finish();
    return n;
}

__device__ __forceinline__ uint32_t wave_incl(uint32_t v, uint32_t lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(v, d, 64);
        if ((int)lane >= d) v += t;
    }
    return v;
}

// pass 1: bytes of every chunk of kFmtThreads positions
__global__ __launch_bounds__(kFmtThreads) void k_format_measure(FormatArgs a, uint32_t* __restrict__ chunk_len) {
    __shared__ uint32_t wsum[kFmtThreads / 64];
    const uint32_t t = threadIdx.x, lane = t & 63u, wv = t >> 6;
    const uint64_t pos = (uint64_t)a.beg + (uint64_t)blockIdx.x * kFmtThreads + t;
    const uint32_t n = pos < a.end ? emit_position<false>(a, (uint32_t)pos, nullptr) : 0u;
    const uint32_t inc = wave_incl(n, lane);
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    if (t == 0) chunk_len[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// pass 2: the text
__global__ __launch_bounds__(kFmtThreads) void k_format_write(FormatArgs a, const uint64_t* __restrict__ chunk_off,
                                                             uint8_t* __restrict__ text, uint32_t lds_cap) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_text[];
    __shared__ uint32_t wsum[kFmtThreads / 64];
    const uint32_t t = threadIdx.x, lane = t & 63u, wv = t >> 6;
    const uint64_t pos = (uint64_t)a.beg + (uint64_t)blockIdx.x * kFmtThreads + t;
    const uint64_t off = chunk_off[blockIdx.x];
    const uint32_t total = (uint32_t)(chunk_off[blockIdx.x + 1] - off);
    if (total == 0) return;
    // (one sample: the row is prepared once -- counters read, digits counted -- and written from the registers)
    const bool one = a.S == 1u;
    RowPrep r1;
    r1.len = 0;
    uint32_t nd_pos = 0;
    uint32_t n = 0;
    if (pos < a.end) {
        if (one) {
            const PosInfo pi = pos_lookup(a, (uint32_t)pos);
            nd_pos = fmt::n_digits32((uint32_t)pos);
            if (pi.prints && row_prepare(a, pi, 0, nd_pos, r1)) n = r1.len;
        } else n = emit_position<false>(a, (uint32_t)pos, nullptr);
    }
    const uint32_t inc = wave_incl(n, lane);
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    uint32_t my = inc - n;
    for (uint32_t w = 0; w < wv; ++w) my += wsum[w];
    uint8_t* out = text + off;
    const bool in_lds = total <= lds_cap;         // (else very long rows: straight to HBM)
    // (two calls: the stores into LDS stay LDS instructions)
    auto rows_to = [&](uint8_t* dst) {
        if (one) {
            fmt::RowSink o;
            o.init(dst);
            row_emit(a, (uint32_t)pos, nd_pos, 0, r1, o);
            o.finish();
        } else emit_position<true>(a, (uint32_t)pos, dst);
    };
    if (n && in_lds) rows_to(lds_text + my);
    if (n && !in_lds) rows_to(out + my);
    if (!in_lds) return;
    __syncthreads();
    for (uint32_t i = 16 * t; i < total; i += 16 * kFmtThreads) {
        if (i + 16 <= total) {
            const uint4 v = *(const uint4*)(lds_text + i);
            __builtin_memcpy(out + i, &v, 16);
        } else {
            for (uint32_t k = i; k < total; ++k) out[k] = lds_text[k];
        }
    }
}

}  // namespace

uint32_t format_chunk_positions() { return kFmtThreads; }

void launch_format_measure(const FormatArgs& a, uint32_t n_chunks, uint32_t* d_chunk_len, hipStream_t stream) {
    if (!n_chunks) return;
    hipLaunchKernelGGL(k_format_measure, dim3(n_chunks), dim3(kFmtThreads), 0, stream, a, d_chunk_len);
    SBX_HIP(hipGetLastError());
}

void launch_format_write(const FormatArgs& a, uint32_t n_chunks, const uint64_t* d_chunk_off, uint8_t* d_text, hipStream_t stream) {
    if (!n_chunks) return;
    // LDS for the text of a chunk: what rows with three-digit counters need -- 13 KB for one sample and a short contig name, eight
    // workgroups per CU -- instead of the 48 KB a chunk may use; a chunk that does not fit writes its rows straight to HBM
    const uint32_t row = a.ref_name_len + 44u + (a.combined ? 0u : a.max_sample_len + 1u) + (a.annotate ? 2u : 0u);
    uint64_t want = (uint64_t)kFmtThreads * a.S * row;
    want = (want + 1023u) & ~1023ull;
    const uint32_t lds = (uint32_t)(want < 8192u ? 8192u : want > 48u * 1024u ? 48u * 1024u : want);
    hipLaunchKernelGGL(k_format_write, dim3(n_chunks), dim3(kFmtThreads), lds, stream, a, d_chunk_off, d_text, lds);
    SBX_HIP(hipGetLastError());
}

}  // namespace sbx
