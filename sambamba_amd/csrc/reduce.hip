// reduce.hip -- K5: per-region / per-window statistics of `sambamba depth region|window`.
//
// Replaces PerRegionPrinter.push / countRead / countOverlappingBases and the per-column threshold
// bookkeeping (sambamba/depth.d:661-698,760-845) plus the stats collectors (depth.d:107-227).
// Without -m these are closed forms over the per-position counters of K3 (SURVEY.md Appendix A,
// verified against a literal restatement):
//     n_bases[r]      = sum_{p in r} (# M/=/X bases at p with qual >= min_bq)      = sum of codes 0..4
//     cov_count[r][t] = #{p in r : COV(p) >= T_t},  COV = all 7 counters (D/N count as quality 255)
//     n_reads[r]      = # admitted reads having >= 1 M/=/X base with qual >= min_bq inside r
// `range_reduce` does the first two as segmented reductions over the counter tiles (one wave per
// range chunk, coalesced 28-byte rows); `count_reads` does the third per record (regions located by
// binary search in the (ref,start)-sorted list with a prefix maximum of ends, so overlapping and
// duplicate BED lines each get their own count, as GeneralRegionStatsCollector gives them).
// Counters are 32-bit and wrap like the reference's `uint` fields (depth.d:618-620).
#include "common.hpp"
#include "kernels.hpp"

namespace sbx {

namespace {

constexpr int kRedThreads = 256;
constexpr uint32_t kCigarType = 0x3C1A7u;

__device__ __forceinline__ uint32_t ld32r(const uint8_t* p) {
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_down(v, d, 64);
    return v;
}

__global__ __launch_bounds__(kRedThreads) void k_range_reduce(
    const RangeChunk* __restrict__ chunks, uint32_t n_chunks, const uint32_t* __restrict__ counters,
    const uint32_t* __restrict__ span, const uint32_t* __restrict__ slot_of, const uint32_t* __restrict__ tile_base,
    uint32_t T, uint32_t S, const uint32_t* __restrict__ thresholds, uint32_t n_thr, uint32_t* n_bases /*[id][S]*/,
    uint32_t* cov_counts /*[id][S][n_thr]*/, uint32_t* seen /*[id]*/, uint32_t compact,
    const uint64_t* __restrict__ win_base, const uint64_t* __restrict__ n_win, uint32_t n_ref, uint32_t window) {
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t ci = blockIdx.x * (kRedThreads / 64) + wv;
    if (ci >= n_chunks) return;
    RangeChunk ch;
    if (chunks) ch = chunks[ci];
    else {
        // window mode without a chunk list (round 5): chunk ci IS window ci of the run -- window k of contig r has id win_base[r] + k
        // (win_base = running sum of the contigs' window counts), so the contig is the last one whose base is <= ci
        uint32_t lo = 0, hi = n_ref;              // invariant: win_base[lo] <= ci
        while (hi - lo > 1) { const uint32_t m = (lo + hi) >> 1; if (win_base[m] <= ci) lo = m; else hi = m; }
        const uint64_t k = ci - win_base[lo];
        if (k >= n_win[lo]) return;
        ch.ref_id = lo;
        ch.start = (uint32_t)(k * window);
        ch.end = ch.start + window;
        ch.id = ci;
    }
    const bool no_bases = (ch.id >> 30) & 1u;     // n_bases of this range is gathered per read (count_reads with min_start)
    ch.id &= 0x3FFFFFFFu;
    const uint32_t tb = tile_base[ch.ref_id];
    const uint32_t t_end = tile_base[ch.ref_id + 1];
    const uint32_t row = S * 7;
    uint32_t any = 0;
    for (uint32_t s = 0; s < S; ++s) {
        uint32_t nb = 0;
        uint32_t cc[kMaxThresholds];
#pragma unroll
        for (int t = 0; t < kMaxThresholds; ++t) cc[t] = 0;
        for (uint32_t p = ch.start + lane; p < ch.end; p += 64) {
            const uint32_t tile = tb + p / T;
            if (tile >= t_end) continue;
            const uint32_t slot = slot_of[tile];
            if (slot == 0xFFFFFFFFu) continue;
            uint32_t m, cov;
            if (compact) {       // one word per position and sample: {bases counted : 16 | depth : 16} (launch_accumulate, compact)
                const uint32_t w = counters[((size_t)slot * T + (p & (T - 1))) * S + s];
                m = w & 0xFFFFu;
                cov = w >> 16;
            } else {
                const uint32_t* c = counters + ((size_t)slot * T + (p & (T - 1))) * row + s * 7;
                m = c[0] + c[1] + c[2] + c[3] + c[4];
                cov = m + c[5] + c[6];
            }
            nb += m;
            if (span) any |= span[(size_t)slot * T + (p & (T - 1))];
            else any |= cov;
#pragma unroll
            for (int t = 0; t < kMaxThresholds; ++t)
                if ((uint32_t)t < n_thr && cov >= thresholds[t]) cc[t] += 1;
        }
        nb = wave_sum(nb);
        if (lane == 0 && nb && !no_bases) atomicAdd(&n_bases[(size_t)ch.id * S + s], nb);
#pragma unroll
        for (int t = 0; t < kMaxThresholds; ++t) {
            if ((uint32_t)t < n_thr) {
                uint32_t v = wave_sum(cc[t]);
                if (lane == 0 && v) atomicAdd(&cov_counts[((size_t)ch.id * S + s) * n_thr + t], v);
            }
        }
    }
    const uint64_t am = __ballot(any != 0);
    if (lane == 0 && am) seen[ch.id] = 1;
}

__device__ uint32_t count_good(const uint8_t* U, const RecDesc& d, uint32_t min_bq, int64_t lo, int64_t hi);

// does the read have an M/=/X base with quality >= min_bq at a reference position inside [rs, re)?
// (countOverlappingBases > 0, depth.d:671-698; zero-length reference-consuming ops occupy one column,
// and the read ends at d.end, exactly as in K3)
__device__ bool has_good_base(const uint8_t* U, const RecDesc& d, uint32_t min_bq, int64_t rs, int64_t re) {
    const uint8_t* rec = U + d.rec_off;
    const uint8_t* cig = rec + 36 + d.l_name;
    const uint8_t* seq = cig + 4 * (uint32_t)d.n_cigar;
    const uint8_t* qual = seq + ((d.l_seq + 1) >> 1);
    auto run = [&](int64_t rp, uint32_t qp, uint32_t len) -> bool {
        int64_t a = rs > rp ? rs : rp, b = re < rp + (int64_t)len ? re : rp + (int64_t)len;
        if (a >= b) return false;
        uint32_t q0 = qp + (uint32_t)(a - rp), q1 = qp + (uint32_t)(b - rp);
        if (q1 > d.l_seq) q1 = d.l_seq;
        if (q0 >= q1) return false;
        if (min_bq == 0) return true;
        for (uint32_t q = q0; q < q1; ++q)
            if (qual[q] >= min_bq) return true;
        return false;
    };
    if (d.kind == 1) return run(d.pos, d.q_start, (uint32_t)(d.end - d.pos));
    int64_t rp = d.pos;
    uint32_t qp = 0;
    for (uint32_t k = 0; k < d.n_cigar; ++k) {
        uint32_t op = ld32r(cig + 4 * k);
        uint32_t ty = (kCigarType >> ((op & 15u) * 2u)) & 3u, len = op >> 4;
        if (ty & 2u) {
            if (len == 0) len = 1;
            int64_t room = (int64_t)d.end - rp;
            if ((int64_t)len > room) len = (uint32_t)(room > 0 ? room : 0);
        }
        if (ty == 3) {
            if (run(rp, qp, len)) return true;
            rp += len;
            qp += len;
        } else if (ty == 2) {
            rp += len;
        } else if (ty == 1) {
            qp += len;
        }
        if (rp >= re || rp >= d.end) break;
    }
    return false;
}

// window mode: windows [k*w, (k+1)*w), k < n_win[ref]; id = win_base[ref] + k
// One lane per record.  The records are sorted by position, so the 64 records of a wavefront fall into two or three windows and a plain
// atomicAdd per record hammers the same few counters from every lane (617 M records: 73 ms of a whole-genome pass, profiles/round4
// call L).  The lanes of a wavefront therefore add up runs of equal counter addresses among themselves -- leaders by comparing with the
// previous lane, run lengths out of the leaders' ballot -- and only the leader of a run touches memory: a few atomics per wavefront.
// The first window of a record goes this way, and so does the second one of a record that straddles a window edge; whatever is left
// (alignments longer than a window) is counted record by record.
__device__ __forceinline__ void add_runs(uint32_t* base, uint64_t slot, bool on, uint32_t lane) {
    // slot: index of the counter this lane adds 1 to (when on)
    const uint64_t prev = __shfl_up(slot, 1, 64);
    const bool prev_on = __shfl_up(on ? 1 : 0, 1, 64) != 0;
    const bool leader = on && (lane == 0 || !prev_on || prev != slot);
    const uint64_t lead = __ballot(leader), act = __ballot(on);
    if (leader) {
        // the run ends in front of the next leader or the first inactive lane above this one
        const uint64_t above = ~0ull << lane << 1;                    // lanes > lane (lane 63: 0)
        const uint64_t stop = (lead | ~act) & above;
        const uint32_t end = stop ? (uint32_t)__builtin_ctzll(stop) : 64u;
        atomicAdd(base + slot, end - lane);
    }
}
__global__ __launch_bounds__(kRedThreads) void k_count_reads_windows(
    const uint8_t* __restrict__ U, const RecDesc* __restrict__ desc, uint64_t n_records, const int32_t* __restrict__ rec_ref,
    uint32_t window, const uint64_t* __restrict__ win_base, const uint64_t* __restrict__ n_win, uint32_t S, uint32_t min_bq,
    uint32_t* n_reads /*[id][S]*/) {
    const uint64_t i = (uint64_t)blockIdx.x * kRedThreads + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    const bool live = i < n_records;
    RecDesc d{};
    if (live) d = desc[i];
    const bool adm = live && d.kind != 0;
    const int32_t ref = adm ? rec_ref[i] : 0;
    const uint64_t k0 = adm ? (uint64_t)d.pos / window : 0, k1 = adm ? (uint64_t)(d.end - 1) / window : 0;
    const uint64_t nw = adm ? n_win[ref] : 0, wb = adm ? win_base[ref] : 0;
    const uint32_t s = S > 1 ? d.sample : 0u;
    // first window, second window: wave-aggregated
    const bool on0 = adm && k0 < nw && has_good_base(U, d, min_bq, (int64_t)(k0 * window), (int64_t)((k0 + 1) * window));
    add_runs(n_reads, (wb + k0) * S + s, on0, lane);
    const bool on1 = adm && k1 > k0 && k0 + 1 < nw && has_good_base(U, d, min_bq, (int64_t)((k0 + 1) * window), (int64_t)((k0 + 2) * window));
    if (__any(adm && k1 > k0)) add_runs(n_reads, (wb + k0 + 1) * S + s, on1, lane);
    // further windows of a long alignment
    if (adm)
        for (uint64_t k = k0 + 2; k <= k1 && k < nw; ++k)
            if (has_good_base(U, d, min_bq, (int64_t)(k * window), (int64_t)((k + 1) * window)))
                atomicAdd(&n_reads[(size_t)(wb + k) * S + s], 1u);
}

// region mode: regions sorted by (ref, start); pmax_end[j] = max end over the contig's regions 0..j
__global__ __launch_bounds__(kRedThreads) void k_count_reads_regions(
    const uint8_t* __restrict__ U, const RecDesc* __restrict__ desc, uint64_t n_records, const int32_t* __restrict__ rec_ref,
    const SortedRegion* __restrict__ regs, const uint32_t* __restrict__ pmax_end, const uint32_t* __restrict__ ref_first /*[n_ref+1]*/,
    uint32_t S, uint32_t min_bq, uint32_t* n_reads /*[id][S]*/, const uint32_t* __restrict__ min_start /*[id] or nullptr*/,
    uint32_t* n_bases /*[id][S], for ranges with min_start != 0*/) {
    const uint64_t i = (uint64_t)blockIdx.x * kRedThreads + threadIdx.x;
    if (i >= n_records) return;
    const RecDesc d = desc[i];
    if (d.kind == 0) return;
    const int32_t ref = rec_ref[i];
    const uint32_t lo0 = ref_first[ref], hi0 = ref_first[ref + 1];
    if (lo0 >= hi0) return;
    // last region with start < d.end
    uint32_t a = lo0, c = hi0;
    while (a < c) { uint32_t m = (a + c) >> 1; if ((int64_t)regs[m].start < (int64_t)d.end) a = m + 1; else c = m; }
    const uint32_t s = S > 1 ? d.sample : 0u;
    for (uint32_t j = a; j > lo0;) {
        --j;
        if ((int64_t)pmax_end[j] <= (int64_t)d.pos) break;      // nothing at or before j reaches the read
        if ((int64_t)regs[j].end <= (int64_t)d.pos) continue;
        const uint32_t ms = min_start ? min_start[regs[j].id] : 0u;
        if (ms) {
            // a window that only counts the reads starting at or after `ms` (the reference's first ring of overlapping
            // windows, depth.d:1031-1032): both numbers come from the reads themselves
            if ((int64_t)d.pos < (int64_t)ms) continue;
            const uint32_t B = count_good(U, d, min_bq, (int64_t)regs[j].start, (int64_t)regs[j].end);
            if (B) { atomicAdd(&n_reads[(size_t)regs[j].id * S + s], 1u); atomicAdd(&n_bases[(size_t)regs[j].id * S + s], B); }
        } else if (has_good_base(U, d, min_bq, (int64_t)regs[j].start, (int64_t)regs[j].end)) {
            atomicAdd(&n_reads[(size_t)regs[j].id * S + s], 1u);
        }
    }
}

// ---- region / window statistics with --fix-mate-overlaps -------------------------------------------------------
// Closed form of PerRegionPrinter.push with mate fixing (depth.d:717-845), derived from a literal restatement of
// the reference's status machine and checked against it (tests/test_gpu_mates.py).  For a region R, with F = the first pileup column inside
// R, a pair's overlap O = [oa, ob) and f = the first column of O that lies inside ANY region (there the pair is
// "fixed", depth.d:751-758):
//   n_bases[R] = sum over reads counted for R of B(r, R)                      (countRead, depth.d:661-669)
//              - for pairs with f in R: B(r1, R from f on) + B(r2, R from f on)   (uncountOverlappingMates, :717-743)
//              + sum over columns of R of addm                                 (process_base, :802-808 -- k_mates_columns)
//   a read is counted for R iff it spans F and is not "fixed" there (f < F, F inside O), or starts inside R after F;
//   n_reads[R] = sum over counted reads of [B > 0], pairs with f in R merged into one read ([B1>0] + [B2>0] ->
//                [B1+B2 > 0]), + 1 for every pair that is already fixed at F and has a base in R (:779-796);
//   cov_count[R][t] = #columns of R with covm >= T_t.
// B(r, [lo, hi)) = number of M/=/X bases of r at reference positions in [lo, hi) with quality >= min_bq.

// number of M/=/X bases of the read at reference positions in [lo, hi) with quality >= min_bq (countOverlappingBases,
// depth.d:671-698; zero-length reference-consuming ops occupy one column and the read ends at d.end, as in K3)
__device__ uint32_t count_good(const uint8_t* U, const RecDesc& d, uint32_t min_bq, int64_t lo, int64_t hi) {
    const uint8_t* rec = U + d.rec_off;
    const uint8_t* cig = rec + 36 + d.l_name;
    const uint8_t* seq = cig + 4 * (uint32_t)d.n_cigar;
    const uint8_t* qual = seq + ((d.l_seq + 1) >> 1);
    uint32_t n = 0;
    auto run = [&](int64_t rp, uint32_t qp, uint32_t len) {
        int64_t a = lo > rp ? lo : rp, b = hi < rp + (int64_t)len ? hi : rp + (int64_t)len;
        if (a >= b) return;
        uint32_t q0 = qp + (uint32_t)(a - rp), q1 = qp + (uint32_t)(b - rp);
        if (q1 > d.l_seq) q1 = d.l_seq;
        for (uint32_t q = q0; q < q1; ++q) n += qual[q] >= min_bq ? 1u : 0u;
    };
    if (lo >= hi) return 0;
    if (d.kind == 1) { run(d.pos, d.q_start, (uint32_t)(d.end - d.pos)); return n; }
    int64_t rp = d.pos;
    uint32_t qp = 0;
    for (uint32_t k = 0; k < d.n_cigar; ++k) {
        uint32_t op = ld32r(cig + 4 * k);
        uint32_t ty = (kCigarType >> ((op & 15u) * 2u)) & 3u, len = op >> 4;
        if (ty & 2u) {
            if (len == 0) len = 1;
            int64_t room = (int64_t)d.end - rp;
            if ((int64_t)len > room) len = (uint32_t)(room > 0 ? room : 0);
        }
        if (ty == 3) {
            run(rp, qp, len);
            rp += len;
            qp += len;
        } else if (ty == 2) {
            rp += len;
        } else if (ty == 1) {
            qp += len;
        }
        if (rp >= hi || rp >= d.end) break;
    }
    return n;
}

// first pileup column of every range: atomicMin over the positions with span > 0
__global__ __launch_bounds__(kRedThreads) void k_range_first(const RangeChunk* __restrict__ chunks, uint32_t n_chunks,
                                                             const uint32_t* __restrict__ span, const uint32_t* __restrict__ slot_of,
                                                             const uint32_t* __restrict__ tile_base, uint32_t T, uint32_t* first /*[id]*/) {
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t ci = blockIdx.x * (kRedThreads / 64) + wv;
    if (ci >= n_chunks) return;
    const RangeChunk ch = chunks[ci];
    const uint32_t tb = tile_base[ch.ref_id], t_end = tile_base[ch.ref_id + 1];
    uint32_t best = 0xFFFFFFFFu;
    for (uint32_t p = ch.start + lane; p < ch.end && best == 0xFFFFFFFFu; p += 64) {
        const uint32_t tile = tb + p / T;
        if (tile >= t_end) break;
        const uint32_t slot = slot_of[tile];
        if (slot == 0xFFFFFFFFu) continue;
        if (span[(size_t)slot * T + (p & (T - 1))]) best = p;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const uint32_t o = __shfl_down(best, d, 64); best = o < best ? o : best; }
    if (lane == 0 && best != 0xFFFFFFFFu) atomicMin(&first[ch.id], best);
}

// column sums of a range: n_bases += addm, cov_count[t] += [covm >= T_t], seen
__global__ __launch_bounds__(kRedThreads) void k_range_reduce_m(
    const RangeChunk* __restrict__ chunks, uint32_t n_chunks, const uint32_t* __restrict__ covm, const uint32_t* __restrict__ addm,
    const uint32_t* __restrict__ span, const uint32_t* __restrict__ slot_of, const uint32_t* __restrict__ tile_base, uint32_t T,
    uint32_t S, const uint32_t* __restrict__ thresholds, uint32_t n_thr, uint32_t* n_bases, uint32_t* cov_counts, uint32_t* seen) {
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t ci = blockIdx.x * (kRedThreads / 64) + wv;
    if (ci >= n_chunks) return;
    const RangeChunk ch = chunks[ci];
    const uint32_t tb = tile_base[ch.ref_id], t_end = tile_base[ch.ref_id + 1];
    uint32_t any = 0;
    for (uint32_t s = 0; s < S; ++s) {
        uint32_t nb = 0;
        uint32_t cc[kMaxThresholds];
#pragma unroll
        for (int t = 0; t < kMaxThresholds; ++t) cc[t] = 0;
        for (uint32_t p = ch.start + lane; p < ch.end; p += 64) {
            const uint32_t tile = tb + p / T;
            if (tile >= t_end) continue;
            const uint32_t slot = slot_of[tile];
            if (slot == 0xFFFFFFFFu) continue;
            const size_t at = (size_t)slot * T + (p & (T - 1));
            if (!span[at]) continue;                     // statistics are gathered per pileup column
            any = 1;
            nb += addm[at * S + s];
            const uint32_t cov = covm[at * S + s];
#pragma unroll
            for (int t = 0; t < kMaxThresholds; ++t)
                if ((uint32_t)t < n_thr && cov >= thresholds[t]) cc[t] += 1;
        }
        nb = wave_sum(nb);
        if (lane == 0 && nb) atomicAdd(&n_bases[(size_t)ch.id * S + s], nb);
#pragma unroll
        for (int t = 0; t < kMaxThresholds; ++t) {
            if ((uint32_t)t < n_thr) {
                uint32_t v = wave_sum(cc[t]);
                if (lane == 0 && v) atomicAdd(&cov_counts[((size_t)ch.id * S + s) * n_thr + t], v);
            }
        }
    }
    const uint64_t am = __ballot(any != 0);
    if (lane == 0 && am) seen[ch.id] = 1;
}

// first column of [oa, ob) inside the union of the regions (sorted, disjoint intervals per contig); 0xFFFFFFFF = none
__device__ uint32_t first_in_union(const SortedRegion* __restrict__ un, uint32_t lo, uint32_t hi, uint32_t oa, uint32_t ob) {
    uint32_t a = lo, c = hi;          // first interval with end > oa
    while (a < c) { const uint32_t m = (a + c) >> 1; if (un[m].end > oa) c = m; else a = m + 1; }
    if (a >= hi || un[a].start >= ob) return 0xFFFFFFFFu;
    return un[a].start > oa ? un[a].start : oa;
}

// per read: countRead / uncountOverlappingMates / countPreviouslySeenMateOverlaps for every region it touches
__global__ __launch_bounds__(kRedThreads) void k_count_reads_mates(
    const uint8_t* __restrict__ U, const RecDesc* __restrict__ desc, uint64_t n_records, const int32_t* __restrict__ rec_ref,
    const uint32_t* __restrict__ mate, const SortedRegion* __restrict__ regs, const uint32_t* __restrict__ pmax_end,
    const uint32_t* __restrict__ ref_first, const SortedRegion* __restrict__ un, const uint32_t* __restrict__ un_first,
    uint32_t everywhere /* windows: every column is inside a region */, const uint32_t* __restrict__ first, uint32_t S,
    uint32_t min_bq, uint32_t* n_bases, uint32_t* n_reads) {
    const uint64_t i = (uint64_t)blockIdx.x * kRedThreads + threadIdx.x;
    if (i >= n_records) return;
    const RecDesc d = desc[i];
    if (d.kind == 0) return;
    const int32_t ref = rec_ref[i];
    const uint32_t lo0 = ref_first[ref], hi0 = ref_first[ref + 1];
    if (lo0 >= hi0) return;
    const uint32_t mi = mate[i];
    RecDesc b;
    b.kind = 0; b.pos = 0; b.end = 0;
    uint32_t oa = 0, ob = 0, f = 0xFFFFFFFFu;
    if (mi != 0xFFFFFFFFu) {
        b = desc[mi];
        oa = (uint32_t)(d.pos > b.pos ? d.pos : b.pos);
        ob = (uint32_t)(d.end < b.end ? d.end : b.end);
        f = everywhere ? oa : first_in_union(un, un_first[ref], un_first[ref + 1], oa, ob);
    }
    const bool pair_owner = b.kind != 0 && i < mi;      // pair terms are applied once, by the pair's first record
    // regions with start < d.end, walking back while anything can still reach the read
    uint32_t a = lo0, c = hi0;
    while (a < c) { uint32_t m = (a + c) >> 1; if ((int64_t)regs[m].start < (int64_t)d.end) a = m + 1; else c = m; }
    const uint32_t s = S > 1 ? d.sample : 0u;
    for (uint32_t j = a; j > lo0;) {
        --j;
        if ((int64_t)pmax_end[j] <= (int64_t)d.pos) break;
        const uint32_t rs = regs[j].start, re = regs[j].end, id = regs[j].id;
        if ((int64_t)re <= (int64_t)d.pos) continue;
        const uint32_t F = first[id];
        if (F == 0xFFFFFFFFu) continue;
        const uint32_t B = count_good(U, d, min_bq, rs, re);
        const bool fixed_at_F = b.kind != 0 && F >= oa && F < ob && f != 0xFFFFFFFFu && f < F;
        const bool spans_F = (int64_t)d.pos <= (int64_t)F && (int64_t)F < (int64_t)d.end;
        const bool counted = (spans_F && !fixed_at_F) || ((int64_t)d.pos > (int64_t)F && (int64_t)d.pos < (int64_t)re);
        if (counted) {
            if (B) atomicAdd(&n_bases[(size_t)id * S + s], B);
            if (B) atomicAdd(&n_reads[(size_t)id * S + s], 1u);
        }
        if (pair_owner && rs < ob && re > oa) {
            const uint32_t B2 = count_good(U, b, min_bq, rs, re);
            if (f != 0xFFFFFFFFu && f >= rs && f < re) {
                const uint32_t from = count_good(U, d, min_bq, f, re) + count_good(U, b, min_bq, f, re);
                if (from) atomicSub(&n_bases[(size_t)id * S + s], from);
                const uint32_t merged = (B + B2 > 0) ? 1u : 0u, apart = (B > 0 ? 1u : 0u) + (B2 > 0 ? 1u : 0u);
                if (apart != merged) atomicSub(&n_reads[(size_t)id * S + s], apart - merged);
            }
            if (fixed_at_F && (B + B2 > 0)) atomicAdd(&n_reads[(size_t)id * S + s], 1u);
        }
    }
}

}  // namespace

void launch_range_reduce(const RangeChunk* d_chunks, uint32_t n_chunks, const uint32_t* d_counters, const uint32_t* d_span,
                         const uint32_t* d_slot_of, const uint32_t* d_tile_base, uint32_t T, uint32_t S,
                         const uint32_t* d_thresholds, uint32_t n_thr, uint32_t* d_n_bases, uint32_t* d_cov_counts,
                         uint32_t* d_seen, hipStream_t stream, bool compact) {
    if (!n_chunks) return;
    const uint32_t per = kRedThreads / 64;
    hipLaunchKernelGGL(k_range_reduce, dim3((n_chunks + per - 1) / per), dim3(kRedThreads), 0, stream, d_chunks, n_chunks,
                       d_counters, d_span, d_slot_of, d_tile_base, T, S, d_thresholds, n_thr, d_n_bases, d_cov_counts, d_seen,
                       compact ? 1u : 0u, (const uint64_t*)nullptr, (const uint64_t*)nullptr, 0u, 0u);
    SBX_HIP(hipGetLastError());
}

void launch_range_reduce_windows(const uint64_t* d_win_base, const uint64_t* d_n_win, uint32_t n_ref, uint32_t window, uint32_t n_windows,
                                 const uint32_t* d_counters, const uint32_t* d_span, const uint32_t* d_slot_of, const uint32_t* d_tile_base,
                                 uint32_t T, uint32_t S, const uint32_t* d_thresholds, uint32_t n_thr, uint32_t* d_n_bases,
                                 uint32_t* d_cov_counts, uint32_t* d_seen, hipStream_t stream, bool compact) {
    if (!n_windows || !n_ref) return;
    const uint32_t per = kRedThreads / 64;
    hipLaunchKernelGGL(k_range_reduce, dim3((n_windows + per - 1) / per), dim3(kRedThreads), 0, stream, (const RangeChunk*)nullptr, n_windows,
                       d_counters, d_span, d_slot_of, d_tile_base, T, S, d_thresholds, n_thr, d_n_bases, d_cov_counts, d_seen,
                       compact ? 1u : 0u, d_win_base, d_n_win, n_ref, window);
    SBX_HIP(hipGetLastError());
}

void launch_count_reads_windows(const uint8_t* d_U, const RecDesc* d_desc, uint64_t n_records, const int32_t* d_rec_ref,
                                uint32_t window, const uint64_t* d_win_base, const uint64_t* d_n_win, uint32_t S,
                                uint32_t min_bq, uint32_t* d_n_reads, hipStream_t stream) {
    if (!n_records) return;
    hipLaunchKernelGGL(k_count_reads_windows, dim3((uint32_t)((n_records + kRedThreads - 1) / kRedThreads)), dim3(kRedThreads), 0,
                       stream, d_U, d_desc, n_records, d_rec_ref, window, d_win_base, d_n_win, S, min_bq, d_n_reads);
    SBX_HIP(hipGetLastError());
}

void launch_count_reads_regions(const uint8_t* d_U, const RecDesc* d_desc, uint64_t n_records, const int32_t* d_rec_ref,
                                const SortedRegion* d_regs, const uint32_t* d_pmax_end, const uint32_t* d_ref_first, uint32_t S,
                                uint32_t min_bq, uint32_t* d_n_reads, const uint32_t* d_min_start, uint32_t* d_n_bases,
                                hipStream_t stream) {
    if (!n_records) return;
    hipLaunchKernelGGL(k_count_reads_regions, dim3((uint32_t)((n_records + kRedThreads - 1) / kRedThreads)), dim3(kRedThreads), 0,
                       stream, d_U, d_desc, n_records, d_rec_ref, d_regs, d_pmax_end, d_ref_first, S, min_bq, d_n_reads, d_min_start,
                       d_n_bases);
    SBX_HIP(hipGetLastError());
}

// dst[dst_slot_of[tile of src slot k]] += src[k] for every tile slot of another BAM's run (multi-BAM merge)
namespace {
__global__ __launch_bounds__(kRedThreads) void k_merge_tiles(const uint32_t* __restrict__ src, const uint32_t* __restrict__ src_active,
                                                             const uint32_t* __restrict__ dst_slot_of, uint32_t per_tile,
                                                             uint32_t* __restrict__ dst) {
    const uint32_t tile = src_active[blockIdx.x];
    const uint32_t ds = dst_slot_of[tile];
    const uint32_t* a = src + (size_t)blockIdx.x * per_tile;
    uint32_t* b = dst + (size_t)ds * per_tile;
    for (uint32_t i = threadIdx.x; i < per_tile; i += kRedThreads) b[i] += a[i];
}
}  // namespace

void launch_merge_tiles(const uint32_t* d_src, const uint32_t* d_src_active, uint32_t n_src_active, const uint32_t* d_dst_slot_of,
                        uint32_t per_tile, uint32_t* d_dst, hipStream_t stream) {
    if (!n_src_active) return;
    hipLaunchKernelGGL(k_merge_tiles, dim3(n_src_active), dim3(kRedThreads), 0, stream, d_src, d_src_active, d_dst_slot_of, per_tile, d_dst);
    SBX_HIP(hipGetLastError());
}

void launch_range_first(const RangeChunk* d_chunks, uint32_t n_chunks, const uint32_t* d_span, const uint32_t* d_slot_of,
                        const uint32_t* d_tile_base, uint32_t T, uint32_t* d_first, hipStream_t stream) {
    if (!n_chunks) return;
    const uint32_t per = kRedThreads / 64;
    hipLaunchKernelGGL(k_range_first, dim3((n_chunks + per - 1) / per), dim3(kRedThreads), 0, stream, d_chunks, n_chunks, d_span, d_slot_of,
                       d_tile_base, T, d_first);
    SBX_HIP(hipGetLastError());
}

void launch_range_reduce_m(const RangeChunk* d_chunks, uint32_t n_chunks, const uint32_t* d_covm, const uint32_t* d_addm,
                           const uint32_t* d_span, const uint32_t* d_slot_of, const uint32_t* d_tile_base, uint32_t T, uint32_t S,
                           const uint32_t* d_thresholds, uint32_t n_thr, uint32_t* d_n_bases, uint32_t* d_cov_counts, uint32_t* d_seen,
                           hipStream_t stream) {
    if (!n_chunks) return;
    const uint32_t per = kRedThreads / 64;
    hipLaunchKernelGGL(k_range_reduce_m, dim3((n_chunks + per - 1) / per), dim3(kRedThreads), 0, stream, d_chunks, n_chunks, d_covm,
                       d_addm, d_span, d_slot_of, d_tile_base, T, S, d_thresholds, n_thr, d_n_bases, d_cov_counts, d_seen);
    SBX_HIP(hipGetLastError());
}

void launch_count_reads_mates(const uint8_t* d_U, const RecDesc* d_desc, uint64_t n_records, const int32_t* d_rec_ref,
                              const uint32_t* d_mate, const SortedRegion* d_regs, const uint32_t* d_pmax_end, const uint32_t* d_ref_first,
                              const SortedRegion* d_union, const uint32_t* d_union_first, bool everywhere, const uint32_t* d_first,
                              uint32_t S, uint32_t min_bq, uint32_t* d_n_bases, uint32_t* d_n_reads, hipStream_t stream) {
    if (!n_records) return;
    hipLaunchKernelGGL(k_count_reads_mates, dim3((uint32_t)((n_records + kRedThreads - 1) / kRedThreads)), dim3(kRedThreads), 0, stream,
                       d_U, d_desc, n_records, d_rec_ref, d_mate, d_regs, d_pmax_end, d_ref_first, d_union, d_union_first,
                       everywhere ? 1u : 0u, d_first, S, min_bq, d_n_bases, d_n_reads);
    SBX_HIP(hipGetLastError());
}

}  // namespace sbx
