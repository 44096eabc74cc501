// mates.hip -- K7: `depth base --fix-mate-overlaps` on the device.
//
// Replaces ColumnPrinter.detectOverlappingMates / selectBetterMate and the pair handling of
// PerBasePrinter.writeColumn (sambamba/depth.d:319-399,520-532).  In the reference, per pileup column,
// reads are sorted by the FNV-1a hash of their name (depth.d:252-258,338); two adjacent reads with the
// same hash, name and sample form a pair and only the *better* mate is counted at that column: if
// either is inside a D/N operation the higher mapping quality wins, otherwise the higher base quality;
// ties go to the second one (depth.d:391-399).  As a histogram over reads this reads: a read A with one
// same-name partner B contributes at position p iff B does not span p, or A wins at p.
//   * `find_mates`      one lane per record scans forward over the records that start inside its span
//                       (coordinate order => a contiguous index range; consecutive lanes read consecutive
//                       hashes => coalesced) and links same-hash, same-sample, overlapping records.
//   * `accumulate_mates` the tile kernel of depth.hip with a per-position path for linked reads: each
//                       lane takes one reference position, evaluates both mates' CIGAR cursors there
//                       and applies the reference's rule.
// Name groups of more than two records (supplementary / secondary alignments next to their primaries; depth.d:373
// "don't consider rare cases of >= 3 reads with the same name") follow the reference's loop literally: at a column the
// same-name records covering it pair up in column order -- (1st, 2nd), (3rd, 4th) -- and an odd one is left alone.  What a
// record contributes then also depends on its status (depth.d:355-371, 522-532): a record is processed on its own at
// every column where its status is not `detected`, and the better mate of every pair is processed in addition, so a
// record that was paired, was alone again (`past`) and is paired a second time counts twice where it wins.  For up to
// three same-name records per column all of this follows from geometry (`k_find_partners` lists up to three partners per
// record, `MultiPlan` below); four or more mutually overlapping same-name records -- where the status of the odd one
// depends on the hash order of unrelated reads in the column -- raise SBX_EUNSUPPORTED, as do groups of more than two in
// region / window mode, whose closed form (reduce.hip) is derived for pairs.  Ties are resolved as in the oracle: the
// record later in the file wins (column order; parity unpinned).  Names are compared byte for byte after the hashes
// match (depth.d:352-353).
#include <cstdlib>

#include "common.hpp"
#include "kernels.hpp"

namespace sbx {

namespace {

constexpr int kMateThreads = 256;
constexpr uint32_t kCigarTypeM = 0x3C1A7u;

__device__ __forceinline__ uint32_t ld32m(const uint8_t* p) {
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}
__device__ __forceinline__ uint32_t base5_m(uint32_t nib) {
    const uint64_t lut = (4ULL << 0) | (0ULL << 3) | (1ULL << 6) | (4ULL << 9) | (2ULL << 12) | (4ULL << 15) |
                         (4ULL << 18) | (4ULL << 21) | (3ULL << 24) | (4ULL << 27) | (4ULL << 30) | (4ULL << 33) |
                         (4ULL << 36) | (4ULL << 39) | (4ULL << 42) | (4ULL << 45);
    return (uint32_t)(lut >> (nib * 3)) & 7u;
}
__device__ __forceinline__ uint32_t pos_dw_m(uint32_t p, uint32_t sub_dw, uint32_t s7) { return (p & 3u) * sub_dw + (p >> 2) * s7; }

// equal read names (depth.d:353 compares the names after the hashes)
__device__ bool same_name(const uint8_t* U, const RecDesc& a, const RecDesc& b) {
    if (a.l_name != b.l_name) return false;
    const uint8_t* x = U + a.rec_off + 36;
    const uint8_t* y = U + b.rec_off + 36;
    for (uint32_t k = 0; k < a.l_name; ++k) if (x[k] != y[k]) return false;
    return true;
}

// A workgroup of find_mates / find_partners looks at the 1024 records from its first one on (a window staged in LDS); a lane whose
// span reaches beyond the window (coverage in the thousands) finishes in global memory.
constexpr uint32_t kFindWin = 1024;

__device__ __forceinline__ void link_if_mates(const uint8_t* U, const RecDesc* desc, const uint64_t* hash, const RecDesc& a, uint64_t h,
                                              uint64_t i, uint64_t j, uint32_t* mate, uint32_t* n_partners) {
    const RecDesc b = desc[j];
    if (b.kind != 0 && hash[j] == h && b.sample == a.sample && b.end > a.pos && same_name(U, a, b)) {
        mate[i] = (uint32_t)j;
        mate[j] = (uint32_t)i;
        atomicAdd(&n_partners[i], 1u);
        atomicAdd(&n_partners[j], 1u);
    }
}


// The links through a hash JOIN (round 4; round 3's kernel scanned the window entries that start inside a record's span -- it read a few hundred window entries per record at
// 300x to find the one or two with its hash.  Here the workgroup puts the admitted records of its window into an open-addressing
// table in LDS keyed by the half hash (2,048 slots for <= 1,024 entries: mates share a key and sit in neighbouring slots), and a
// record probes for its own key: two or three slots instead of three hundred entries.  A candidate links under the scan's conditions
// -- it lies behind the record in the file, on the same reference, and starts inside the record's span (the scan stops at the first
// entry that does not; the file is sorted, so that is the same set) -- and a span that reaches beyond the window is finished in
// global memory as before.
constexpr uint32_t kJoinSlots = 2048;
__global__ __launch_bounds__(kMateThreads) void k_find_mates_join(const uint8_t* __restrict__ U, const RecDesc* __restrict__ desc,
                                                                   const uint64_t* __restrict__ hash, const int32_t* __restrict__ rec_ref,
                                                                   uint64_t n, uint32_t* mate, uint32_t* n_partners) {
    __shared__ int32_t s_pos[kFindWin];
    __shared__ uint32_t s_h[kFindWin], s_ref[kFindWin];
    __shared__ uint32_t t_key[kJoinSlots], t_val[kJoinSlots];
    const uint64_t i0 = (uint64_t)blockIdx.x * kMateThreads;
    for (uint32_t k = threadIdx.x; k < kJoinSlots; k += kMateThreads) t_val[k] = 0xFFFFFFFFu;
    for (uint32_t k = threadIdx.x; k < kFindWin; k += kMateThreads) {
        const uint64_t j = i0 + k;
        if (j < n) {
            const RecDesc d = desc[j];
            s_pos[k] = d.pos;
            s_h[k] = (uint32_t)hash[j];
            s_ref[k] = ((uint32_t)rec_ref[j] & 0x7FFFFFFFu) | (d.kind == 0 ? 0x80000000u : 0u);
        } else {
            s_pos[k] = 0x7FFFFFFF;
            s_h[k] = 0;
            s_ref[k] = 0x7FFFFFFEu;      // no reference has this id
        }
    }
    __syncthreads();
    // insert: entry k of the window (k >= 1: entry 0 can only be somebody's predecessor, never a partner behind a record)
    for (uint32_t k = threadIdx.x; k < kFindWin; k += kMateThreads) {
        if (k == 0u || (s_ref[k] >> 31) || s_ref[k] == 0x7FFFFFFEu) continue;
        const uint32_t key = s_h[k];
        uint32_t slot = (key * 0x9E3779B1u) >> 21;
        for (;;) {
            if (atomicCAS(&t_val[slot], 0xFFFFFFFFu, k) == 0xFFFFFFFFu) { t_key[slot] = key; break; }
            slot = (slot + 1u) & (kJoinSlots - 1u);
        }
    }
    __syncthreads();
    const uint64_t i = i0 + threadIdx.x;
    if (i >= n) return;
    const RecDesc a = desc[i];
    if (a.kind == 0) return;
    const uint64_t h = hash[i];
    const int32_t ref = rec_ref[i];
    const uint32_t ref_tag = (uint32_t)ref & 0x7FFFFFFFu, h32 = (uint32_t)h;
    for (uint32_t slot = (h32 * 0x9E3779B1u) >> 21;; slot = (slot + 1u) & (kJoinSlots - 1u)) {
        const uint32_t k = t_val[slot];
        if (k == 0xFFFFFFFFu) break;
        if (t_key[slot] == h32 && k > threadIdx.x && (s_ref[k] & 0x7FFFFFFFu) == ref_tag && s_pos[k] < a.end)
            link_if_mates(U, desc, hash, a, h, i, i0 + k, mate, n_partners);
    }
    // the span reaches beyond the window: the rest in global memory
    if ((s_ref[kFindWin - 1] & 0x7FFFFFFFu) == ref_tag && s_pos[kFindWin - 1] < a.end)
        for (uint64_t j = i0 + kFindWin; j < n; ++j) {
            const RecDesc b = desc[j];
            if (rec_ref[j] != ref || b.pos >= a.end) break;
            if (hash[j] == h) link_if_mates(U, desc, hash, a, h, i, j, mate, n_partners);
        }
}

// Second pass, only when some record has more than one partner: up to three partners per record in ext[3 i ..],
// in no particular order (n_partners is counted again; a fourth partner only raises the count).
__global__ __launch_bounds__(kMateThreads) void k_find_partners(const uint8_t* __restrict__ U, const RecDesc* __restrict__ desc,
                                                                 const uint64_t* __restrict__ hash, const int32_t* __restrict__ rec_ref,
                                                                 uint64_t n, uint32_t* ext, uint32_t* n_partners) {
    const uint64_t i = (uint64_t)blockIdx.x * kMateThreads + threadIdx.x;
    if (i >= n) return;
    const RecDesc a = desc[i];
    if (a.kind == 0) return;
    const uint64_t h = hash[i];
    const int32_t ref = rec_ref[i];
    for (uint64_t j = i + 1; j < n; ++j) {
        const RecDesc b = desc[j];
        if (rec_ref[j] != ref || b.pos >= a.end) break;
        if (b.kind != 0 && hash[j] == h && b.sample == a.sample && b.end > a.pos && same_name(U, a, b)) {
            const uint32_t si = atomicAdd(&n_partners[i], 1u), sj = atomicAdd(&n_partners[j], 1u);
            if (si < 3u) ext[3 * i + si] = (uint32_t)j;
            if (sj < 3u) ext[3 * j + sj] = (uint32_t)i;
        }
    }
}

struct Cursor {       // what PileupRead shows at one reference position (pileup.d:115-134)
    uint32_t kind;    // 0 not spanning, 1 M/=/X base, 2 D, 3 N
    uint32_t nib, qual;
};

// CIGAR cursor of record d at reference position p (same conventions as K3: a zero-length
// reference-consuming op occupies one column, the read ends at d.end)
__device__ Cursor state_at(const uint8_t* U, const RecDesc& d, int32_t p) {
    Cursor c{0, 0, 0};
    if (d.kind == 0 || p < d.pos || p >= d.end) return c;
    const uint8_t* rec = U + d.rec_off;
    const uint8_t* cig = rec + 36 + d.l_name;
    const uint8_t* seq = cig + 4 * (uint32_t)d.n_cigar;
    const uint8_t* qual = seq + ((d.l_seq + 1) >> 1);
    auto at_query = [&](uint32_t q) {
        if (q >= d.l_seq) return;
        const uint32_t sb = seq[q >> 1];
        c.kind = 1;
        c.nib = (q & 1u) ? (sb & 15u) : (sb >> 4);
        c.qual = qual[q];
    };
    if (d.kind == 1) { at_query((uint32_t)d.q_start + (uint32_t)(p - d.pos)); return c; }
    int32_t rp = d.pos;
    uint32_t qp = 0;
    for (uint32_t k = 0; k < d.n_cigar; ++k) {
        uint32_t op = ld32m(cig + 4 * k);
        uint32_t ty = (kCigarTypeM >> ((op & 15u) * 2u)) & 3u, len = op >> 4;
        if (ty & 2u) {
            if (len == 0) len = 1;
            const int32_t room = d.end - rp;
            if ((int64_t)len > (int64_t)room) len = (uint32_t)(room > 0 ? room : 0);
            if (p < rp + (int32_t)len) {
                if (ty == 3) at_query(qp + (uint32_t)(p - rp));
                else c.kind = (op & 15u) == 2u ? 2u : 3u;
                return c;
            }
            rp += (int32_t)len;
            if (ty == 3) qp += len;
        } else if (ty == 1) {
            qp += len;
        }
    }
    return c;
}

// What record A (index ia) with two or three same-name partners does along its span, wave-uniform: the span is cut at
// the partners' starts and ends into at most seven intervals of constant company; in each, the records covering it pair
// up in file order, which gives A's partner there (or none).  A's status is `none` before its first paired column,
// `detected` from there until the first column at which it is alone again, `past` ever after (depth.d:355-371).
struct MultiPlan {
    int32_t cut[8];          // interval k = [cut[k], cut[k + 1])
    uint32_t partner[7];     // index of A's partner in interval k, 0xFFFFFFFF = alone
    int n_iv;
    int32_t first_paired, first_alone_again;      // columns; INT32_MAX when there is none
    bool too_many;           // four or more same-name records cover some column
};

__device__ MultiPlan plan_multi(const RecDesc* __restrict__ desc, uint32_t ia, const RecDesc& a, const uint32_t* __restrict__ ext, uint32_t np) {
    MultiPlan P;
    uint32_t pi[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    int32_t pb[3] = {0, 0, 0}, pe[3] = {0, 0, 0};
    const uint32_t n = np < 3u ? np : 3u;
    for (uint32_t k = 0; k < n; ++k) {
        pi[k] = ext[3 * (size_t)ia + k];
        const RecDesc b = desc[pi[k]];
        pb[k] = b.pos; pe[k] = b.end;
    }
    // breakpoints inside (a.pos, a.end), sorted, duplicates removed
    int32_t c[8];
    int nc = 0;
    c[nc++] = a.pos;
    for (uint32_t k = 0; k < n; ++k) {
        if (pb[k] > a.pos && pb[k] < a.end) c[nc++] = pb[k];
        if (pe[k] > a.pos && pe[k] < a.end) c[nc++] = pe[k];
    }
    c[nc++] = a.end;
    for (int x = 1; x < nc; ++x)
        for (int y = x; y > 0 && c[y] < c[y - 1]; --y) { const int32_t t = c[y]; c[y] = c[y - 1]; c[y - 1] = t; }
    int m = 0;
    for (int x = 0; x < nc; ++x) if (m == 0 || c[x] != P.cut[m - 1]) P.cut[m++] = c[x];
    P.n_iv = m - 1;
    P.too_many = np > 3u;
    P.first_paired = 0x7FFFFFFF;
    P.first_alone_again = 0x7FFFFFFF;
    for (int v = 0; v < P.n_iv; ++v) {
        const int32_t q = P.cut[v];                   // company is constant on the interval: test its first column
        // records covering q, in file order: rank of A and its neighbours
        uint32_t before = 0, total = 1, prev = 0xFFFFFFFFu, next = 0xFFFFFFFFu;
        for (uint32_t k = 0; k < n; ++k) {
            if (q < pb[k] || q >= pe[k]) continue;
            ++total;
            if (pi[k] < ia) { ++before; if (prev == 0xFFFFFFFFu || pi[k] > prev) prev = pi[k]; }
            else if (next == 0xFFFFFFFFu || pi[k] < next) next = pi[k];
        }
        if (total >= 4u) P.too_many = true;
        uint32_t partner = 0xFFFFFFFFu;
        if (before & 1u) partner = prev;              // second of a pair
        else if (before + 1u < total) partner = next; // first of a pair
        P.partner[v] = partner;
        if (partner != 0xFFFFFFFFu) { if (P.first_paired == 0x7FFFFFFF) P.first_paired = q; }
        else if (P.first_paired != 0x7FFFFFFF && P.first_alone_again == 0x7FFFFFFF) P.first_alone_again = q;
    }
    return P;
}

// ---- fast path of k_accumulate_mates: a single-run read and its (at most one) single-run mate ----------------------------
// The per-position path below evaluates two CIGAR cursors per position and lane: ten times the work of the plain tile
// kernel per read (config 5: 116 of 246 ms).  For a read A whose alignment is one run of M/=/X and whose partner B (if it
// has one) is of the same kind -- every ordinary pair of a paired-end library -- the cursors are arithmetic: A's base at
// position p is query offset q_start + p - pos, and the reference's rule (depth.d:391-399, ties to the record later in the
// file) is a comparison of two quality bytes.  Every lane takes an ALIGNED block of 16 tile positions of one read: 12 bytes of
// packed sequence, 16 quality bytes of A and the 16 quality bytes of B at the same positions, loaded as dwords; reads are
// packed into lanes by a prefix sum of their block counts (as in k_accumulate16b).  A read is eligible iff eligible_fast();
// the per-position path skips exactly those.
#define SBX_M_LUT_OFF_ENTRY(n) (((n) == 1 || (n) == 2 ? 0u : (n) == 4 || (n) == 8 ? 1u : 2u) << (2 * (n)))
#define kMLutOff (SBX_M_LUT_OFF_ENTRY(0) | SBX_M_LUT_OFF_ENTRY(1) | SBX_M_LUT_OFF_ENTRY(2) | SBX_M_LUT_OFF_ENTRY(3) | SBX_M_LUT_OFF_ENTRY(4) | \
                  SBX_M_LUT_OFF_ENTRY(5) | SBX_M_LUT_OFF_ENTRY(6) | SBX_M_LUT_OFF_ENTRY(7) | SBX_M_LUT_OFF_ENTRY(8) | SBX_M_LUT_OFF_ENTRY(9) | \
                  SBX_M_LUT_OFF_ENTRY(10) | SBX_M_LUT_OFF_ENTRY(11) | SBX_M_LUT_OFF_ENTRY(12) | SBX_M_LUT_OFF_ENTRY(13) | SBX_M_LUT_OFF_ENTRY(14) | \
                  SBX_M_LUT_OFF_ENTRY(15))
#define kMLutHalf ((1u << 4) | (1u << 16))
constexpr uint32_t kMateMapBytes = 704;      // 64 reads x 11 blocks
constexpr uint32_t kMateWorkCap = 382;       // records of a tile left to the per-position path (u16 indices) + two counters: 768 bytes

__device__ __forceinline__ bool single_run_ok(const RecDesc& d) {      // one run of aligned bases that lies inside the sequence
    return d.kind == 1 && (uint64_t)d.q_start + (uint64_t)(d.end - d.pos) <= (uint64_t)d.l_seq;
}
__device__ __forceinline__ bool eligible_fast(const RecDesc& a, uint32_t np, bool has_mate, const RecDesc& b) {
    return np <= 1u && single_run_ok(a) && (!has_mate || single_run_ok(b));
}
__device__ __forceinline__ uint32_t nibble_swap_m(uint32_t x) { return ((x & 0x0F0F0F0Fu) << 4) | ((x >> 4) & 0x0F0F0F0Fu); }

template <bool kSpan>
__global__ __launch_bounds__(kMateThreads) void k_accumulate_mates(
    const uint8_t* __restrict__ U, const RecDesc* __restrict__ desc, const uint32_t* __restrict__ mate,
    const uint32_t* __restrict__ tile_lo, const uint32_t* __restrict__ tile_hi, const uint32_t* __restrict__ active,
    const uint32_t* __restrict__ tile_base, int32_t n_ref, uint32_t T, uint32_t S, uint32_t min_bq,
    const uint32_t* __restrict__ ext, const uint32_t* __restrict__ n_partners, uint32_t* __restrict__ too_many,
    uint32_t* __restrict__ counters, uint32_t* __restrict__ span_out, int fast_path) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const uint32_t s7 = S * 7;
    const uint32_t sub_dw = (T / 4) * s7 + 8;
    uint32_t* cnt = lds;
    uint32_t* spn = lds + 4 * sub_dw;
    const uint32_t tile = active[blockIdx.x];
    const uint32_t spn_dw = kSpan ? T : 0u;
    for (uint32_t i = threadIdx.x; i < 4 * sub_dw + spn_dw; i += kMateThreads) lds[i] = 0;
    int lo_r = 0, hi_r = n_ref;
    while (hi_r - lo_r > 1) {
        int mid = (lo_r + hi_r) >> 1;
        if (tile_base[mid] <= tile) lo_r = mid; else hi_r = mid;
    }
    const int32_t ts = (int32_t)((tile - tile_base[lo_r]) * T), te = ts + (int32_t)T;
    const uint32_t r_lo = tile_lo[tile], r_hi = tile_hi[tile];
    // the records phase 1 leaves to the per-position path are listed here (indices relative to r_lo), so that phase 2 does not
    // have to walk all records of the tile again -- descriptor -> mate -> mate's descriptor, three dependent loads each -- just
    // to find that there is nothing left for it; work_n[0] = entries, work_n[1] != 0: the list overflowed, phase 2 scans
    uint16_t* const work = (uint16_t*)((uint8_t*)(lds + 4 * sub_dw + spn_dw) + (kMateThreads / 64) * kMateMapBytes);
    uint32_t* const work_n = (uint32_t*)(work + kMateWorkCap);
    if (fast_path && threadIdx.x < 2) work_n[threadIdx.x] = threadIdx.x == 1 && r_hi - r_lo > 0xFFFFu ? 1u : 0u;
    // span counts of the reads phase 1 takes: a difference array (+1 at the first, -1 behind the last position of the clipped run --
    // two atomics per read instead of one per base, whose lanes, 16 positions apart, would share four banks), integrated when the
    // tile is written; [T + 1] entries, then 4 wave totals
    int32_t* const sdf = (int32_t*)(work_n + 2);
    if (kSpan && fast_path)
        for (uint32_t i = threadIdx.x; i < T + 5u; i += kMateThreads) sdf[i] = 0;
    __syncthreads();
    if (fast_path) {
        // ---- phase 1: eligible reads, 16 aligned positions per lane ------------------------------------------------------
        const uint32_t wlane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        uint8_t* map = (uint8_t*)(lds + 4 * sub_dw + spn_dw) + wave * kMateMapBytes;
        for (uint32_t c0 = r_lo + wave * 64u; c0 < r_hi; c0 += (kMateThreads / 64) * 64u) {
            const uint32_t ri = c0 + wlane;
            RecDesc a, b;
            a.kind = 0; a.pos = 0; a.end = 0; a.rec_off = 0; a.l_seq = 0; a.n_cigar = 0; a.l_name = 0; a.q_start = 0; a.sample = 0; a.mapq = 0;
            b = a;
            uint32_t mi = 0xFFFFFFFFu, np = 0;
            if (ri < r_hi) {
                a = desc[ri];
                if (a.kind != 0 && a.pos < te && a.end > ts) {
                    np = ext ? n_partners[ri] : 0u;
                    mi = mate[ri];
                    if (mi != 0xFFFFFFFFu) b = desc[mi];
                }
            }
            const bool has_mate = mi != 0xFFFFFFFFu;
            const bool el = a.kind != 0 && a.pos < te && a.end > ts && eligible_fast(a, np, has_mate, b);
            // clipped run of A: tile offsets [t0, t0 + n), first base = query offset q0
            const int32_t i0 = ts > a.pos ? ts - a.pos : 0;
            const int32_t i1 = a.end - a.pos < te - a.pos ? a.end - a.pos : te - a.pos;
            const uint32_t n_run = el && i1 > i0 ? (uint32_t)(i1 - i0) : 0u;
            const uint32_t t0 = (uint32_t)(a.pos + i0 - ts);
            const uint32_t nblk_all = n_run ? ((t0 + n_run - 1u) >> 4) - (t0 >> 4) + 1u : 0u;
            // (runs of more than eleven blocks -- long reads -- would overflow the map: they stay on the per-position path; the
            //  test below is repeated there)
            const uint32_t nblk = nblk_all <= 11u ? nblk_all : 0u;
            {
                const bool todo = a.kind != 0 && a.pos < te && a.end > ts && !(el && nblk_all <= 11u);
                const uint64_t tm = __ballot(todo);
                if (tm) {
                    uint32_t at = 0;
                    if (wlane == 0) at = atomicAdd(&work_n[0], (uint32_t)__popcll(tm));
                    at = __builtin_amdgcn_readfirstlane(at) + (uint32_t)__popcll(tm & ((1ull << wlane) - 1ull));
                    if (todo) {
                        if (at < kMateWorkCap) work[at] = (uint16_t)(ri - r_lo);
                        else work_n[1] = 1u;
                    }
                }
            }
            const uint32_t q0 = (uint32_t)a.q_start + (uint32_t)i0;
            if (kSpan && nblk != 0u) { atomicAdd(&sdf[t0], 1); atomicAdd(&sdf[t0 + n_run], -1); }
            const uint64_t a_seq = a.rec_off + 36u + a.l_name + 4u * (uint32_t)a.n_cigar;
            const uint64_t a_nib = a_seq + (q0 >> 1);                                   // byte of the run's first base
            const uint64_t a_qual = a_seq + ((a.l_seq + 1u) >> 1) + q0;                 // quality of the run's first base
            // B's quality byte at tile offset x: b_q0 + x (only meaningful inside the overlap [ob0, ob1))
            const uint64_t b_q0 = b.rec_off + 36u + b.l_name + 4u * (uint32_t)b.n_cigar + ((b.l_seq + 1u) >> 1) + b.q_start + (uint64_t)(int64_t)(ts - b.pos);
            int32_t ob0 = 0, ob1 = 0;
            if (has_mate) { ob0 = (b.pos > ts ? b.pos : ts) - ts; ob1 = (b.end < te ? b.end : te) - ts; if (ob1 < ob0) ob1 = ob0; }
            const uint32_t tie = has_mate && !(ri < mi) ? 1u : 0u;                     // ties go to the record later in the file
            uint32_t incl = nblk;
#pragma unroll
            for (int dlt = 1; dlt < 64; dlt <<= 1) {
                const uint32_t o = __shfl_up(incl, dlt, 64);
                if ((int)wlane >= dlt) incl += o;
            }
            const uint32_t start = incl - nblk;
            const uint32_t total = __builtin_amdgcn_readlane(incl, 63);
            const uint32_t p0w = (uint32_t)a_nib, p1w = (uint32_t)(a_nib >> 32) | (start << 16) | ((q0 & 1u) << 31);
            const uint32_t p2w = t0 | (n_run << 16);
            const uint32_t p3w = (uint32_t)a_qual, p4w = (uint32_t)(a_qual >> 32) | (tie << 16) | ((S > 1 ? (uint32_t)a.sample : 0u) << 17);
            const uint32_t p5w = (uint32_t)b_q0, p6w = (uint32_t)(b_q0 >> 32);
            const uint32_t p7w = (uint32_t)ob0 | ((uint32_t)ob1 << 16);
            for (uint32_t j = 0; j < 11u; ++j)
                if (j < nblk) map[start + j] = (uint8_t)wlane;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            for (uint32_t g0 = 0; g0 < total; g0 += 64u) {
                const uint32_t g = g0 + wlane;
                const bool on = g < total;
                const uint32_t r = on ? (uint32_t)map[g] : 0u;
                const uint32_t w0 = __shfl(p0w, r, 64), w1 = __shfl(p1w, r, 64), w2 = __shfl(p2w, r, 64), w3 = __shfl(p3w, r, 64);
                const uint32_t w4 = __shfl(p4w, r, 64), w5 = __shfl(p5w, r, 64), w6 = __shfl(p6w, r, 64), w7 = __shfl(p7w, r, 64);
                if (!on) continue;
                const uint32_t rt0 = w2 & 0xFFFFu, rn = w2 >> 16;
                const uint32_t j = g - ((w1 >> 16) & 0x7FFFu);
                const uint32_t pb = ((rt0 >> 4) + j) << 4;
                const uint32_t k0 = pb < rt0 ? rt0 - pb : 0u;
                const uint32_t k1 = rt0 + rn - pb < 16u ? rt0 + rn - pb : 16u;
                const uint32_t vm = ((1u << k1) - 1u) & ~((1u << k0) - 1u);
                // overlap with the mate inside this block
                const int32_t o0 = (int32_t)(w7 & 0xFFFFu) - (int32_t)pb, o1 = (int32_t)(w7 >> 16) - (int32_t)pb;
                const uint32_t ok0 = o0 < 0 ? 0u : o0 > 16 ? 16u : (uint32_t)o0, ok1 = o1 < 0 ? 0u : o1 > 16 ? 16u : (uint32_t)o1;
                const uint32_t ovm = ok1 > ok0 ? ((1u << ok1) - 1u) & ~((1u << ok0) - 1u) : 0u;
                // sequence nibbles of positions pb .. pb + 15 (stream order), as k_accumulate16b
                const int32_t d0 = (int32_t)pb - (int32_t)rt0;                                   // >= -15
                const int32_t qi = (int32_t)(w1 >> 31) + d0;
                const uint8_t* sp = U + (((uint64_t)(w1 & 0xFFFFu) << 32) | w0) + (qi >> 1);
                const uint32_t y0 = nibble_swap_m(ld32m(sp)), y1 = nibble_swap_m(ld32m(sp + 4)), y2 = nibble_swap_m(ld32m(sp + 8));
                const uint32_t nsh = ((uint32_t)qi & 1u) * 4u;
                const uint32_t b_lo = __builtin_amdgcn_alignbit(y1, y0, nsh), b_hi = __builtin_amdgcn_alignbit(y2, y1, nsh);
                // qualities of A and of B at the same 16 positions
                const uint8_t* qa_p = U + (((uint64_t)(w4 & 0xFFFFu) << 32) | w3) + d0;
                const uint8_t* qb_p = U + (((uint64_t)w6 << 32) | w5) + pb;
                uint32_t QA[4], QB[4];
#pragma unroll
                for (int x = 0; x < 4; ++x) { QA[x] = ld32m(qa_p + 4 * x); QB[x] = ovm ? ld32m(qb_p + 4 * x) : 0u; }
                const uint32_t tie_r = (w4 >> 16) & 1u, smp = w4 >> 17;
                uint32_t* cbase = cnt + __umul24(pb >> 2, s7) + __umul24(smp, 7u);
#pragma unroll
                for (uint32_t k = 0; k < 16; ++k) {
                    const uint32_t wv = k < 8 ? b_lo : b_hi, sh = 4u * (k & 7u);
                    const uint32_t nib2 = sh ? (wv >> (sh - 1u)) & 0x1Eu : (wv << 1) & 0x1Eu;
                    const uint32_t code = (((kMLutOff >> nib2) & 3u) << 1) | ((kMLutHalf >> nib2) & 1u);      // A 0 C 1 G 2 T 3 other 4
                    const uint32_t qa = (QA[k >> 2] >> (8u * (k & 3u))) & 0xFFu, qb = (QB[k >> 2] >> (8u * (k & 3u))) & 0xFFu;
                    const uint32_t win = (qb - qa - tie_r) >> 31;            // 1: A is the better mate here (qa + tie > qb)
                    const uint32_t low = (qa - min_bq) >> 31;                 // 1: below the base-quality threshold
                    const uint32_t v = (vm >> k) & 1u, ov = (ovm >> k) & 1u;
                    const uint32_t inc = v & (low ^ 1u) & ((ov ^ 1u) | win);
                    // dword of position pb + k: (k & 3) * sub_dw + ((pb >> 2) + (k >> 2)) * s7
                    atomicAdd(cbase + (k & 3u) * sub_dw + (k >> 2) * s7 + code, inc);
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    // One record per QUARTER wave and iteration, every lane one reference position of it per pass: a 150-base read
    // fills 16 lanes ten times over (94 % of the lanes busy, 78 % with 64 lanes per read), and the chain of dependent
    // loads a record needs -- descriptor -> mate index -> mate's descriptor -> both reads' bytes -- is in flight for
    // four records per wave instead of one.
    const uint32_t lane = threadIdx.x & 15u, quarter = threadIdx.x >> 4;
    __syncthreads();                                                // (the list is complete)
    const bool listed = fast_path && work_n[1] == 0u;
    const uint32_t n_it = listed ? work_n[0] : r_hi - r_lo;
    for (uint32_t it = quarter; it < n_it; it += kMateThreads / 16) {
        const uint32_t ri = r_lo + (listed ? (uint32_t)work[it] : it);
        const RecDesc a = desc[ri];
        if (a.kind == 0 || a.pos >= te || a.end <= ts) continue;
        const uint32_t sample = S > 1 ? a.sample : 0u;
        const int32_t p0 = a.pos > ts ? a.pos : ts, p1 = a.end < te ? a.end : te;
        const uint32_t np = ext ? n_partners[ri] : 0u;
        if (!listed && fast_path && np <= 1u && single_run_ok(a)) {
            // handled by phase 1?  (the same test, with the same inputs)
            const uint32_t mi_f = mate[ri];
            RecDesc bf;
            bf.kind = 0; bf.pos = 0; bf.end = 0; bf.l_seq = 0; bf.q_start = 0;
            if (mi_f != 0xFFFFFFFFu) bf = desc[mi_f];
            const int32_t fi0 = ts > a.pos ? ts - a.pos : 0;
            const int32_t fi1 = a.end - a.pos < te - a.pos ? a.end - a.pos : te - a.pos;
            const uint32_t ft0 = (uint32_t)(a.pos + fi0 - ts), fn = fi1 > fi0 ? (uint32_t)(fi1 - fi0) : 0u;
            const uint32_t fblk = fn ? ((ft0 + fn - 1u) >> 4) - (ft0 >> 4) + 1u : 0u;
            if (eligible_fast(a, np, mi_f != 0xFFFFFFFFu, bf) && fblk <= 11u) continue;
        }
        if (np >= 2u) {
            // ---- two or three same-name partners: pairing and status per interval of constant company ----------------
            const MultiPlan P = plan_multi(desc, ri, a, ext, np);
            if (P.too_many) { if (lane == 0) atomicOr(too_many, 1u); continue; }
            for (int32_t p = p0 + (int32_t)lane; p < p1; p += 16) {
                const Cursor ca = state_at(U, a, p);
                if (ca.kind == 0) continue;
                if (kSpan) atomicAdd(&spn[p - ts], 1u);
                int v = 0;
                while (v + 1 < P.n_iv && p >= P.cut[v + 1]) ++v;
                const uint32_t mi2 = P.partner[v];
                // processed on its own unless `detected`: paired now or earlier, and never alone in between
                uint32_t times = (p >= P.first_paired && p < P.first_alone_again) ? 0u : 1u;
                if (mi2 != 0xFFFFFFFFu) {
                    const RecDesc b2 = desc[mi2];
                    const Cursor cb = state_at(U, b2, p);
                    if (cb.kind != 0) {
                        const bool a_first = ri < mi2;
                        const Cursor& c1 = a_first ? ca : cb;
                        const Cursor& c2 = a_first ? cb : ca;
                        const uint32_t q1 = a_first ? a.mapq : b2.mapq, q2 = a_first ? b2.mapq : a.mapq;
                        bool first_wins;
                        if (c1.kind != 1 || c2.kind != 1) first_wins = q1 > q2;
                        else first_wins = c1.qual > c2.qual;
                        if (first_wins == a_first) ++times;          // the better mate of the pair is processed (once more)
                    }
                }
                if (!times) continue;
                uint32_t* cp = &cnt[pos_dw_m((uint32_t)(p - ts), sub_dw, s7) + sample * 7];
                if (ca.kind == 1) { if (ca.qual >= min_bq) atomicAdd(&cp[base5_m(ca.nib)], times); }
                else atomicAdd(&cp[ca.kind == 2 ? 5 : 6], times);
            }
            continue;
        }
        const uint32_t mi = mate[ri];
        RecDesc b;
        b.kind = 0; b.pos = 0; b.end = 0; b.rec_off = 0; b.l_seq = 0; b.n_cigar = 0; b.l_name = 0; b.q_start = 0; b.sample = 0; b.mapq = 0;
        if (mi != 0xFFFFFFFFu) b = desc[mi];
        for (int32_t p = p0 + (int32_t)lane; p < p1; p += 16) {
            const Cursor ca = state_at(U, a, p);
            if (ca.kind == 0) continue;
            if (kSpan) atomicAdd(&spn[p - ts], 1u);
            bool counts = true;
            if (b.kind != 0) {
                const Cursor cb = state_at(U, b, p);
                if (cb.kind != 0) {
                    // selectBetterMate(m1, m2) with m1 = the record earlier in the file; ties -> m2
                    const bool a_first = ri < mi;
                    const Cursor& c1 = a_first ? ca : cb;
                    const Cursor& c2 = a_first ? cb : ca;
                    const uint32_t q1 = a_first ? a.mapq : b.mapq, q2 = a_first ? b.mapq : a.mapq;
                    bool first_wins;
                    if (c1.kind != 1 || c2.kind != 1) first_wins = q1 > q2;
                    else first_wins = c1.qual > c2.qual;
                    counts = (first_wins == a_first);
                }
            }
            if (!counts) continue;
            uint32_t* cp = &cnt[pos_dw_m((uint32_t)(p - ts), sub_dw, s7) + sample * 7];
            if (ca.kind == 1) { if (ca.qual >= min_bq) atomicAdd(&cp[base5_m(ca.nib)], 1u); }
            else atomicAdd(&cp[ca.kind == 2 ? 5 : 6], 1u);
        }
    }
    __syncthreads();
    const uint32_t n_cnt = T * s7;
    uint32_t* out = counters + (size_t)blockIdx.x * n_cnt;
    for (uint32_t i = threadIdx.x; i < n_cnt; i += kMateThreads) {
        const uint32_t p = i / s7, k = i - p * s7;
        out[i] = cnt[pos_dw_m(p, sub_dw, s7) + k];
    }
    if (kSpan) {
        uint32_t* so = span_out + (size_t)blockIdx.x * T;
        if (!fast_path) {
            for (uint32_t i = threadIdx.x; i < T; i += kMateThreads) so[i] = spn[i];
        } else {
            // per-position counts of phase 2 + the integrated difference array of phase 1: every thread owns `chunk` consecutive positions
            const uint32_t wl = threadIdx.x & 63u, wv = threadIdx.x >> 6;
            const uint32_t chunk = (T + kMateThreads - 1) / kMateThreads;
            const uint32_t q0 = threadIdx.x * chunk;
            int32_t loc = 0;
            for (uint32_t k = 0; k < chunk; ++k) if (q0 + k < T) loc += sdf[q0 + k];
            int32_t incl = loc;
#pragma unroll
            for (int dlt = 1; dlt < 64; dlt <<= 1) {
                const int32_t o = __shfl_up(incl, dlt, 64);
                if ((int)wl >= dlt) incl += o;
            }
            int32_t* wtot = sdf + T + 1;
            if (wl == 63) wtot[wv] = incl;
            __syncthreads();
            int32_t run = incl - loc;
            for (uint32_t w = 0; w < wv; ++w) run += wtot[w];
            for (uint32_t k = 0; k < chunk; ++k)
                if (q0 + k < T) { run += sdf[q0 + k]; so[q0 + k] = spn[q0 + k] + (uint32_t)run; }
        }
    }
}

// Per-column quantities of `depth region|window --fix-mate-overlaps` (PerRegionPrinter.push with mate
// fixing, depth.d:760-845), for every position of a tile and every sample:
//   covm = what the column contributes to the threshold test: reads that are not paired at this column count
//          with their own base quality (D/N: 255), a pair counts once, with the better mate's (depth.d:820-836);
//   addm = what the column adds to n_bases through process_base: the better mate of every pair, and every read
//          that HAS BEEN paired earlier and is alone again (status "past", depth.d:824-826) -- D/N included.
// A read's status at a column follows from geometry alone: before the overlap with its partner it is "none", inside
// it the two form a pair, after it the survivor is "past" (detectOverlappingMates, depth.d:319-388).
__global__ __launch_bounds__(kMateThreads) void k_mates_columns(
    const uint8_t* __restrict__ U, const RecDesc* __restrict__ desc, const uint32_t* __restrict__ mate,
    const uint32_t* __restrict__ tile_lo, const uint32_t* __restrict__ tile_hi, const uint32_t* __restrict__ active,
    const uint32_t* __restrict__ tile_base, int32_t n_ref, uint32_t T, uint32_t S, uint32_t min_bq,
    uint32_t* __restrict__ covm_out, uint32_t* __restrict__ addm_out, uint32_t* __restrict__ span_out) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    uint32_t* covm = lds;                 // [T][S]
    uint32_t* addm = lds + T * S;         // [T][S]
    uint32_t* spn = lds + 2 * T * S;      // [T]
    const uint32_t tile = active[blockIdx.x];
    for (uint32_t i = threadIdx.x; i < 2 * T * S + T; i += kMateThreads) lds[i] = 0;
    int lo_r = 0, hi_r = n_ref;
    while (hi_r - lo_r > 1) {
        int mid = (lo_r + hi_r) >> 1;
        if (tile_base[mid] <= tile) lo_r = mid; else hi_r = mid;
    }
    const int32_t ts = (int32_t)((tile - tile_base[lo_r]) * T), te = ts + (int32_t)T;
    const uint32_t r_lo = tile_lo[tile], r_hi = tile_hi[tile];
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // (wave-uniform for the compiler too)
    for (uint32_t ri = r_lo + wave; ri < r_hi; ri += kMateThreads / 64) {
        const RecDesc a = desc[ri];
        if (a.kind == 0 || a.pos >= te || a.end <= ts) continue;
        const uint32_t mi = mate[ri];
        RecDesc b;
        b.kind = 0; b.pos = 0; b.end = 0; b.rec_off = 0; b.l_seq = 0; b.n_cigar = 0; b.l_name = 0; b.q_start = 0; b.sample = 0; b.mapq = 0;
        if (mi != 0xFFFFFFFFu) b = desc[mi];
        const uint32_t sample = S > 1 ? a.sample : 0u;
        const int32_t ob = b.kind != 0 ? (a.end < b.end ? a.end : b.end) : 0x7FFFFFFF;     // end of the overlap with the partner
        const int32_t p0 = a.pos > ts ? a.pos : ts, p1 = a.end < te ? a.end : te;
        for (int32_t p = p0 + (int32_t)lane; p < p1; p += 64) {
            const Cursor ca = state_at(U, a, p);
            if (ca.kind == 0) continue;
            atomicAdd(&spn[p - ts], 1u);
            uint32_t q = ca.kind == 1 ? ca.qual : 255u;
            bool adds = false;                       // does this column add to n_bases through process_base?
            if (b.kind != 0) {
                const Cursor cb = state_at(U, b, p);
                if (cb.kind != 0) {
                    if (!(ri < mi)) continue;        // the pair is handled once, by its first record
                    // selectBetterMate(m1, m2), m1 = the record earlier in the file; ties -> m2 (depth.d:391-399)
                    bool first_wins;
                    if (ca.kind != 1 || cb.kind != 1) first_wins = a.mapq > b.mapq;
                    else first_wins = ca.qual > cb.qual;
                    if (!first_wins) q = cb.kind == 1 ? cb.qual : 255u;
                    adds = true;
                } else if (p >= ob) {
                    adds = true;                     // alone again after having been paired: "past"
                }
            }
            if (q >= min_bq) {
                atomicAdd(&covm[(uint32_t)(p - ts) * S + sample], 1u);
                if (adds) atomicAdd(&addm[(uint32_t)(p - ts) * S + sample], 1u);
            }
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < T * S; i += kMateThreads) {
        covm_out[(size_t)blockIdx.x * T * S + i] = covm[i];
        addm_out[(size_t)blockIdx.x * T * S + i] = addm[i];
    }
    for (uint32_t i = threadIdx.x; i < T; i += kMateThreads) span_out[(size_t)blockIdx.x * T + i] = spn[i];
}

}  // namespace

void launch_find_mates(const uint8_t* d_U, const RecDesc* d_desc, const uint64_t* d_hash, const int32_t* d_rec_ref, uint64_t n_records,
                       uint32_t* d_mate, uint32_t* d_n_partners, hipStream_t stream) {
    if (!n_records) return;
    hipLaunchKernelGGL(k_find_mates_join, dim3((uint32_t)((n_records + kMateThreads - 1) / kMateThreads)), dim3(kMateThreads), 0, stream,
                           d_U, d_desc, d_hash, d_rec_ref, n_records, d_mate, d_n_partners);
    SBX_HIP(hipGetLastError());
}

namespace {
// Several BAMs with --fix-mate-overlaps: the reference merges the files into ONE stream before the pileup (multireader.d:265-268) and
// pairs same-name, same-sample records of a column whatever file they came from (depth.d:338-377); the engine runs every file
// through the pipeline on its own and pairs within a file.  The two are the same iff no admitted record of file A shares name and
// sample with an admitted record of file B that overlaps it -- which this kernel decides: one lane per record of A, a binary search
// for the first record of B that could overlap it (B is coordinate-sorted; its alignments are at most max_span_b long), then the
// records of B that start inside A's span.  A hit sets *flag; the engine refuses the run (no silent divergence).
__global__ __launch_bounds__(256) void k_cross_file_mates(const uint8_t* __restrict__ Ua, const RecDesc* __restrict__ da,
                                                          const uint64_t* __restrict__ ha, const int32_t* __restrict__ ra, uint64_t na,
                                                          const uint8_t* __restrict__ Ub, const RecDesc* __restrict__ db,
                                                          const uint64_t* __restrict__ hb, const int32_t* __restrict__ rb, uint64_t nb,
                                                          uint32_t max_span_b, uint32_t* flag) {
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= na) return;
    const RecDesc a = da[i];
    if (a.kind == 0) return;
    const int32_t ref = ra[i];
    const uint64_t h = ha[i];
    // (ref, pos) of a coordinate-sorted file ascend, the unmapped reads (ref -1) come last
    const int64_t want_pos = (int64_t)a.pos - (int64_t)max_span_b;
    uint64_t lo = 0, hi = nb;
    while (lo < hi) {
        const uint64_t m = (lo + hi) >> 1;
        const int32_t r = rb[m];
        const bool before = r >= 0 && (r < ref || (r == ref && (int64_t)db[m].pos <= want_pos));
        if (before) lo = m + 1; else hi = m;
    }
    for (uint64_t j = lo; j < nb; ++j) {
        if (rb[j] != ref) break;
        const RecDesc b = db[j];
        if (b.pos >= a.end) break;
        if (b.kind != 0 && hb[j] == h && b.sample == a.sample && b.end > a.pos && a.l_name == b.l_name) {
            const uint8_t* x = Ua + a.rec_off + 36;
            const uint8_t* y = Ub + b.rec_off + 36;
            bool same = true;
            for (uint32_t k = 0; k < a.l_name && same; ++k) same = x[k] == y[k];
            if (same) { atomicOr(flag, 1u); return; }
        }
    }
}
}  // namespace

void launch_cross_file_mates(const uint8_t* d_Ua, const RecDesc* d_desc_a, const uint64_t* d_hash_a, const int32_t* d_ref_a, uint64_t n_a,
                             const uint8_t* d_Ub, const RecDesc* d_desc_b, const uint64_t* d_hash_b, const int32_t* d_ref_b, uint64_t n_b,
                             uint32_t max_span_b, uint32_t* d_flag, hipStream_t stream) {
    if (!n_a || !n_b) return;
    hipLaunchKernelGGL(k_cross_file_mates, dim3((uint32_t)((n_a + 255) / 256)), dim3(256), 0, stream, d_Ua, d_desc_a, d_hash_a, d_ref_a, n_a,
                       d_Ub, d_desc_b, d_hash_b, d_ref_b, n_b, max_span_b, d_flag);
    SBX_HIP(hipGetLastError());
}

void launch_find_partners(const uint8_t* d_U, const RecDesc* d_desc, const uint64_t* d_hash, const int32_t* d_rec_ref, uint64_t n_records,
                          uint32_t* d_ext, uint32_t* d_n_partners, hipStream_t stream) {
    if (!n_records) return;
    hipLaunchKernelGGL(k_find_partners, dim3((uint32_t)((n_records + kMateThreads - 1) / kMateThreads)), dim3(kMateThreads), 0, stream,
                       d_U, d_desc, d_hash, d_rec_ref, n_records, d_ext, d_n_partners);
    SBX_HIP(hipGetLastError());
}

void launch_accumulate_mates(const uint8_t* d_U, const RecDesc* d_desc, const uint32_t* d_mate, const uint32_t* d_tile_lo,
                             const uint32_t* d_tile_hi, const uint32_t* d_active, uint32_t n_active, const uint32_t* d_tile_base,
                             int32_t n_ref, uint32_t tile_pos, uint32_t n_samples, uint32_t min_bq, const uint32_t* d_ext,
                             const uint32_t* d_n_partners, uint32_t* d_too_many, uint32_t* d_counters, uint32_t* d_span, hipStream_t stream) {
    if (!n_active) return;
    static const int fast = [] { const char* e = getenv("SBX_K7_VARIANT"); return e ? atoi(e) : 1; }();
    size_t lds = ((size_t)(tile_pos / 4) * n_samples * 7 + 8) * 16 + (d_span ? (size_t)tile_pos * 4 : 0) +
                 (fast ? (size_t)(kMateThreads / 64) * kMateMapBytes + kMateWorkCap * 2 + 8 + (d_span ? (size_t)(tile_pos + 5) * 4 : 0) : 0);
    if (d_span) {
        SBX_HIP(hipFuncSetAttribute((const void*)k_accumulate_mates<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_accumulate_mates<true>, dim3(n_active), dim3(kMateThreads), lds, stream, d_U, d_desc, d_mate, d_tile_lo,
                           d_tile_hi, d_active, d_tile_base, n_ref, tile_pos, n_samples, min_bq, d_ext, d_n_partners, d_too_many, d_counters, d_span, fast);
    } else {
        SBX_HIP(hipFuncSetAttribute((const void*)k_accumulate_mates<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_accumulate_mates<false>, dim3(n_active), dim3(kMateThreads), lds, stream, d_U, d_desc, d_mate, d_tile_lo,
                           d_tile_hi, d_active, d_tile_base, n_ref, tile_pos, n_samples, min_bq, d_ext, d_n_partners, d_too_many, d_counters, d_span, fast);
    }
    SBX_HIP(hipGetLastError());
}

}  // namespace sbx

namespace sbx {
void launch_mates_columns(const uint8_t* d_U, const RecDesc* d_desc, const uint32_t* d_mate, const uint32_t* d_tile_lo,
                          const uint32_t* d_tile_hi, const uint32_t* d_active, uint32_t n_active, const uint32_t* d_tile_base,
                          int32_t n_ref, uint32_t tile_pos, uint32_t n_samples, uint32_t min_bq, uint32_t* d_covm, uint32_t* d_addm,
                          uint32_t* d_span, hipStream_t stream) {
    if (!n_active) return;
    const size_t lds = ((size_t)2 * tile_pos * n_samples + tile_pos) * 4;
    SBX_HIP(hipFuncSetAttribute((const void*)k_mates_columns, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_mates_columns, dim3(n_active), dim3(kMateThreads), lds, stream, d_U, d_desc, d_mate, d_tile_lo, d_tile_hi,
                       d_active, d_tile_base, n_ref, tile_pos, n_samples, min_bq, d_covm, d_addm, d_span);
    SBX_HIP(hipGetLastError());
}
}  // namespace sbx
