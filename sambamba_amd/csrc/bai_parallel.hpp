// bai_parallel.hpp -- the BAI of a coordinate-sorted BAM as a data-parallel computation: one step per record, in any order.
//
// IndexBuilder (BioD/bio/std/hts/bam/bai/indexing.d:52-346) is a loop over the reads of a file that carries the previous read
// along; bai_writer.hpp restates it as such (it stays the checker of this file and the path for irregular input).  Everything the
// loop computes is a function of a record, its predecessor and sums / minima / maxima over records, so it does not need the loop:
//   * a chunk is a maximal run of consecutive placed records (reference id and position >= 0) with one reference and one stored
//     bin; it begins where the previous run ended (indexing.d:215-240: chunk_beg = the previous read's end offset) -- so a run is
//     described by its FIRST record alone {record index, reference, bin, end offset of the record before it}, and its end is the
//     begin of the next run.  Run heads are found by comparing a record with the placed record before it;
//   * the linear index of a reference keeps, per 16 kbp window, the start offset of the first read overlapping it (:124-156) --
//     reads arrive in file order, so "first" is the minimum;
//   * the metadata pseudo-bin (:107-122, :182-188) counts the records between the first placed record of a reference and the first
//     placed record of the next one, and keeps the end offset of the last of them -- sums and a maximum keyed by the reference of
//     the last placed record at or before a record;
//   * the number of records without a reference (:341) is a count.
// bai_record_step is that step (device: index.hip's descriptors in, atomics on small per-reference arrays out, one lane per
// record; host: the CPU test calls it for the records in shuffled order and compares the file with BaiBuilder's).  The run
// list (a record in fifty is a run head on a 30x BAM) and the per-reference arrays go to the host, where BaiAssembler writes the
// file: per reference the runs in order (the "same BGZF block" merge rule of :215-240 applies between consecutive chunks of one
// bin), bins in ascending id order, pseudo-bin, linear index with empty windows repeating the last filled one (:158-170).
// Virtual offsets are binary searches in the file's block table (bai_writer.hpp VoffCursor states the rule; stateless here).
// Whatever the loop treats specially and this formulation does not -- unsorted input (the reference throws), more linear-index
// windows than the reference's length asks for, more runs than the buffer holds -- raises `irregular`, and the caller repeats the
// build with the serial builder, which also words the error.
#pragma once
#include <algorithm>
#include <cstdint>
#include <map>
#include <vector>

#include "kernels.hpp"

namespace sbx {

constexpr uint32_t kBaiWindows = 37450 - 4681 + 1;     // linear-index windows IndexBuilder keeps per reference (indexing.d:66)
constexpr uint32_t kBaiBackScan = 256;                   // records bai_record_step looks back for the placed record before its own

struct BaiRun {            // a chunk in the making: its first record
    uint64_t rec;          // index of the record in the file (runs are sorted by it on the host)
    int32_t ref;
    uint32_t bin;
    uint64_t beg_vo;
};

struct BaiCarry {          // the last placed record before a batch (sbx_build_index streams the file in batches)
    int32_t ref, pos;
    uint32_t bin, have;
    uint64_t end_vo;
};

enum BaiScalar { kBaiNoCoord = 0, kBaiFirstVo = 1, kBaiIrregular = 2, kBaiNumRuns = 3, kBaiLastPlaced = 4, kBaiScalars = 8 };

struct BaiArgs {
    // the records of the batch, in file order (index.hip describe) and the inflated bytes they point into
    const uint8_t* U;
    const RecDesc* desc;
    const int32_t* rec_ref;
    uint64_t n;
    uint64_t rec_base;         // index in the file of record 0 of the batch
    uint64_t u_base;           // offset in the file's inflated stream of U[0]
    uint64_t u_next;           // ... of the byte behind the last record of the batch
    // the file's block table: block b starts at file offset coff[b] and holds the inflated bytes [ustart[b], ustart[b + 1])
    const uint64_t* coff;
    const uint64_t* ustart;
    uint32_t n_blocks;
    uint64_t file_end;
    BaiCarry carry;
    int32_t n_ref;
    // results, accumulated over the batches
    uint64_t* lin;             // [lin_off[n_ref]] minima, ~0 = empty
    const uint32_t* lin_off;   // [n_ref + 1] first window of a reference in lin
    uint32_t* lin_len;         // [n_ref] IndexBuilder's linear_len
    uint64_t* meta_end;        // [n_ref + 1] end offset of the last record of the reference's stretch (slot n_ref: before any placed record)
    uint64_t* n_mapped;        // [n_ref + 1]
    uint64_t* n_unmapped;      // [n_ref + 1]
    unsigned long long* scalars;   // [kBaiScalars]
    BaiRun* runs;
    uint64_t runs_cap;
};

namespace bai_detail {
#if defined(__HIP_DEVICE_COMPILE__) && __HIP_DEVICE_COMPILE__
__device__ __forceinline__ void amin(uint64_t* p, uint64_t v) { atomicMin((unsigned long long*)p, (unsigned long long)v); }
__device__ __forceinline__ void amax(uint64_t* p, uint64_t v) { atomicMax((unsigned long long*)p, (unsigned long long)v); }
__device__ __forceinline__ void amax32(uint32_t* p, uint32_t v) { atomicMax(p, v); }
__device__ __forceinline__ uint64_t aadd(uint64_t* p, uint64_t v) { return atomicAdd((unsigned long long*)p, (unsigned long long)v); }
#else
inline void amin(uint64_t* p, uint64_t v) { if (v < *p) *p = v; }
inline void amax(uint64_t* p, uint64_t v) { if (v > *p) *p = v; }
inline void amax32(uint32_t* p, uint32_t v) { if (v > *p) *p = v; }
inline uint64_t aadd(uint64_t* p, uint64_t v) { const uint64_t o = *p; *p = o + v; return o; }
#endif

// number of entries of t[0, n) that are <= u (upper bound) / < u (lower bound)
__host__ __device__ inline uint32_t upper(const uint64_t* t, uint32_t n, uint64_t u) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) { const uint32_t m = (lo + hi) >> 1; if (t[m] <= u) lo = m + 1; else hi = m; }
    return lo;
}
__host__ __device__ inline uint32_t lower(const uint64_t* t, uint32_t n, uint64_t u) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) { const uint32_t m = (lo + hi) >> 1; if (t[m] < u) lo = m + 1; else hi = m; }
    return lo;
}
}  // namespace bai_detail

// virtual offset OF the byte at stream offset u (the start of a record): the block that holds it
__host__ __device__ inline uint64_t bai_vo_of(const BaiArgs& a, uint64_t u) {
    const uint32_t k = bai_detail::upper(a.ustart, a.n_blocks + 1, u);       // blocks that start at or before u
    if (k == 0 || k > a.n_blocks) return a.file_end << 16;
    return (a.coff[k - 1] << 16) | (u - a.ustart[k - 1]);
}
// virtual offset BEHIND the byte u - 1 (the end of a record): offset 0 of the first block that starts at u if there is one -- the
// reader has moved on to it, even if it is empty (the end of the last read of a file is the start of its EOF block) --, else the
// block that holds u
__host__ __device__ inline uint64_t bai_vo_behind(const BaiArgs& a, uint64_t u) {
    const uint32_t k = bai_detail::lower(a.ustart, a.n_blocks + 1, u);
    if (k < a.n_blocks && a.ustart[k] == u) return a.coff[k] << 16;
    return bai_vo_of(a, u);
}

struct BaiFields { int32_t ref, pos, end; uint32_t bin; bool unmapped, placed; };
__host__ __device__ inline BaiFields bai_fields(const BaiArgs& a, uint64_t i) {
    BaiFields f;
    const RecDesc d = a.desc[i];
    f.ref = a.rec_ref[i];
    f.pos = d.pos;
    f.end = d.end;
    const uint8_t* r = a.U + d.rec_off + 4;
    f.bin = (uint32_t)r[10] | ((uint32_t)r[11] << 8);          // the stored bin (bin_mq_nl >> 16, read.d:919-921)
    f.unmapped = (d.flag & 0x4) != 0;
    f.placed = f.ref >= 0 && f.pos >= 0;
    return f;
}
__host__ __device__ inline uint64_t bai_end_vo(const BaiArgs& a, uint64_t i) {
    return bai_vo_behind(a, a.u_base + (i + 1 < a.n ? a.desc[i + 1].rec_off : a.u_next - a.u_base));
}

// one record of the batch
__host__ __device__ inline void bai_record_step(const BaiArgs& a, uint64_t i) {
    using namespace bai_detail;
    uint64_t* sc = (uint64_t*)a.scalars;
    const BaiFields f = bai_fields(a, i);
    if (f.ref < 0) { aadd(sc + kBaiNoCoord, 1); return; }      // (indexing.d:108-111, :264: no further part in the index)
    const uint64_t svo = bai_vo_of(a, a.u_base + a.desc[i].rec_off), evo = bai_end_vo(a, i);
    // the placed record before this one: nearly always the record before it
    bool have_p = false;
    int32_t pref = -1, ppos = 0;
    uint32_t pbin = 0;
    uint64_t pevo = 0;
    // (a stretch of more than kBaiBackScan records with a reference but no position in front of this one -- O(n^2) loads in one launch
    // if every record of a long stretch walked all of it -- is irregular input: the serial builder takes the file)
    bool scan_cut = false;
    {
        uint32_t steps = 0;
        for (uint64_t j = i; j-- > 0;) {
            const BaiFields g = bai_fields(a, j);
            if (g.placed) { have_p = true; pref = g.ref; ppos = g.pos; pbin = g.bin; pevo = bai_end_vo(a, j); break; }
            if (++steps >= kBaiBackScan) { scan_cut = true; break; }
        }
    }
    if (scan_cut) { aadd(sc + kBaiIrregular, 1); return; }
    if (!have_p && a.carry.have) { have_p = true; pref = a.carry.ref; ppos = a.carry.pos; pbin = a.carry.bin; pevo = a.carry.end_vo; }
    // checkThatInputIsSorted (:242-257)
    if (have_p && !(pref < f.ref || (pref == f.ref && f.pos >= ppos))) aadd(sc + kBaiIrregular, 1);
    if (f.ref >= a.n_ref) { aadd(sc + kBaiIrregular, 1); return; }
    // metadata of the stretch this record belongs to (:107-122)
    const int32_t seg = f.placed ? f.ref : have_p ? pref : a.n_ref;
    aadd((f.unmapped ? a.n_unmapped : a.n_mapped) + seg, 1);
    amax(a.meta_end + seg, evo);
    amin(sc + kBaiFirstVo, svo);
    if (!f.placed) return;
    amax(sc + kBaiLastPlaced, a.rec_base + i + 1);
    // linear index (:124-156)
    const uint32_t wbeg = (uint32_t)f.pos >> 14;
    const int64_t last = (int64_t)f.end - 1;
    const uint32_t wend = f.unmapped ? wbeg : last < 0 ? 0u : (uint32_t)(last >> 14);
    const uint32_t room = a.lin_off[f.ref + 1] - a.lin_off[f.ref];
    for (uint32_t w = wbeg; w <= wend && w < kBaiWindows; ++w) {
        if (w >= room) { aadd(sc + kBaiIrregular, 1); break; }
        amin(a.lin + a.lin_off[f.ref] + w, svo);
    }
    amax32(a.lin_len + f.ref, wend + 1);
    // a run head (:283-296)
    if (!have_p || pref != f.ref || pbin != f.bin) {
        const uint64_t slot = aadd(sc + kBaiNumRuns, 1);
        if (slot < a.runs_cap) a.runs[slot] = BaiRun{a.rec_base + i, f.ref, f.bin, have_p ? pevo : svo};
    }
}

// the last placed record of everything seen so far, for the next batch (run by one thread after the steps of a batch)
__host__ __device__ inline void bai_carry_out(const BaiArgs& a, BaiCarry* out) {
    const uint64_t last = ((const uint64_t*)a.scalars)[kBaiLastPlaced];
    if (last <= a.rec_base) { *out = a.carry; return; }        // no placed record in this batch
    const uint64_t i = last - 1 - a.rec_base;
    const BaiFields f = bai_fields(a, i);
    *out = BaiCarry{f.ref, f.pos, f.bin, 1u, bai_end_vo(a, i)};
}

// ---- host: the file from the runs and the per-reference arrays --------------------------------------------------------------
struct BaiHostResults {
    std::vector<BaiRun> runs;                  // of all batches
    std::vector<uint64_t> lin;
    std::vector<uint32_t> lin_off, lin_len;
    std::vector<uint64_t> meta_end, n_mapped, n_unmapped;
    uint64_t scalars[kBaiScalars];
    BaiCarry last;                             // the last placed record of the file
};

inline std::vector<uint8_t> bai_assemble(int n_ref, BaiHostResults& R) {
    std::vector<uint8_t> out{'B', 'A', 'I', 1};
    auto put32 = [&](uint32_t v) { for (int k = 0; k < 4; ++k) out.push_back((uint8_t)(v >> (8 * k))); };
    auto put64 = [&](uint64_t v) { for (int k = 0; k < 8; ++k) out.push_back((uint8_t)(v >> (8 * k))); };
    put32((uint32_t)n_ref);
    std::sort(R.runs.begin(), R.runs.end(), [](const BaiRun& x, const BaiRun& y) { return x.rec < y.rec; });
    size_t j = 0;
    bool first_dumped = true;
    for (int r = 0; r < n_ref; ++r) {
        if (j >= R.runs.size() || R.runs[j].ref != r) { put32(0); put32(0); continue; }     // no placed read: an empty reference
        const uint64_t ref_beg = R.runs[j].beg_vo;
        std::map<uint32_t, std::vector<std::pair<uint64_t, uint64_t>>> bins;
        for (; j < R.runs.size() && R.runs[j].ref == r; ++j) {
            const uint64_t beg = R.runs[j].beg_vo, end = j + 1 < R.runs.size() ? R.runs[j + 1].beg_vo : R.last.end_vo;
            auto& cs = bins[R.runs[j].bin];
            if (cs.empty() || (cs.back().second >> 16) != (beg >> 16)) cs.push_back({beg, end});
            else cs.back().second = end;
        }
        put32((uint32_t)bins.size() + 1);
        for (auto& kv : bins) {
            put32(kv.first);
            put32((uint32_t)kv.second.size());
            for (auto& c : kv.second) { put64(c.first); put64(c.second); }
        }
        // the pseudo-bin: the first reference written starts at the first record of the file that has a reference, every later
        // one where the reference before it ended (:190-204); counts of the records in front of the first placed one go to the first
        uint64_t mapped = R.n_mapped[(size_t)r], unmapped = R.n_unmapped[(size_t)r], end_vo = R.meta_end[(size_t)r];
        if (first_dumped) { mapped += R.n_mapped[(size_t)n_ref]; unmapped += R.n_unmapped[(size_t)n_ref]; end_vo = std::max(end_vo, R.meta_end[(size_t)n_ref]); }
        put32(37450);
        put32(2);
        put64(first_dumped ? R.scalars[kBaiFirstVo] : ref_beg); put64(end_vo); put64(mapped); put64(unmapped);
        first_dumped = false;
        const uint32_t n = std::min<uint32_t>(R.lin_len[(size_t)r], kBaiWindows);
        put32(n);
        uint64_t last = 0;
        for (uint32_t w = 0; w < n; ++w) {
            uint64_t v = w < R.lin_off[(size_t)r + 1] - R.lin_off[(size_t)r] ? R.lin[R.lin_off[(size_t)r] + w] : ~0ull;
            if (v == ~0ull) v = last; else last = v;
            put64(v);
        }
    }
    put64(R.scalars[kBaiNoCoord]);
    return out;
}

// windows kept for a reference of this length: what its positions need, plus one for reads that end on the boundary
inline uint32_t bai_windows_for(int32_t length) {
    const uint64_t w = ((uint64_t)std::max(0, length) >> 14) + 2;
    return (uint32_t)std::min<uint64_t>(w, kBaiWindows);
}

}  // namespace sbx
